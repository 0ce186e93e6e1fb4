"""Pure-python HDF5 reader (+ a small writer) for Keras ``.h5`` archives (reference: deeplearning4j-modelimport/.../keras/
Hdf5Archive.java, which binds the HDF5 C library through JavaCPP; neither that library nor h5py is available here, so the file
format is parsed directly).

Scope = what h5py writes by default for Keras models: superblock v0/v1, version-1 object headers (+ continuation blocks),
old-style groups (symbol-table message -> v1 B-tree -> SNOD -> local heap), contiguous / compact / chunked (v1 B-tree, deflate +
shuffle filters) dataset layouts, fixed-point / IEEE float / fixed-length string / variable-length string datatypes (global heap),
attribute messages v1-v3.  ``Hdf5Writer`` emits the same classic structures (contiguous datasets, fixed-length string
attributes) and is used for export and for the self-consistency tests — no h5py-written fixture ships with the reference, so
parity with h5py output itself is unpinned.
"""
from __future__ import annotations

import json
import struct
import zlib
from typing import Dict, List, Optional, Tuple

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5FormatError(ValueError):
    pass


class _Type:
    def __init__(self, cls, size, np_dtype=None, vlen_str=False, base=None, strpad=0):
        self.cls, self.size, self.np_dtype, self.vlen_str, self.base, self.strpad = cls, size, np_dtype, vlen_str, base, strpad


class Hdf5Archive:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = f.read()
        self._parse_superblock()
        self._gheap: Dict[int, Dict[int, bytes]] = {}

    # ------------------------------------------------------------------ low level
    def _u(self, off, n):
        return int.from_bytes(self.buf[off:off + n], "little")

    def _parse_superblock(self):
        base = self.buf.find(SIG)
        if base != 0:
            raise Hdf5FormatError("not an HDF5 file (signature not at offset 0)")
        ver = self.buf[8]
        if ver not in (0, 1):
            raise Hdf5FormatError("unsupported superblock version %d (only the classic v0/v1 layout written by default h5py is supported)" % ver)
        self.O, self.Ln = self.buf[13], self.buf[14]
        p = 24 if ver == 0 else 28
        p += 4 * self.O                                   # base, free-space, eof, driver addresses
        # root group symbol table entry
        self.root_header = self._u(p + self.O, self.O)

    def _messages(self, addr) -> List[Tuple[int, int, int]]:
        """[(type, data offset, size)] of a version-1 object header, following continuation messages."""
        if self.buf[addr] != 1:
            raise Hdf5FormatError("unsupported object header version %d at 0x%x" % (self.buf[addr], addr))
        nmsg = self._u(addr + 2, 2)
        hsize = self._u(addr + 8, 4)
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize = self._u(p, 2), self._u(p + 2, 2)
                data = p + 8
                if mtype == 0x10:
                    blocks.append((self._u(data, self.O), self._u(data + self.O, self.Ln)))
                out.append((mtype, data, msize))
                p = data + msize
        return out

    # ------------------------------------------------------------------ groups
    def _heap_str(self, heap_addr, off):
        if self.buf[heap_addr:heap_addr + 4] != b"HEAP":
            raise Hdf5FormatError("bad local heap signature")
        data = self._u(heap_addr + 8 + 2 * self.Ln, self.O)
        s = data + off
        e = self.buf.index(b"\0", s)
        return self.buf[s:e].decode("utf-8")

    def _btree_group(self, node, heap) -> Dict[str, int]:
        if self.buf[node:node + 4] != b"TREE":
            raise Hdf5FormatError("bad B-tree signature")
        level, used = self.buf[node + 5], self._u(node + 6, 2)
        p = node + 8 + 2 * self.O
        out: Dict[str, int] = {}
        for i in range(used):
            child = self._u(p + self.Ln + i * (self.Ln + self.O), self.O)
            if level > 0:
                out.update(self._btree_group(child, heap))
            else:
                if self.buf[child:child + 4] != b"SNOD":
                    raise Hdf5FormatError("bad symbol table node signature")
                n = self._u(child + 6, 2)
                q = child + 8
                for _ in range(n):
                    name = self._heap_str(heap, self._u(q, self.O))
                    out[name] = self._u(q + self.O, self.O)
                    q += 2 * self.O + 24
        return out

    def _children(self, header) -> Optional[Dict[str, int]]:
        for t, d, _ in self._messages(header):
            if t == 0x11:
                return self._btree_group(self._u(d, self.O), self._u(d + self.O, self.O))
        return None

    def _resolve(self, path) -> int:
        h = self.root_header
        for part in [p for p in path.split("/") if p]:
            ch = self._children(h)
            if ch is None or part not in ch:
                raise KeyError(path)
            h = ch[part]
        return h

    def list(self, path="/") -> List[str]:
        ch = self._children(self._resolve(path))
        return sorted(ch) if ch else []

    getGroups = list

    def is_group(self, path):
        return self._children(self._resolve(path)) is not None

    def has(self, path):
        try:
            self._resolve(path)
            return True
        except KeyError:
            return False

    # ------------------------------------------------------------------ datatypes / dataspaces
    def _dtype(self, p) -> _Type:
        cv = self.buf[p]
        cls, bits0 = cv & 0x0F, self.buf[p + 1]
        size = self._u(p + 4, 4)
        if cls == 0:
            signed = bool(bits0 & 0x08)
            return _Type(0, size, np.dtype(("<" if not bits0 & 1 else ">") + ("i" if signed else "u") + str(size)))
        if cls == 1:
            return _Type(1, size, np.dtype(("<" if not bits0 & 1 else ">") + "f" + str(size)))
        if cls == 3:
            return _Type(3, size, strpad=bits0 & 0x0F)
        if cls == 9:
            return _Type(9, size, vlen_str=(bits0 & 0x0F) == 1, base=self._dtype(p + 8))
        raise Hdf5FormatError("unsupported datatype class %d" % cls)

    def _dspace(self, p) -> Tuple[int, ...]:
        ver, rank = self.buf[p], self.buf[p + 1]
        q = p + (8 if ver == 1 else 4)
        return tuple(self._u(q + i * self.Ln, self.Ln) for i in range(rank))

    def _gheap_obj(self, addr, idx) -> bytes:
        if addr not in self._gheap:
            if self.buf[addr:addr + 4] != b"GCOL":
                raise Hdf5FormatError("bad global heap signature")
            size = self._u(addr + 8, self.Ln)
            p, end = addr + 8 + self.Ln, addr + size
            objs = {}
            while p + 8 + self.Ln <= end:
                i = self._u(p, 2)
                n = self._u(p + 8, self.Ln)
                if i == 0:
                    break
                objs[i] = self.buf[p + 8 + self.Ln:p + 8 + self.Ln + n]
                p += 8 + self.Ln + ((n + 7) & ~7)
            self._gheap[addr] = objs
        return self._gheap[addr][idx]

    def _decode(self, t: _Type, shape, raw: bytes):
        n = int(np.prod(shape)) if shape else 1
        if t.cls in (0, 1):
            return np.frombuffer(raw, dtype=t.np_dtype, count=n).reshape(shape).astype(t.np_dtype.newbyteorder("="))
        if t.cls == 3:
            vals = [raw[i * t.size:(i + 1) * t.size].split(b"\0")[0].decode("utf-8") for i in range(n)]
            return vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
        if t.cls == 9 and t.vlen_str:
            esz = 4 + self.O + 4
            vals = []
            for i in range(n):
                q = i * esz
                addr = int.from_bytes(raw[q + 4:q + 4 + self.O], "little")
                idx = int.from_bytes(raw[q + 4 + self.O:q + esz], "little")
                vals.append(self._gheap_obj(addr, idx).decode("utf-8") if addr not in (0, UNDEF) else "")
            return vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
        raise Hdf5FormatError("unsupported datatype for decoding")

    # ------------------------------------------------------------------ attributes
    def attrs(self, path="/") -> Dict[str, object]:
        out = {}
        for t, d, size in self._messages(self._resolve(path)):
            if t != 0x0C:
                continue
            ver = self.buf[d]
            nsz, tsz, ssz = self._u(d + 2, 2), self._u(d + 4, 2), self._u(d + 6, 2)
            p = d + 8 + (1 if ver == 3 else 0)
            pad = (lambda v: (v + 7) & ~7) if ver == 1 else (lambda v: v)
            name = self.buf[p:p + nsz].split(b"\0")[0].decode("utf-8")
            p += pad(nsz)
            ty = self._dtype(p)
            p += pad(tsz)
            shape = self._dspace(p)
            p += pad(ssz)
            out[name] = self._decode(ty, shape, self.buf[p:d + size])
        return out

    def read_attribute_as_string(self, name, path="/") -> str:
        v = self.attrs(path)[name]
        return v if isinstance(v, str) else (v.item() if hasattr(v, "item") and v.size == 1 else str(v))

    readAttributeAsString = read_attribute_as_string
    readAttributeAsJson = read_attribute_as_string

    def has_attribute(self, name, path="/"):
        return name in self.attrs(path)

    hasAttribute = has_attribute

    # ------------------------------------------------------------------ datasets
    def read_dataset(self, path) -> np.ndarray:
        msgs = self._messages(self._resolve(path))
        ty = shape = layout = None
        filters: List[int] = []
        for t, d, size in msgs:
            if t == 0x03:
                ty = self._dtype(d)
            elif t == 0x01:
                shape = self._dspace(d)
            elif t == 0x08:
                layout = (d, size)
            elif t == 0x0B:
                ver, nf = self.buf[d], self.buf[d + 1]
                p = d + (8 if ver == 1 else 2)
                for _ in range(nf):
                    fid, nlen = self._u(p, 2), (self._u(p + 2, 2) if (ver == 1 or self._u(p, 2) >= 256) else 0)
                    ncv = self._u(p + 6, 2)
                    filters.append(fid)
                    p += 8 + ((nlen + 7) & ~7 if ver == 1 else nlen) + 4 * ncv
                    if ver == 1 and ncv % 2:
                        p += 4
        if ty is None or shape is None or layout is None:
            raise Hdf5FormatError("%s is not a dataset" % path)
        d, _ = layout
        ver, cls = self.buf[d], self.buf[d + 1]
        if ver != 3:
            raise Hdf5FormatError("unsupported data layout message version %d" % ver)
        n = int(np.prod(shape)) if shape else 1
        if cls == 0:
            sz = self._u(d + 2, 2)
            raw = self.buf[d + 4:d + 4 + sz]
        elif cls == 1:
            addr, sz = self._u(d + 2, self.O), self._u(d + 2 + self.O, self.Ln)
            raw = b"\0" * (n * ty.size) if addr == UNDEF else self.buf[addr:addr + sz]
        elif cls == 2:
            nd = self.buf[d + 2]
            bt = self._u(d + 3, self.O)
            cdims = tuple(self._u(d + 3 + self.O + 4 * i, 4) for i in range(nd - 1))
            out = np.zeros(shape, dtype=ty.np_dtype)
            self._read_chunks(bt, nd, cdims, ty, filters, out)
            return out.astype(ty.np_dtype.newbyteorder("="))
        else:
            raise Hdf5FormatError("unsupported layout class %d" % cls)
        return self._decode(ty, shape, raw)

    readDataSet = read_dataset

    def _read_chunks(self, node, nd, cdims, ty, filters, out):
        if node == UNDEF:
            return
        if self.buf[node:node + 4] != b"TREE":
            raise Hdf5FormatError("bad chunk B-tree signature")
        level, used = self.buf[node + 5], self._u(node + 6, 2)
        p = node + 8 + 2 * self.O
        ksz = 8 + 8 * nd
        for i in range(used):
            k = p + i * (ksz + self.O)
            csize, fmask = self._u(k, 4), self._u(k + 4, 4)
            offs = tuple(self._u(k + 8 + 8 * j, 8) for j in range(nd - 1))
            child = self._u(k + ksz, self.O)
            if level > 0:
                self._read_chunks(child, nd, cdims, ty, filters, out)
                continue
            raw = self.buf[child:child + csize]
            for j, fid in reversed(list(enumerate(filters))):
                if fmask & (1 << j):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    a = np.frombuffer(raw, dtype=np.uint8).reshape(ty.size, -1)
                    raw = a.T.tobytes()
                else:
                    raise Hdf5FormatError("unsupported filter id %d" % fid)
            chunk = np.frombuffer(raw, dtype=ty.np_dtype, count=int(np.prod(cdims))).reshape(cdims)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, out.shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]

    # ------------------------------------------------------------------ Keras conventions
    def keras_weights(self) -> Dict[str, Dict[str, np.ndarray]]:
        """{layer: {weight_path: array}} from either a full-model file (``/model_weights``) or a weights-only file."""
        root = "/model_weights" if self.has("/model_weights") else "/"
        a = self.attrs(root)
        layers = [str(x) for x in a["layer_names"]] if "layer_names" in a else [n for n in self.list(root) if self.is_group(root + "/" + n)]
        out: Dict[str, Dict[str, np.ndarray]] = {}
        for lname in layers:
            g = root.rstrip("/") + "/" + lname
            la = self.attrs(g)
            if "weight_names" in la:
                names = [str(x) for x in np.asarray(la["weight_names"]).reshape(-1)]
            else:
                names = self._walk_datasets(g, "")
            if names:
                out[lname] = {n: self.read_dataset(g + "/" + n) for n in names}
        return out

    def _walk_datasets(self, base, rel) -> List[str]:
        res = []
        for n in self.list(base + ("/" + rel if rel else "")):
            p = (rel + "/" + n) if rel else n
            if self.is_group(base + "/" + p):
                res += self._walk_datasets(base, p)
            else:
                res.append(p)
        return res


# ============================================================================= writer
class Hdf5Writer:
    """Writes groups / contiguous datasets / fixed-length-string + numeric attributes in the classic layout."""
    O = Ln = 8
    LEAF_K = 64                       # up to 128 symbols per SNOD

    def __init__(self):
        self.tree = {"groups": {}, "datasets": {}, "attrs": {}}

    def _node(self, path, create=True):
        cur = self.tree
        for part in [p for p in path.split("/") if p]:
            if part not in cur["groups"]:
                if not create:
                    raise KeyError(path)
                cur["groups"][part] = {"groups": {}, "datasets": {}, "attrs": {}}
            cur = cur["groups"][part]
        return cur

    def create_group(self, path):
        self._node(path)
        return self

    def create_dataset(self, path, arr):
        parent, _, name = path.rstrip("/").rpartition("/")
        self._node(parent)["datasets"][name] = {"data": np.ascontiguousarray(arr), "attrs": {}}
        return self

    def set_attr(self, path, name, value):
        parent, _, leaf = path.rstrip("/").rpartition("/")
        if path in ("", "/"):
            self.tree["attrs"][name] = value
            return self
        n = self._node(parent, create=True)
        if leaf in n["datasets"]:
            n["datasets"][leaf]["attrs"][name] = value
        else:
            self._node(path)["attrs"][name] = value
        return self

    # -- encoding helpers
    @staticmethod
    def _pad8(b: bytes) -> bytes:
        return b + b"\0" * (-len(b) % 8)

    @staticmethod
    def _dtype_msg(arr: np.ndarray = None, strlen: int = None) -> bytes:
        if strlen is not None:
            return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, strlen)                        # class 3 (string) v1, null-terminated ASCII/UTF-8 flags 0
        dt = arr.dtype
        if dt.kind == "f":
            size = dt.itemsize
            if size == 4:
                props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
                bits = (0x20, 0x1F, 0x00)
            else:
                props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
                bits = (0x20, 0x3F, 0x00)
            return struct.pack("<BBBBI", 0x11, bits[0], bits[1], bits[2], size) + props
        if dt.kind in "iu":
            size = dt.itemsize
            return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, size) + struct.pack("<HH", 0, 8 * size)
        raise TypeError("unsupported dtype %s" % dt)

    def _dspace_msg(self, shape) -> bytes:
        return struct.pack("<BBBB4x", 1, len(shape), 0, 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)

    def _attr_msg(self, name, value) -> bytes:
        nm = name.encode("utf-8") + b"\0"
        if isinstance(value, str):
            raw = value.encode("utf-8") + b"\0"
            ty, sp, data = self._dtype_msg(strlen=len(raw)), self._dspace_msg(()), raw
        elif isinstance(value, (list, tuple)) and value and isinstance(value[0], str):
            enc = [v.encode("utf-8") for v in value]
            sl = max(len(e) for e in enc) + 1
            ty, sp = self._dtype_msg(strlen=sl), self._dspace_msg((len(enc),))
            data = b"".join(e + b"\0" * (sl - len(e)) for e in enc)
        else:
            arr = np.ascontiguousarray(value)
            if arr.dtype == np.float16 or arr.dtype.kind == "b":
                arr = arr.astype(np.float32 if arr.dtype.kind == "f" else np.int8)
            ty, sp, data = self._dtype_msg(arr), self._dspace_msg(arr.shape), arr.astype(arr.dtype.newbyteorder("<")).tobytes()
        body = struct.pack("<BBHHH", 1, 0, len(nm), len(ty), len(sp)) + self._pad8(nm) + self._pad8(ty) + self._pad8(sp) + data
        return body

    @staticmethod
    def _msg(mtype, body: bytes) -> bytes:
        body = body + b"\0" * (-len(body) % 8)
        return struct.pack("<HHB3x", mtype, len(body), 0) + body

    def _header(self, msgs: List[bytes]) -> bytes:
        blob = b"".join(msgs)
        return struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(blob)) + blob

    # -- layout
    def write(self, path):
        self.out = bytearray(b"\0" * 96)                 # superblock v0 placeholder (24 + 4*8 + 40)
        root = self._emit_group(self.tree)
        sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.out), UNDEF)
        sb += struct.pack("<QQII16x", 0, root, 0, 0)
        self.out[:len(sb)] = sb
        with open(path, "wb") as f:
            f.write(bytes(self.out))

    def _alloc(self, data: bytes) -> int:
        pad = -len(self.out) % 8
        self.out += b"\0" * pad
        addr = len(self.out)
        self.out += data
        return addr

    def _emit_dataset(self, ds) -> int:
        arr = ds["data"]
        if arr.dtype == np.float64 or arr.dtype == np.float32 or arr.dtype.kind in "iu":
            pass
        else:
            arr = arr.astype(np.float32)
        raw = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
        daddr = self._alloc(raw) if raw else UNDEF
        msgs = [self._msg(0x01, self._dspace_msg(arr.shape)), self._msg(0x03, self._dtype_msg(arr)),
                self._msg(0x08, struct.pack("<BBQQ", 3, 1, daddr, len(raw)))]
        msgs += [self._msg(0x0C, self._attr_msg(k, v)) for k, v in ds["attrs"].items()]
        return self._alloc(self._header(msgs))

    def _emit_group(self, g) -> int:
        entries: List[Tuple[str, int]] = []
        for name, sub in g["groups"].items():
            entries.append((name, self._emit_group(sub)))
        for name, ds in g["datasets"].items():
            entries.append((name, self._emit_dataset(ds)))
        entries.sort(key=lambda e: e[0].encode("utf-8"))
        # local heap: offset 0 = empty string, then names
        heap_data = bytearray(b"\0" * 8)
        offs = {}
        for name, _ in entries:
            offs[name] = len(heap_data)
            heap_data += name.encode("utf-8") + b"\0"
            heap_data += b"\0" * (-len(heap_data) % 8)
        hd_addr = self._alloc(bytes(heap_data))
        heap_addr = self._alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), UNDEF, hd_addr))
        # symbol nodes
        cap = 2 * self.LEAF_K
        snods = []
        for i in range(0, max(len(entries), 1), cap):
            chunk = entries[i:i + cap]
            body = b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk))
            for name, addr in chunk:
                body += struct.pack("<QQII16x", offs[name], addr, 0, 0)
            body += b"\0" * ((cap - len(chunk)) * 40)
            snods.append((self._alloc(body), offs[chunk[-1][0]] if chunk else 0))
        bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF) + struct.pack("<Q", 0)
        for addr, last_key in snods:
            bt += struct.pack("<QQ", addr, last_key)
        bt += b"\0" * ((2 * 16 + 1 - len(snods)) * 16)
        bt_addr = self._alloc(bt)
        msgs = [self._msg(0x11, struct.pack("<QQ", bt_addr, heap_addr))]
        msgs += [self._msg(0x0C, self._attr_msg(k, v)) for k, v in g["attrs"].items()]
        return self._alloc(self._header(msgs))


def write_keras_h5(path, model_config: dict, weights: Dict[str, Dict[str, np.ndarray]], keras_version="2.4.0", backend="tensorflow"):
    """Write a Keras-style full-model archive (``model_config`` root attribute + ``/model_weights/<layer>/<weight>``)."""
    w = Hdf5Writer()
    w.set_attr("/", "model_config", json.dumps(model_config))
    w.set_attr("/", "keras_version", keras_version)
    w.set_attr("/", "backend", backend)
    w.create_group("/model_weights")
    w.set_attr("/model_weights", "layer_names", list(weights) or [""])
    w.set_attr("/model_weights", "backend", backend)
    w.set_attr("/model_weights", "keras_version", keras_version)
    for lname, ws in weights.items():
        g = "/model_weights/" + lname
        w.create_group(g)
        names = []
        for wname, arr in ws.items():
            full = wname if "/" in wname else "%s/%s:0" % (lname, wname)
            names.append(full)
            parts = full.split("/")
            for i in range(1, len(parts)):
                w.create_group(g + "/" + "/".join(parts[:i]))
            w.create_dataset(g + "/" + full, np.asarray(arr))
        w.set_attr(g, "weight_names", names or [""])
    w.write(path)
