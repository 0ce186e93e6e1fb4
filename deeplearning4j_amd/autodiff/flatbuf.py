"""Minimal FlatBuffers wire-format reader / writer (no external dependency).

The SameDiff ``.fb`` checkpoint format is a FlatBuffers ``FlatGraph`` (reference schemas:
libnd4j/include/graph/scheme/*.fbs; ADR 0001-SameDiff_File_Format). The flatbuffers runtime is not
available in this image, so the wire format is implemented here directly:

* buffer = uoffset32 to the root table
* table  = soffset32 to its vtable, then inline fields; vtable = [u16 vtable bytes, u16 table bytes,
  u16 field offset ...] (0 = field absent)
* strings / vectors / sub-tables are referenced by uoffset32 *relative to the referring location*
  and therefore always live at higher addresses — the writer lays objects out front-to-back:
  parent first, children appended afterwards and patched in.
"""
from __future__ import annotations

import struct
from typing import Any, Dict, List, Optional, Sequence, Tuple

_FMT = {"bool": ("<?", 1), "byte": ("<b", 1), "ubyte": ("<B", 1), "short": ("<h", 2), "ushort": ("<H", 2), "int": ("<i", 4),
        "uint": ("<I", 4), "long": ("<q", 8), "ulong": ("<Q", 8), "float": ("<f", 4), "double": ("<d", 8)}


# ----------------------------------------------------------------------------- reader
class Table:
    def __init__(self, buf: bytes, pos: int):
        self.buf, self.pos = buf, pos
        self.vt = pos - struct.unpack_from("<i", buf, pos)[0]
        self.vt_len = struct.unpack_from("<H", buf, self.vt)[0]

    def _off(self, field: int) -> int:
        o = 4 + 2 * field
        if o >= self.vt_len:
            return 0
        return struct.unpack_from("<H", self.buf, self.vt + o)[0]

    def has(self, field):
        return self._off(field) != 0

    def scalar(self, field, kind, default=0):
        o = self._off(field)
        if not o:
            return default
        return struct.unpack_from(_FMT[kind][0], self.buf, self.pos + o)[0]

    def _indirect(self, field) -> Optional[int]:
        o = self._off(field)
        if not o:
            return None
        p = self.pos + o
        return p + struct.unpack_from("<I", self.buf, p)[0]

    def string(self, field, default=None):
        p = self._indirect(field)
        if p is None:
            return default
        n = struct.unpack_from("<I", self.buf, p)[0]
        return self.buf[p + 4:p + 4 + n].decode("utf-8")

    def table(self, field) -> Optional["Table"]:
        p = self._indirect(field)
        return None if p is None else Table(self.buf, p)

    def vector_len(self, field):
        p = self._indirect(field)
        return 0 if p is None else struct.unpack_from("<I", self.buf, p)[0]

    def scalar_vector(self, field, kind) -> list:
        p = self._indirect(field)
        if p is None:
            return []
        n = struct.unpack_from("<I", self.buf, p)[0]
        fmt, sz = _FMT[kind]
        return list(struct.unpack_from("<%d%s" % (n, fmt[1]), self.buf, p + 4)) if n else []

    def bytes_vector(self, field) -> bytes:
        p = self._indirect(field)
        if p is None:
            return b""
        n = struct.unpack_from("<I", self.buf, p)[0]
        return bytes(self.buf[p + 4:p + 4 + n])

    def table_vector(self, field) -> List["Table"]:
        p = self._indirect(field)
        if p is None:
            return []
        n = struct.unpack_from("<I", self.buf, p)[0]
        out = []
        for k in range(n):
            e = p + 4 + 4 * k
            out.append(Table(self.buf, e + struct.unpack_from("<I", self.buf, e)[0]))
        return out

    def string_vector(self, field) -> List[str]:
        p = self._indirect(field)
        if p is None:
            return []
        n = struct.unpack_from("<I", self.buf, p)[0]
        out = []
        for k in range(n):
            e = p + 4 + 4 * k
            s = e + struct.unpack_from("<I", self.buf, e)[0]
            ln = struct.unpack_from("<I", self.buf, s)[0]
            out.append(self.buf[s + 4:s + 4 + ln].decode("utf-8"))
        return out


def root(buf: bytes) -> Table:
    return Table(buf, struct.unpack_from("<I", buf, 0)[0])


# ----------------------------------------------------------------------------- writer
class S:
    """Field value wrappers for the writer."""

    class Scalar:
        def __init__(self, kind, v):
            self.kind, self.v = kind, v

    class Str:
        def __init__(self, s):
            self.s = s

    class Vec:
        def __init__(self, kind, items):
            self.kind, self.items = kind, list(items)        # scalar kind | "string" | "table"

    class Bytes:
        def __init__(self, b):
            self.b = bytes(b)

    class Tbl:
        def __init__(self, fields: Dict[int, Any]):
            self.fields = fields


class Writer:
    def __init__(self):
        self.buf = bytearray(b"\0\0\0\0")          # root uoffset placeholder

    def _align(self, n):
        while len(self.buf) % n:
            self.buf.append(0)

    def _patch_u32(self, at, target):
        struct.pack_into("<I", self.buf, at, target - at)

    def finish(self, root_tbl: "S.Tbl") -> bytes:
        pos = self._table(root_tbl)
        struct.pack_into("<I", self.buf, 0, pos)
        return bytes(self.buf)

    def _string(self, s: str) -> int:
        b = s.encode("utf-8")
        self._align(4)
        pos = len(self.buf)
        self.buf += struct.pack("<I", len(b)) + b + b"\0"
        return pos

    def _vector(self, v: "S.Vec") -> int:
        n = len(v.items)
        if v.kind in _FMT:
            fmt, sz = _FMT[v.kind]
            # the element block must be aligned to the element size; the length prefix sits right before it
            while (len(self.buf) + 4) % max(sz, 4):
                self.buf.append(0)
            pos = len(self.buf)
            self.buf += struct.pack("<I", n)
            if n:
                self.buf += struct.pack("<%d%s" % (n, fmt[1]), *v.items)
            return pos
        self._align(4)
        pos = len(self.buf)
        self.buf += struct.pack("<I", n) + b"\0" * (4 * n)
        for k, it in enumerate(v.items):
            tgt = self._string(it) if v.kind == "string" else self._table(it)
            self._patch_u32(pos + 4 + 4 * k, tgt)
        return pos

    def _bytes(self, b: "S.Bytes") -> int:
        while (len(self.buf) + 4) % 8:                 # keep payloads 8-byte aligned (tensor data)
            self.buf.append(0)
        pos = len(self.buf)
        self.buf += struct.pack("<I", len(b.b)) + b.b
        return pos

    def _table(self, t: "S.Tbl") -> int:
        fields = {k: v for k, v in t.fields.items() if v is not None}
        nfields = (max(fields) + 1) if fields else 0
        # inline layout: 4-byte soffset, then fields sorted by size (largest first) with natural alignment
        layout, cur = {}, 4
        order = sorted(fields, key=lambda k: -(_FMT[fields[k].kind][1] if isinstance(fields[k], S.Scalar) else 4))
        for k in order:
            sz = _FMT[fields[k].kind][1] if isinstance(fields[k], S.Scalar) else 4
            cur = (cur + sz - 1) // sz * sz
            layout[k] = (cur, sz)
            cur += sz
        table_size = cur
        vt_size = 4 + 2 * nfields
        self._align(2)
        # place vtable, then the table 8-aligned (tables may hold 8-byte scalars at 8-aligned inline offsets)
        vt_pos = len(self.buf)
        self.buf += b"\0" * vt_size
        while (len(self.buf) + 4) % 8 != 4 and len(self.buf) % 8:
            self.buf.append(0)
        self._align(8)
        tpos = len(self.buf)
        self.buf += b"\0" * table_size
        struct.pack_into("<HH", self.buf, vt_pos, vt_size, table_size)
        for k in range(nfields):
            struct.pack_into("<H", self.buf, vt_pos + 4 + 2 * k, layout[k][0] if k in layout else 0)
        struct.pack_into("<i", self.buf, tpos, tpos - vt_pos)
        for k, v in fields.items():
            at = tpos + layout[k][0]
            if isinstance(v, S.Scalar):
                struct.pack_into(_FMT[v.kind][0], self.buf, at, v.v)
            elif isinstance(v, S.Str):
                self._patch_u32(at, self._string(v.s))
            elif isinstance(v, S.Vec):
                self._patch_u32(at, self._vector(v))
            elif isinstance(v, S.Bytes):
                self._patch_u32(at, self._bytes(v))
            elif isinstance(v, S.Tbl):
                self._patch_u32(at, self._table(v))
            else:
                raise TypeError(type(v))
        return tpos
