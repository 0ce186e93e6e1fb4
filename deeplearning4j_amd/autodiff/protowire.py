"""Protocol-buffers wire format (decode + encode) without generated classes.

TensorFlow GraphDef and ONNX ModelProto are plain protobuf messages; neither ``tensorflow`` nor
``onnx`` is installed in this image, so the importers read the wire format directly with the field
numbers of the public schemas (tensorflow/core/framework/{graph,node_def,attr_value,tensor,
tensor_shape,types}.proto, onnx/onnx.proto; the reference vendors the same schemas under
nd4j-api/src/main/resources/onnx.pbtxt and nd4j/samediff-import).
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

VARINT, I64, LEN, I32 = 0, 1, 2, 5


def _varint(buf, pos):
    res, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        res |= (b & 0x7F) << shift
        if not b & 0x80:
            return res, pos
        shift += 7


def decode(buf: bytes) -> Dict[int, List[Tuple[int, object]]]:
    """field number -> [(wire type, raw value)]; LEN values stay bytes (decode nested messages on demand)."""
    out: Dict[int, List[Tuple[int, object]]] = {}
    pos, n = 0, len(buf)
    mv = memoryview(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == VARINT:
            v, pos = _varint(buf, pos)
        elif wt == I64:
            v = bytes(mv[pos:pos + 8])
            pos += 8
        elif wt == LEN:
            ln, pos = _varint(buf, pos)
            v = bytes(mv[pos:pos + ln])
            pos += ln
        elif wt == I32:
            v = bytes(mv[pos:pos + 4])
            pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        out.setdefault(field, []).append((wt, v))
    return out


def signed(v: int, bits=64) -> int:
    return v - (1 << bits) if v >= 1 << (bits - 1) else v


class Msg:
    """Thin typed accessors over a decoded message."""

    def __init__(self, buf: bytes):
        self.f = decode(buf)

    def has(self, n):
        return n in self.f

    def int(self, n, default=0):
        return signed(self.f[n][-1][1]) if n in self.f else default

    def ints(self, n) -> List[int]:
        out = []
        for wt, v in self.f.get(n, []):
            if wt == VARINT:
                out.append(signed(v))
            else:                                   # packed
                pos = 0
                while pos < len(v):
                    x, pos = _varint(v, pos)
                    out.append(signed(x))
        return out

    def float(self, n, default=0.0):
        if n not in self.f:
            return default
        wt, v = self.f[n][-1]
        return struct.unpack("<f", v)[0] if wt == I32 else struct.unpack("<d", v)[0]

    def floats(self, n, double=False) -> List[float]:
        out = []
        fmt, sz = ("<d", 8) if double else ("<f", 4)
        for wt, v in self.f.get(n, []):
            if wt in (I32, I64):
                out.append(struct.unpack(fmt, v)[0])
            else:
                out += [x[0] for x in struct.iter_unpack(fmt, v)]
        return out

    def bytes(self, n, default=b""):
        return self.f[n][-1][1] if n in self.f else default

    def str(self, n, default=""):
        return self.f[n][-1][1].decode("utf-8") if n in self.f else default

    def strs(self, n) -> List[str]:
        return [v.decode("utf-8", "replace") for _, v in self.f.get(n, [])]

    def msg(self, n):
        return Msg(self.f[n][-1][1]) if n in self.f else None

    def msgs(self, n) -> List["Msg"]:
        return [Msg(v) for _, v in self.f.get(n, [])]


# ----------------------------------------------------------------------------- encoder (tests / export)
def enc_varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def enc_field(field: int, value) -> bytes:
    """int -> varint; float -> fixed32; bytes/str -> length-delimited; list of the former -> repeated."""
    if isinstance(value, (list, tuple)):
        return b"".join(enc_field(field, v) for v in value)
    if isinstance(value, bool):
        return enc_varint(field << 3 | VARINT) + enc_varint(int(value))
    if isinstance(value, int):
        return enc_varint(field << 3 | VARINT) + enc_varint(value)
    if isinstance(value, float):
        return enc_varint(field << 3 | I32) + struct.pack("<f", value)
    if isinstance(value, str):
        value = value.encode("utf-8")
    return enc_varint(field << 3 | LEN) + enc_varint(len(value)) + bytes(value)


def enc_msg(*fields) -> bytes:
    """fields: (number, value) pairs, value as accepted by enc_field (nested messages pre-encoded to bytes)."""
    return b"".join(enc_field(n, v) for n, v in fields)
