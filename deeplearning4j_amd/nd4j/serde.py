"""Binary array encoding of Nd4j.write / Nd4j.read (the coefficients.bin / updaterState.bin format).

Reference: nd4j-api/.../linalg/factory/Nd4j.java:2748-2791 (write = shapeInfo buffer then data
buffer) and nd4j-api/.../linalg/api/buffer/BaseDataBuffer.java:1852-1912,2060-2100 (per buffer:
writeUTF(allocationMode), writeLong(length), writeUTF(dataType), big-endian elements; legacy
allocation modes stored the length as int). shapeInfo layout + dtype flags: SURVEY Appendix B.1.
"""
from __future__ import annotations

import io
import struct

import numpy as np
import torch

# ArrayOptions dtype flags (libnd4j/include/array/ArrayOptions.h:36-134)
_FLAG = {"BFLOAT16": 2048, "HALF": 4096, "FLOAT": 8192, "DOUBLE": 16384, "BYTE": 32768, "SHORT": 65536, "INT": 131072,
         "LONG": 262144, "BOOL": 524288, "UBYTE": 32768 | 8388608, "UINT16": 65536 | 8388608, "UINT32": 131072 | 8388608,
         "UINT64": 262144 | 8388608}
_NP = {"DOUBLE": ">f8", "FLOAT": ">f4", "HALF": ">f2", "BFLOAT16": ">u2", "LONG": ">i8", "INT": ">i4", "SHORT": ">i2",
       "BYTE": "i1", "UBYTE": "u1", "BOOL": "u1", "UINT16": ">u2", "UINT32": ">u4", "UINT64": ">u8"}
_TORCH2NAME = {torch.float64: "DOUBLE", torch.float32: "FLOAT", torch.float16: "HALF", torch.bfloat16: "BFLOAT16",
               torch.int64: "LONG", torch.int32: "INT", torch.int16: "SHORT", torch.int8: "BYTE", torch.uint8: "UBYTE",
               torch.bool: "BOOL"}
_NAME2TORCH = {v: k for k, v in _TORCH2NAME.items()}


def _write_utf(out, s):
    b = s.encode("utf-8")
    out.write(struct.pack(">H", len(b)))
    out.write(b)


def _read_utf(inp):
    (n,) = struct.unpack(">H", inp.read(2))
    return inp.read(n).decode("utf-8")


def shape_info(shape, order="c", dtype_name="FLOAT", strides=None, empty=False):
    """LongType[2*rank+4] = [rank, shape.., strides.., extras, ews, order]."""
    rank = len(shape)
    if strides is None:
        strides, s = [0] * rank, 1
        rng = range(rank - 1, -1, -1) if order == "c" else range(rank)
        for d in rng:
            strides[d] = s
            s *= shape[d]
    extras = _FLAG[dtype_name] | (8 if empty else 0)
    return [rank] + list(shape) + list(strides) + [extras, 1, ord(order)]


def parse_shape_info(si):
    rank = int(si[0])
    shape = [int(x) for x in si[1:1 + rank]]
    strides = [int(x) for x in si[1 + rank:1 + 2 * rank]]
    extras, ews, order = int(si[1 + 2 * rank]), int(si[2 + 2 * rank]), chr(int(si[3 + 2 * rank]))
    dt = None
    for name in ("DOUBLE", "FLOAT", "HALF", "BFLOAT16", "LONG", "INT", "SHORT", "BOOL", "BYTE"):
        if extras & _FLAG[name]:
            dt = name
            break
    if dt in ("BYTE", "SHORT", "INT", "LONG") and extras & 8388608:
        dt = {"BYTE": "UBYTE", "SHORT": "UINT16", "INT": "UINT32", "LONG": "UINT64"}[dt]
    return shape, strides, order, dt, bool(extras & 8)


def write_buffer(out, np_array, dtype_name, allocation_mode="MIXED_DATA_TYPES"):
    _write_utf(out, allocation_mode)
    out.write(struct.pack(">q", int(np_array.size)))
    _write_utf(out, dtype_name)
    out.write(np.ascontiguousarray(np_array).astype(_NP[dtype_name], copy=False).tobytes())


def read_buffer(inp):
    mode = _read_utf(inp)
    if mode in ("DIRECT", "HEAP", "JAVACPP"):      # legacy: int length
        (n,) = struct.unpack(">i", inp.read(4))
    else:
        (n,) = struct.unpack(">q", inp.read(8))
    dt = _read_utf(inp)
    if dt == "COMPRESSED":
        raise ValueError("compressed buffers are not supported")
    npdt = np.dtype(_NP[dt])
    data = np.frombuffer(inp.read(n * npdt.itemsize), dtype=npdt, count=n)
    return data, dt


def write_array(t: torch.Tensor, out, order="c"):
    """Nd4j.write(arr, dos): views are dup'ed; 'c' order data."""
    t = t.detach().cpu()
    name = _TORCH2NAME[t.dtype]
    shape = list(t.shape) if t.dim() > 0 else []
    if order == "f" and t.dim() > 1:
        flat = t.permute(*reversed(range(t.dim()))).contiguous().reshape(-1)
    else:
        flat = t.contiguous().reshape(-1)
    si = np.asarray(shape_info(shape, order, name), dtype=np.int64)
    write_buffer(out, si, "LONG")
    if name == "BFLOAT16":
        arr = flat.view(torch.int16).numpy().view(np.uint16)
    elif name == "BOOL":
        arr = flat.to(torch.uint8).numpy()
    else:
        arr = flat.numpy()
    write_buffer(out, arr, name)


def read_array(inp) -> torch.Tensor:
    si, sdt = read_buffer(inp)
    shape, strides, order, dt_hint, empty = parse_shape_info(si.astype(np.int64))
    data, dt = read_buffer(inp)
    name = dt
    if name == "BFLOAT16":
        t = torch.from_numpy(data.astype(np.uint16).view(np.int16).copy()).view(torch.bfloat16)
    elif name == "BOOL":
        t = torch.from_numpy(data.astype(np.uint8).copy()).bool()
    else:
        native = data.astype(data.dtype.newbyteorder("="), copy=True)
        t = torch.from_numpy(native)
    if empty:
        return t.reshape([0] * max(len(shape), 1))
    if order == "f" and len(shape) > 1:
        return t.reshape(list(reversed(shape))).permute(*reversed(range(len(shape))))
    return t.reshape(shape)


def to_bytes(t, order="c"):
    b = io.BytesIO()
    write_array(t, b, order)
    return b.getvalue()


def from_bytes(b):
    return read_array(io.BytesIO(b))
