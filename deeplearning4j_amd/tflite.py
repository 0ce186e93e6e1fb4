"""TensorFlow Lite model execution on this framework's device tensors.

Reference: nd4j-tensorflow-lite (org/nd4j/tensorflowlite/runner/TensorFlowLiteRunner.java) hands the ``.tflite`` file to
the TFLite C runtime on the CPU. Here the FlatBuffer is decoded with the framework's own wire-format reader
(autodiff/flatbuf.py; field numbering of tensorflow/lite/schema/schema.fbs v3, written down below) and the operator list
of subgraph 0 — TFLite graphs are already in execution order — is run op by op on torch tensors of the chosen device,
so activations stay resident in HBM and convolutions / GEMMs go through the same MIOpen / hipBLASLt / in-tree kernels
as native models. Tensors are NHWC as in TFLite; convolutions run in channels_last memory format (no data transposes).

No ``.tflite`` fixture ships with the reference and the TFLite runtime is not installable here, so the tests build
models with the in-tree FlatBuffers writer and check them against a plain PyTorch computation: parity with the real
TFLite interpreter is unpinned.
"""
from __future__ import annotations

import struct
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .autodiff import flatbuf as fb

# TensorType -> (torch dtype, numpy dtype)
TENSOR_TYPES = {0: (torch.float32, "<f4"), 1: (torch.float16, "<f2"), 2: (torch.int32, "<i4"), 3: (torch.uint8, "u1"),
                4: (torch.int64, "<i8"), 6: (torch.bool, "?"), 7: (torch.int16, "<i2"), 9: (torch.int8, "i1"),
                10: (torch.float64, "<f8")}

# BuiltinOperator codes (schema.fbs)
BUILTIN = {0: "ADD", 1: "AVERAGE_POOL_2D", 2: "CONCATENATION", 3: "CONV_2D", 4: "DEPTHWISE_CONV_2D", 5: "DEPTH_TO_SPACE",
           6: "DEQUANTIZE", 8: "FLOOR", 9: "FULLY_CONNECTED", 11: "L2_NORMALIZATION", 14: "LOGISTIC", 17: "MAX_POOL_2D",
           18: "MUL", 19: "RELU", 20: "RELU_N1_TO_1", 21: "RELU6", 22: "RESHAPE", 23: "RESIZE_BILINEAR", 25: "SOFTMAX",
           26: "SPACE_TO_DEPTH", 28: "TANH", 34: "PAD", 36: "GATHER", 39: "TRANSPOSE", 40: "MEAN", 41: "SUB", 42: "DIV",
           43: "SQUEEZE", 45: "STRIDED_SLICE", 47: "EXP", 49: "SPLIT", 50: "LOG_SOFTMAX", 53: "CAST", 54: "PRELU",
           55: "MAXIMUM", 56: "ARG_MAX", 57: "MINIMUM", 58: "LESS", 59: "NEG", 61: "GREATER", 62: "GREATER_EQUAL",
           63: "LESS_EQUAL", 64: "SELECT", 65: "SLICE", 66: "SIN", 69: "TILE", 70: "EXPAND_DIMS", 71: "EQUAL", 72: "NOT_EQUAL",
           73: "LOG", 74: "SUM", 75: "SQRT", 76: "RSQRT", 77: "SHAPE", 78: "POW", 79: "ARG_MIN", 82: "REDUCE_MAX", 83: "PACK",
           88: "UNPACK", 89: "REDUCE_MIN", 92: "SQUARE", 97: "RESIZE_NEAREST_NEIGHBOR", 98: "LEAKY_RELU", 99: "SQUARED_DIFFERENCE",
           101: "ABS", 117: "HARD_SWISH"}
BUILTIN_CODE = {v: k for k, v in BUILTIN.items()}


def _act(x, code: int):
    """ActivationFunctionType: NONE, RELU, RELU_N1_TO_1, RELU6, TANH."""
    if code == 0:
        return x
    if code == 1:
        return F.relu(x)
    if code == 2:
        return x.clamp(-1.0, 1.0)
    if code == 3:
        return x.clamp(0.0, 6.0)
    if code == 4:
        return torch.tanh(x)
    raise NotImplementedError("fused activation %d" % code)


def _same_pad(size, k, stride, dil):
    out = -(-size // stride)
    total = max((out - 1) * stride + (k - 1) * dil + 1 - size, 0)
    return total // 2, total - total // 2


def _nchw(x):
    """NHWC tensor -> logical NCHW view in channels_last memory (no copy when x is contiguous NHWC)."""
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def _pad_hw(x_nchw, padding_same, kh, kw, sh, sw, dh, dw, value=0.0):
    if not padding_same:
        return x_nchw
    pt, pb = _same_pad(x_nchw.shape[2], kh, sh, dh)
    pl, pr = _same_pad(x_nchw.shape[3], kw, sw, dw)
    if pt or pb or pl or pr:
        x_nchw = F.pad(x_nchw, (pl, pr, pt, pb), value=value)
    return x_nchw


class _Op:
    __slots__ = ("name", "inputs", "outputs", "opt", "custom")

    def __init__(self, name, inputs, outputs, opt, custom):
        self.name, self.inputs, self.outputs, self.opt, self.custom = name, inputs, outputs, opt, custom


_KERNELS: Dict[str, Callable] = {}


def tflite_op(*names):
    def deco(fn):
        for n in names:
            _KERNELS[n] = fn
        return fn
    return deco


def _o(opt: Optional[fb.Table], field, kind, default=0):
    return default if opt is None else opt.scalar(field, kind, default)


@tflite_op("CONV_2D")
def _conv(op, x, w, b=None):
    same = _o(op.opt, 0, "byte") == 0
    sw, sh, act = _o(op.opt, 1, "int", 1), _o(op.opt, 2, "int", 1), _o(op.opt, 3, "byte")
    dw, dh = _o(op.opt, 4, "int", 1), _o(op.opt, 5, "int", 1)
    wt = w.permute(0, 3, 1, 2)                                     # OHWI -> OIHW view
    xi = _pad_hw(_nchw(x), same, w.shape[1], w.shape[2], sh, sw, dh, dw)
    y = F.conv2d(xi, wt, b, stride=(sh, sw), dilation=(dh, dw))
    return _act(_nhwc(y), act)


@tflite_op("DEPTHWISE_CONV_2D")
def _dwconv(op, x, w, b=None):
    same = _o(op.opt, 0, "byte") == 0
    sw, sh, act = _o(op.opt, 1, "int", 1), _o(op.opt, 2, "int", 1), _o(op.opt, 4, "byte")
    dw, dh = _o(op.opt, 5, "int", 1), _o(op.opt, 6, "int", 1)
    cin = x.shape[3]
    wt = w.permute(3, 0, 1, 2)                                     # [1,kh,kw,cin*mult] -> [cin*mult,1,kh,kw]
    xi = _pad_hw(_nchw(x), same, w.shape[1], w.shape[2], sh, sw, dh, dw)
    y = F.conv2d(xi, wt, b, stride=(sh, sw), dilation=(dh, dw), groups=cin)
    return _act(_nhwc(y), act)


def _pool(op, x, kind):
    same = _o(op.opt, 0, "byte") == 0
    sw, sh = _o(op.opt, 1, "int", 1), _o(op.opt, 2, "int", 1)
    kw, kh, act = _o(op.opt, 3, "int", 1), _o(op.opt, 4, "int", 1), _o(op.opt, 5, "byte")
    xi = _nchw(x)
    if kind == "max":
        y = F.max_pool2d(_pad_hw(xi, same, kh, kw, sh, sw, 1, 1, value=float("-inf")), (kh, kw), (sh, sw))
    else:
        if same:                                                    # TFLite averages over the valid elements only
            ones = torch.ones_like(xi[:1, :1])
            num = F.avg_pool2d(_pad_hw(xi, True, kh, kw, sh, sw, 1, 1), (kh, kw), (sh, sw))
            den = F.avg_pool2d(_pad_hw(ones, True, kh, kw, sh, sw, 1, 1), (kh, kw), (sh, sw))
            y = num / den
        else:
            y = F.avg_pool2d(xi, (kh, kw), (sh, sw))
    return _act(_nhwc(y), act)


@tflite_op("MAX_POOL_2D")
def _maxpool(op, x):
    return _pool(op, x, "max")


@tflite_op("AVERAGE_POOL_2D")
def _avgpool(op, x):
    return _pool(op, x, "avg")


@tflite_op("FULLY_CONNECTED")
def _fc(op, x, w, b=None):
    act, keep = _o(op.opt, 0, "byte"), bool(_o(op.opt, 2, "bool", False))
    xi = x if keep else x.reshape(-1, w.shape[1])
    return _act(F.linear(xi, w, b), act)


def _binary(fn):
    def k(op, a, b):
        return _act(fn(a, b), _o(op.opt, 0, "byte"))
    return k


_KERNELS["ADD"] = _binary(torch.add)
_KERNELS["SUB"] = _binary(torch.sub)
_KERNELS["MUL"] = _binary(torch.mul)
_KERNELS["DIV"] = _binary(lambda a, b: torch.div(a, b) if a.is_floating_point() else torch.div(a, b, rounding_mode="trunc"))
for _n, _f in {"MAXIMUM": torch.maximum, "MINIMUM": torch.minimum, "POW": torch.pow, "LESS": torch.lt, "GREATER": torch.gt,
               "GREATER_EQUAL": torch.ge, "LESS_EQUAL": torch.le, "EQUAL": torch.eq, "NOT_EQUAL": torch.ne,
               "SQUARED_DIFFERENCE": lambda a, b: (a - b) ** 2}.items():
    _KERNELS[_n] = (lambda f: lambda op, a, b: f(a, b))(_f)
for _n, _f in {"RELU": F.relu, "RELU6": lambda x: x.clamp(0, 6), "RELU_N1_TO_1": lambda x: x.clamp(-1, 1), "TANH": torch.tanh,
               "LOGISTIC": torch.sigmoid, "EXP": torch.exp, "LOG": torch.log, "SQRT": torch.sqrt, "RSQRT": torch.rsqrt, "NEG": torch.neg,
               "ABS": torch.abs, "SIN": torch.sin, "FLOOR": torch.floor, "SQUARE": torch.square, "HARD_SWISH": F.hardswish,
               "DEQUANTIZE": lambda x: x.float()}.items():
    _KERNELS[_n] = (lambda f: lambda op, x: f(x))(_f)


@tflite_op("LEAKY_RELU")
def _lrelu(op, x):
    return F.leaky_relu(x, _o(op.opt, 0, "float", 0.2))


@tflite_op("PRELU")
def _prelu(op, x, alpha):
    return torch.where(x >= 0, x, x * alpha)


@tflite_op("SOFTMAX")
def _softmax(op, x):
    return torch.softmax(x * _o(op.opt, 0, "float", 1.0), dim=-1)


@tflite_op("LOG_SOFTMAX")
def _log_softmax(op, x):
    return torch.log_softmax(x, dim=-1)


@tflite_op("L2_NORMALIZATION")
def _l2norm(op, x):
    return _act(F.normalize(x, dim=-1, eps=1e-6), _o(op.opt, 0, "byte"))


@tflite_op("RESHAPE")
def _reshape(op, x, shape=None):
    if shape is not None:
        new = [int(v) for v in shape.reshape(-1).tolist()]
    else:
        new = op.opt.scalar_vector(0, "int") if op.opt is not None else []
    return x.reshape(new)


@tflite_op("SQUEEZE")
def _squeeze(op, x):
    dims = op.opt.scalar_vector(0, "int") if op.opt is not None else []
    if not dims:
        return x.squeeze()
    return x.reshape([s for i, s in enumerate(x.shape) if not (i in [d % x.dim() for d in dims] and s == 1)])


@tflite_op("EXPAND_DIMS")
def _expand(op, x, axis):
    return x.unsqueeze(int(axis))


@tflite_op("CONCATENATION")
def _concat(op, *xs):
    return _act(torch.cat(xs, dim=_o(op.opt, 0, "int")), _o(op.opt, 1, "byte"))


@tflite_op("TRANSPOSE")
def _transpose(op, x, perm):
    return x.permute([int(p) for p in perm.tolist()])


@tflite_op("PAD")
def _pad(op, x, pads):
    p = pads.reshape(-1, 2).tolist()
    flat = []
    for before, after in reversed(p):
        flat += [int(before), int(after)]
    return F.pad(x, flat)


def _reduce(fn):
    def k(op, x, axes):
        dims = tuple(int(a) % x.dim() for a in axes.reshape(-1).tolist())
        return fn(x, dims, bool(_o(op.opt, 0, "bool", False)))
    return k


_KERNELS["MEAN"] = _reduce(lambda x, d, k: x.mean(dim=d, keepdim=k))
_KERNELS["SUM"] = _reduce(lambda x, d, k: x.sum(dim=d, keepdim=k))
_KERNELS["REDUCE_MAX"] = _reduce(lambda x, d, k: x.amax(dim=d, keepdim=k))
_KERNELS["REDUCE_MIN"] = _reduce(lambda x, d, k: x.amin(dim=d, keepdim=k))


@tflite_op("ARG_MAX")
def _argmax(op, x, axis):
    out = torch.argmax(x, dim=int(axis))
    return out.to(torch.int32) if _o(op.opt, 0, "byte", 2) == 2 else out


@tflite_op("ARG_MIN")
def _argmin(op, x, axis):
    out = torch.argmin(x, dim=int(axis))
    return out.to(torch.int32) if _o(op.opt, 0, "byte", 2) == 2 else out


@tflite_op("CAST")
def _cast(op, x):
    return x.to(TENSOR_TYPES[_o(op.opt, 1, "byte")][0]) if op.opt is not None and op.opt.has(1) else x


@tflite_op("GATHER")
def _gather(op, x, idx):
    axis = _o(op.opt, 0, "int")
    return torch.index_select(x, axis, idx.reshape(-1).long()).reshape(list(x.shape[:axis]) + list(idx.shape) + list(x.shape[axis + 1:]))


@tflite_op("SLICE")
def _slice(op, x, begin, size):
    sl = tuple(slice(int(b), None if int(s) < 0 else int(b) + int(s)) for b, s in zip(begin.tolist(), size.tolist()))
    return x[sl]


@tflite_op("TILE")
def _tile(op, x, mult):
    return x.repeat([int(m) for m in mult.tolist()])


@tflite_op("SELECT")
def _select(op, c, a, b):
    return torch.where(c, a, b)


@tflite_op("SHAPE")
def _shape(op, x):
    return torch.tensor(list(x.shape), dtype=torch.int32, device=x.device)


@tflite_op("PACK")
def _pack(op, *xs):
    return torch.stack(xs, dim=_o(op.opt, 1, "int"))


@tflite_op("UNPACK")
def _unpack(op, x):
    return list(torch.unbind(x, dim=_o(op.opt, 1, "int")))


@tflite_op("SPLIT")
def _split(op, axis, x):
    return list(torch.chunk(x, _o(op.opt, 0, "int", 1), dim=int(axis)))


@tflite_op("SPACE_TO_DEPTH")
def _s2d(op, x):
    return _space_to_depth(x, _o(op.opt, 0, "int", 1))


def _space_to_depth(x, bs):
    n, h, w, c = x.shape
    return x.reshape(n, h // bs, bs, w // bs, bs, c).permute(0, 1, 3, 2, 4, 5).reshape(n, h // bs, w // bs, bs * bs * c)


@tflite_op("DEPTH_TO_SPACE")
def _d2s(op, x):
    bs = _o(op.opt, 0, "int", 1)
    n, h, w, c = x.shape
    return x.reshape(n, h, w, bs, bs, c // (bs * bs)).permute(0, 1, 3, 2, 4, 5).reshape(n, h * bs, w * bs, c // (bs * bs))


def _resize(mode):
    def k(op, x, size):
        align = bool(_o(op.opt, 2, "bool", False))
        oh, ow = [int(v) for v in size.tolist()]
        if mode == "nearest":
            y = F.interpolate(_nchw(x), size=(oh, ow), mode="nearest")
        else:
            half = bool(_o(op.opt, 3, "bool", False))
            if not align and not half:                              # TFLite legacy: src = dst * scale (no half-pixel shift)
                ih, iw = x.shape[1], x.shape[2]
                ys = torch.arange(oh, device=x.device, dtype=torch.float32) * (ih / oh)
                xs = torch.arange(ow, device=x.device, dtype=torch.float32) * (iw / ow)
                y0, x0 = ys.floor().long().clamp(max=ih - 1), xs.floor().long().clamp(max=iw - 1)
                y1, x1 = (y0 + 1).clamp(max=ih - 1), (x0 + 1).clamp(max=iw - 1)
                wy, wx = (ys - y0).view(1, oh, 1, 1), (xs - x0).view(1, 1, ow, 1)
                top = x[:, y0][:, :, x0] * (1 - wx) + x[:, y0][:, :, x1] * wx
                bot = x[:, y1][:, :, x0] * (1 - wx) + x[:, y1][:, :, x1] * wx
                return top * (1 - wy) + bot * wy
            y = F.interpolate(_nchw(x), size=(oh, ow), mode="bilinear", align_corners=align)
        return _nhwc(y)
    return k


_KERNELS["RESIZE_BILINEAR"] = _resize("bilinear")
_KERNELS["RESIZE_NEAREST_NEIGHBOR"] = _resize("nearest")


@tflite_op("STRIDED_SLICE")
def _strided_slice(op, x, begin, end, strides):
    bm, em = _o(op.opt, 0, "int"), _o(op.opt, 1, "int")
    ell, new, shrink = _o(op.opt, 2, "int"), _o(op.opt, 3, "int"), _o(op.opt, 4, "int")
    if ell or new:
        raise NotImplementedError("STRIDED_SLICE with ellipsis/new_axis masks")
    idx = []
    for d, (b, e, s) in enumerate(zip(begin.tolist(), end.tolist(), strides.tolist())):
        if s <= 0:
            raise NotImplementedError("STRIDED_SLICE with non-positive stride")
        if shrink >> d & 1:
            idx.append(int(b))
            continue
        idx.append(slice(None if bm >> d & 1 else int(b), None if em >> d & 1 else int(e), int(s)))
    return x[tuple(idx)]


class TFLiteModel:
    """Decoded ``.tflite`` file: tensors, constant buffers and the operator list of one subgraph."""

    def __init__(self, path_or_bytes, subgraph: int = 0):
        buf = bytes(path_or_bytes) if isinstance(path_or_bytes, (bytes, bytearray)) else open(str(path_or_bytes), "rb").read()
        if len(buf) < 8:
            raise ValueError("not a TFLite FlatBuffer (too short)")
        if buf[4:8] not in (b"TFL3", b"\0\0\0\0") and struct.unpack_from("<I", buf, 0)[0] >= len(buf):
            raise ValueError("not a TFLite FlatBuffer")
        m = fb.root(buf)
        self.version = m.scalar(0, "uint")
        self.description = m.string(3, "")
        codes = []
        for oc in m.table_vector(1):
            code = max(oc.scalar(0, "byte"), oc.scalar(3, "int"))      # builtin_code supersedes the deprecated byte field
            codes.append(("CUSTOM:" + (oc.string(1) or "")) if code == 32 else BUILTIN.get(code, "BUILTIN_%d" % code))
        self._buffers = m.table_vector(4)
        graphs = m.table_vector(2)
        if not 0 <= subgraph < len(graphs):
            raise ValueError("subgraph %d out of range (%d subgraphs)" % (subgraph, len(graphs)))
        g = graphs[subgraph]
        self.name = g.string(4, "")
        self.tensors = []
        for t in g.table_vector(0):
            tt = t.scalar(1, "byte")
            if tt not in TENSOR_TYPES:
                raise NotImplementedError("TFLite tensor type %d" % tt)
            q = t.table(4)
            quant = None
            if q is not None and q.vector_len(2):
                quant = (q.scalar_vector(2, "float"), q.scalar_vector(3, "long"), q.scalar(6, "int"))
            self.tensors.append({"shape": t.scalar_vector(0, "int"), "type": tt, "buffer": t.scalar(2, "uint"),
                                 "name": t.string(3, ""), "quant": quant})
        self.inputs = g.scalar_vector(1, "int")
        self.outputs = g.scalar_vector(2, "int")
        self.operators: List[_Op] = []
        for o in g.table_vector(3):
            ci = o.scalar(0, "uint")
            if ci >= len(codes):
                raise ValueError("operator refers to opcode %d of %d" % (ci, len(codes)))
            self.operators.append(_Op(codes[ci], o.scalar_vector(1, "int"), o.scalar_vector(2, "int"), o.table(4), o.bytes_vector(5)))
        nt = len(self.tensors)
        for op in self.operators:
            for i in op.inputs + op.outputs:
                if not -1 <= i < nt:
                    raise ValueError("operator %s refers to tensor %d of %d" % (op.name, i, nt))

    def constant(self, idx: int, device=None) -> Optional[torch.Tensor]:
        t = self.tensors[idx]
        if t["buffer"] == 0 or t["buffer"] >= len(self._buffers):
            return None
        data = self._buffers[t["buffer"]].bytes_vector(0)
        if not data:
            return None
        dt, npdt = TENSOR_TYPES[t["type"]]
        n = 1
        for s in t["shape"]:
            n *= s
        if len(data) != n * np.dtype(npdt).itemsize:
            raise ValueError("tensor '%s': buffer holds %d bytes, shape %s needs %d" % (t["name"], len(data), t["shape"], n * np.dtype(npdt).itemsize))
        arr = np.frombuffer(data, dtype=npdt).reshape(t["shape"]).copy()
        out = torch.from_numpy(arr).to(device)
        if t["quant"] is not None and t["type"] in (3, 9, 2):              # weights stored quantized: dequantize once at load
            scale, zp, axis = t["quant"]
            sc = torch.tensor(scale, device=device)
            z = torch.tensor(zp or [0] * len(scale), device=device, dtype=torch.float32)
            if len(scale) > 1:
                view = [1] * out.dim()
                view[axis] = -1
                sc, z = sc.view(view), z.view(view)
            out = (out.float() - z) * sc
        return out

    def opNames(self) -> List[str]:
        return [o.name for o in self.operators]


class TensorFlowLiteRunner:
    """``TensorFlowLiteRunner(modelUri).exec({input name: tensor}) -> {output name: tensor}``."""

    class Builder:
        def __init__(self):
            self._uri, self._device = None, None

        def modelUri(self, uri):
            self._uri = uri
            return self

        def device(self, d):
            self._device = d
            return self

        def build(self):
            return TensorFlowLiteRunner(self._uri, self._device)

    @staticmethod
    def builder():
        return TensorFlowLiteRunner.Builder()

    def __init__(self, modelUri, device=None):
        self.device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.model = TFLiteModel(modelUri)
        missing = sorted({o.name for o in self.model.operators if o.name not in _KERNELS})
        if missing:
            raise NotImplementedError("TFLite operator(s) without a kernel: %s" % missing)
        self._consts = {i: c for i in range(len(self.model.tensors)) for c in [self.model.constant(i, self.device)] if c is not None}
        self.inputNames = [self.model.tensors[i]["name"] for i in self.model.inputs]
        self.outputNames = [self.model.tensors[i]["name"] for i in self.model.outputs]

    def getInputNames(self):
        return list(self.inputNames)

    def getOutputNames(self):
        return list(self.outputNames)

    @torch.no_grad()
    def exec(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        m = self.model
        vals: Dict[int, torch.Tensor] = dict(self._consts)
        for idx, name in zip(m.inputs, self.inputNames):
            if name not in inputs:
                raise ValueError("Missing input '%s'; model inputs are %s" % (name, self.inputNames))
            v = inputs[name]
            v = getattr(v, "_t", v)
            v = torch.as_tensor(v).to(self.device, TENSOR_TYPES[m.tensors[idx]["type"]][0])
            want = m.tensors[idx]["shape"]
            if len(want) != v.dim() or any(w not in (-1, s) and i > 0 for i, (w, s) in enumerate(zip(want, v.shape))):
                raise ValueError("input '%s' has shape %s, model expects %s (batch dimension free)" % (name, list(v.shape), want))
            vals[idx] = v
        for op in m.operators:
            args = []
            for i in op.inputs:
                if i < 0:
                    args.append(None)                              # optional input left out (e.g. no bias)
                elif i not in vals:
                    raise RuntimeError("operator %s reads tensor '%s' before it is produced" % (op.name, m.tensors[i]["name"]))
                else:
                    args.append(vals[i])
            while args and args[-1] is None:
                args.pop()
            out = _KERNELS[op.name](op, *args)
            outs = out if isinstance(out, (list, tuple)) else [out]
            if len(outs) != len(op.outputs):
                raise RuntimeError("operator %s produced %d outputs, model declares %d" % (op.name, len(outs), len(op.outputs)))
            for i, o in zip(op.outputs, outs):
                vals[i] = o
        return {n: vals[i] for i, n in zip(m.outputs, self.outputNames)}

    run = exec

    def close(self):
        self._consts = {}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False


# ------------------------------------------------------------------------------------------------ writer (tests, export)
class TFLiteBuilder:
    """Assembles a ``.tflite`` FlatBuffer (float models): ``t = b.tensor(...)``, ``b.op("CONV_2D", [x, w, bias], [y],
    options={field: S.Scalar(...)})``, ``b.finish([inputs], [outputs])``."""

    def __init__(self, description="deeplearning4j_amd"):
        self.description = description
        self._buffers: List[bytes] = [b""]                          # buffer 0 is the empty sentinel
        self._tensors: List[fb.S.Tbl] = []
        self._ops: List[fb.S.Tbl] = []
        self._codes: List[str] = []

    def tensor(self, name: str, shape: Sequence[int], data: Optional[torch.Tensor] = None, ttype: int = 0) -> int:
        buf = 0
        if data is not None:
            npdt = TENSOR_TYPES[ttype][1]
            self._buffers.append(np.ascontiguousarray(data.detach().cpu().numpy().astype(npdt)).tobytes())
            buf = len(self._buffers) - 1
        self._tensors.append(fb.S.Tbl({0: fb.S.Vec("int", list(shape)), 1: fb.S.Scalar("byte", ttype), 2: fb.S.Scalar("uint", buf),
                                       3: fb.S.Str(name)}))
        return len(self._tensors) - 1

    def op(self, name: str, inputs: Sequence[int], outputs: Sequence[int], options: Optional[Dict[int, object]] = None, optionsType: int = 0):
        if name not in self._codes:
            self._codes.append(name)
        f = {0: fb.S.Scalar("uint", self._codes.index(name)), 1: fb.S.Vec("int", list(inputs)), 2: fb.S.Vec("int", list(outputs))}
        if options:
            f[3] = fb.S.Scalar("ubyte", optionsType or 1)
            f[4] = fb.S.Tbl(dict(options))
        self._ops.append(fb.S.Tbl(f))

    def finish(self, inputs: Sequence[int], outputs: Sequence[int]) -> bytes:
        codes = []
        for n in self._codes:
            c = BUILTIN_CODE[n]
            codes.append(fb.S.Tbl({0: fb.S.Scalar("byte", min(c, 127)), 2: fb.S.Scalar("int", 1), 3: fb.S.Scalar("int", c)}))
        sub = fb.S.Tbl({0: fb.S.Vec("table", self._tensors), 1: fb.S.Vec("int", list(inputs)), 2: fb.S.Vec("int", list(outputs)),
                        3: fb.S.Vec("table", self._ops), 4: fb.S.Str("main")})
        bufs = [fb.S.Tbl({0: fb.S.Bytes(b)} if b else {}) for b in self._buffers]
        model = fb.S.Tbl({0: fb.S.Scalar("uint", 3), 1: fb.S.Vec("table", codes), 2: fb.S.Vec("table", [sub]),
                          3: fb.S.Str(self.description), 4: fb.S.Vec("table", bufs)})
        return fb.Writer().finish(model)
