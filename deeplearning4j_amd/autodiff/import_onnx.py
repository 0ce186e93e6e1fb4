"""ONNX ModelProto -> SameDiff importer.

Reference: nd4j/samediff-import/samediff-import-onnx (OnnxImportGraph, per-op mapping rules, IR),
schema onnx/onnx.proto (vendored by the reference as nd4j-api/src/main/resources/onnx.pbtxt).
Mapping rules are Python functions registered with ``@onnx_op``; see import_tf.py for the pattern.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from .protowire import Msg
from .samediff import SameDiff, SDVariable

ONNX_DT = {1: torch.float32, 2: torch.uint8, 3: torch.int8, 5: torch.int16, 6: torch.int32, 7: torch.int64, 9: torch.bool, 10: torch.float16,
           11: torch.float64, 16: torch.bfloat16, 4: torch.int32, 12: torch.int64, 13: torch.int64}
_NP = {1: "<f4", 2: "u1", 3: "i1", 5: "<i2", 6: "<i4", 7: "<i8", 9: "u1", 10: "<f2", 11: "<f8", 16: "<u2", 4: "<u2", 12: "<u4", 13: "<u8"}
_DTN = {torch.float32: "FLOAT", torch.float64: "DOUBLE", torch.int32: "INT", torch.int64: "LONG", torch.bool: "BOOL", torch.uint8: "UBYTE",
        torch.int16: "SHORT", torch.int8: "BYTE", torch.float16: "HALF", torch.bfloat16: "BFLOAT16"}


def parse_tensor(m: Msg) -> torch.Tensor:
    dims = m.ints(1)
    dt = m.int(2)
    td = ONNX_DT[dt]
    raw = m.bytes(9)
    if raw:
        arr = np.frombuffer(raw, dtype=np.dtype(_NP[dt])).copy()
        if dt == 16:
            t = torch.from_numpy(arr.view(np.int16)).view(torch.bfloat16)
        elif dt in (4, 12, 13):
            t = torch.from_numpy(arr.astype(np.int64)).to(td)
        else:
            t = torch.from_numpy(arr)
            t = t.bool() if dt == 9 else t
    elif dt == 1:
        t = torch.tensor(m.floats(4), dtype=torch.float32)
    elif dt == 11:
        t = torch.tensor(m.floats(10, True), dtype=torch.float64)
    elif dt == 7:
        t = torch.tensor(m.ints(7), dtype=torch.int64)
    elif dt == 10:
        t = torch.from_numpy(np.asarray(m.ints(5), dtype=np.uint16).view(np.float16).copy())
    else:
        t = torch.tensor(m.ints(5)).to(td)
    return t.reshape(dims).to(td)


class OnnxNode:
    def __init__(self, m: Msg):
        self.inputs, self.outputs = m.strs(1), m.strs(2)
        self.name, self.op = m.str(3), m.str(4)
        self.domain = m.str(7)
        self.attrs: Dict[str, Msg] = {a.str(1): a for a in m.msgs(5)}
        if not self.name:
            self.name = (self.outputs[0] if self.outputs else self.op)

    def a_int(self, k, default=0):
        return self.attrs[k].int(3, default) if k in self.attrs else default

    def a_float(self, k, default=0.0):
        return self.attrs[k].float(2, default) if k in self.attrs else default

    def a_ints(self, k, default=None):
        return self.attrs[k].ints(8) if k in self.attrs else (default or [])

    def a_floats(self, k):
        return self.attrs[k].floats(7) if k in self.attrs else []

    def a_str(self, k, default=""):
        return self.attrs[k].bytes(4).decode() if k in self.attrs else default

    def a_tensor(self, k):
        return parse_tensor(self.attrs[k].msg(5)) if k in self.attrs and self.attrs[k].has(5) else None


_RULES: Dict[str, Callable] = {}


def onnx_op(*names):
    def deco(fn):
        for n in names:
            _RULES[n] = fn
        return fn
    return deco


class _Ctx:
    def __init__(self, sd, opset):
        self.sd, self.opset = sd, opset
        self.vars: Dict[str, SDVariable] = {}
        self.consts: Dict[str, torch.Tensor] = {}

    def inp(self, name):
        return self.vars[name]

    def opt(self, n, k):
        return self.vars[n.inputs[k]] if len(n.inputs) > k and n.inputs[k] else None

    def const_of(self, name):
        return self.consts.get(name)

    def ints(self, name):
        c = self.consts.get(name)
        if c is None:
            raise ValueError("input '%s' must be an initializer/constant for static import" % name)
        return [int(v) for v in c.reshape(-1).tolist()]

    def op(self, n, opName, inputs, nOut=1, **kw):
        r = self.sd._op(opName, inputs, name=n.name if isinstance(n, OnnxNode) else n, nOut=nOut, **kw)
        return r[0] if nOut == 1 else r

    def shape_of(self, v: SDVariable):
        return self.sd.variables[v.name].getShape()


def _simple(opName, **kw):
    return lambda c, n: c.op(n, opName, [c.inp(i) for i in n.inputs if i], **kw)


for _o, _s in (("Add", "add"), ("Sub", "subtract"), ("Mul", "multiply"), ("Div", "divide"), ("Pow", "Pow"), ("Max", "maximum"), ("Min", "minimum"),
               ("Relu", "relu"), ("Sigmoid", "sigmoid"), ("Tanh", "tanh"), ("Softplus", "softplus"), ("Softsign", "softsign"), ("Identity", "identity"),
               ("Equal", "equals"), ("Less", "less"), ("Greater", "greater"), ("LessOrEqual", "less_equal"), ("GreaterOrEqual", "greater_equal"),
               ("And", "boolean_and"), ("Or", "boolean_or"), ("Xor", "boolean_xor"), ("Not", "boolean_not"), ("MatMul", "matmul"), ("Shape", "shape_of"),
               ("Size", "size"), ("Sum", "mergeadd"), ("Mean", "mergeavg"), ("Where", "Where"), ("Selu", "selu"), ("Det", "matrix_determinant")):
    onnx_op(_o)(_simple(_s))


def _legacy(opn):
    return lambda c, n: getattr(c.sd.math, opn[0].lower() + opn[1:])(c.inp(n.inputs[0]), name=n.name)


for _o, _l in (("Exp", "Exp"), ("Log", "Log"), ("Sqrt", "Sqrt"), ("Neg", "Neg"), ("Abs", "Abs"), ("Erf", "Erf"), ("Sin", "Sin"), ("Cos", "Cosine"),
               ("Tan", "Tan"), ("Asin", "ASin"), ("Acos", "ACos"), ("Atan", "ATan"), ("Sinh", "Sinh"), ("Cosh", "Cosh"), ("Ceil", "Ceiling"),
               ("Floor", "Floor"), ("Sign", "Sign"), ("Reciprocal", "Reciprocal"), ("Round", "Rint"), ("IsNaN", "IsNan"), ("IsInf", "IsInf"),
               ("Mish", "Mish"), ("HardSigmoid", "HardSigmoid")):
    onnx_op(_o)(_legacy(_l))


@onnx_op("Constant")
def _constant(c, n):
    t = n.a_tensor("value")
    if t is None:
        if "value_float" in n.attrs:
            t = torch.tensor(n.a_float("value_float"))
        elif "value_int" in n.attrs:
            t = torch.tensor(n.a_int("value_int"), dtype=torch.int64)
        elif "value_ints" in n.attrs:
            t = torch.tensor(n.a_ints("value_ints"), dtype=torch.int64)
        else:
            t = torch.tensor(n.a_floats("value_floats"))
    c.consts[n.outputs[0]] = t
    return c.sd.constant(n.outputs[0], t)


@onnx_op("Gemm")
def _gemm(c, n):
    a, b = c.inp(n.inputs[0]), c.inp(n.inputs[1])
    alpha, beta = n.a_float("alpha", 1.0), n.a_float("beta", 1.0)
    y = c.op(n.name + "/mm", "matmul", [a, b], iArgs=[n.a_int("transA"), n.a_int("transB"), 0], tArgs=[alpha, 0.0])
    bias = c.opt(n, 2)
    if bias is None:
        return c.op(n, "identity", [y])
    if beta != 1.0:
        bias = bias * beta
    return c.op(n, "add", [y, bias])


@onnx_op("Softmax", "LogSoftmax")
def _softmax(c, n):
    axis = n.a_int("axis", -1 if c.opset >= 13 else 1)
    opn = "softmax" if n.op == "Softmax" else "log_softmax"
    x = c.inp(n.inputs[0])
    if c.opset < 13:
        shp = c.shape_of(x)
        if shp and axis not in (-1, len(shp) - 1):        # opset < 13: flatten to 2D at `axis`
            lead = int(np.prod(shp[:axis]))
            f = c.op(n.name + "/flat", "reshape", [x], iArgs=[lead, -1])
            s = c.op(n.name + "/sm", opn, [f], iArgs=[-1])
            return c.op(n, "reshape", [s], iArgs=list(shp))
        axis = -1
    return c.op(n, opn, [x], iArgs=[axis])


@onnx_op("LeakyRelu")
def _lrelu(c, n):
    return c.op(n, "lrelu", [c.inp(n.inputs[0])], tArgs=[n.a_float("alpha", 0.01)])


@onnx_op("Elu")
def _elu(c, n):
    return c.op(n, "elu", [c.inp(n.inputs[0])], tArgs=[n.a_float("alpha", 1.0)])


@onnx_op("PRelu")
def _prelu(c, n):
    return c.op(n, "prelu", [c.inp(n.inputs[0]), c.inp(n.inputs[1])])


@onnx_op("Gelu")
def _gelu(c, n):
    x = c.inp(n.inputs[0])
    return c.sd.math.preciseGELU(x, name=n.name) if n.a_str("approximate", "none") == "tanh" else \
        c.op(n, "multiply", [x * 0.5, c.sd.math.erf(x * 0.7071067811865476) + 1.0])


@onnx_op("Clip")
def _clip(c, n):
    lo = n.a_float("min", float("-inf")) if "min" in n.attrs else (float(c.const_of(n.inputs[1])) if len(n.inputs) > 1 and n.inputs[1] else -3.4e38)
    hi = n.a_float("max", float("inf")) if "max" in n.attrs else (float(c.const_of(n.inputs[2])) if len(n.inputs) > 2 and n.inputs[2] else 3.4e38)
    return c.op(n, "clipbyvalue", [c.inp(n.inputs[0])], tArgs=[lo, hi])


def _pads2(pads, nd):
    pads = list(pads) or [0] * (2 * nd)
    return pads[:nd], pads[nd:]


@onnx_op("Conv")
def _conv(c, n):
    x, w = c.inp(n.inputs[0]), c.inp(n.inputs[1])
    b = c.opt(n, 2)
    wshape = c.shape_of(w)
    nd = len(wshape) - 2
    k = n.a_ints("kernel_shape", list(wshape[2:]))
    s = n.a_ints("strides", [1] * nd)
    d = n.a_ints("dilations", [1] * nd)
    lo, hi = _pads2(n.a_ints("pads"), nd)
    group = n.a_int("group", 1)
    auto = n.a_str("auto_pad", "NOTSET")
    same = auto in ("SAME_UPPER", "SAME_LOWER")
    if lo != hi and not same:
        pad = torch.tensor([[0, 0], [0, 0]] + [[a, bb] for a, bb in zip(lo, hi)], dtype=torch.int64)
        x = c.op(n.name + "/pad", "pad", [x, c.sd.constant(None, pad)], iArgs=[0], tArgs=[0.0])
        lo = [0] * nd
    ins = [x, w] + ([b] if b is not None else [])
    if nd == 2 and group == 1:
        return c.op(n, "conv2d", ins, iArgs=[k[0], k[1], s[0], s[1], lo[0], lo[1], d[0], d[1], int(same), 0, 1])
    if nd == 2 and wshape[1] == 1 and group == wshape[0]:
        wd = c.op(n.name + "/w", "permute", [w], iArgs=[1, 0, 2, 3])           # [1, C, kH, kW] = [mC, iC, kH, kW]
        return c.op(n, "depthwise_conv2d", [x, wd] + ([b] if b is not None else []),
                    iArgs=[k[0], k[1], s[0], s[1], lo[0], lo[1], d[0], d[1], int(same), 0, 1])
    if nd == 1 and group == 1:
        return c.op(n, "conv1d", ins, iArgs=[k[0], s[0], lo[0], d[0], int(same), 0, 1])
    if nd == 3 and group == 1:
        return c.op(n, "conv3dnew", ins, iArgs=list(k) + list(s) + list(lo) + list(d) + [int(same), 0, 1])
    raise NotImplementedError("Conv with group=%d, %dD" % (group, nd))


@onnx_op("ConvTranspose")
def _deconv(c, n):
    x, w = c.inp(n.inputs[0]), c.inp(n.inputs[1])
    b = c.opt(n, 2)
    wshape = c.shape_of(w)
    k = n.a_ints("kernel_shape", list(wshape[2:]))
    s = n.a_ints("strides", [1, 1])
    d = n.a_ints("dilations", [1, 1])
    lo, _ = _pads2(n.a_ints("pads"), 2)
    return c.op(n, "deconv2d", [x, w] + ([b] if b is not None else []), iArgs=[k[0], k[1], s[0], s[1], lo[0], lo[1], d[0], d[1], 0, 0, 1])


def _pool(opName):
    def rule(c, n):
        x = c.inp(n.inputs[0])
        k = n.a_ints("kernel_shape")
        nd = len(k)
        s = n.a_ints("strides", [1] * nd)
        d = n.a_ints("dilations", [1] * nd)
        lo, hi = _pads2(n.a_ints("pads"), nd)
        same = n.a_str("auto_pad", "NOTSET") in ("SAME_UPPER", "SAME_LOWER")
        if n.a_int("ceil_mode"):
            shp = c.shape_of(x)
            hi = list(hi)
            for a in range(nd):                    # ceil_mode == extra right padding so that floor() gives the ceil size
                size = shp[2 + a] + lo[a] + hi[a] - ((k[a] - 1) * d[a] + 1)
                if size % s[a]:
                    hi[a] += s[a] - size % s[a]
        if lo != list(hi) and not same:
            val = float("-inf") if opName.startswith("max") else 0.0
            pad = torch.tensor([[0, 0], [0, 0]] + [[a, bb] for a, bb in zip(lo, hi)], dtype=torch.int64)
            x = c.op(n.name + "/pad", "pad", [x, c.sd.constant(None, pad)], iArgs=[0], tArgs=[val if val != float("-inf") else -3.4e38])
            lo = [0] * nd
        extra = 0 if n.a_int("count_include_pad") else 1
        if nd == 2:
            return c.op(n, opName, [x], iArgs=[k[0], k[1], s[0], s[1], lo[0], lo[1], d[0], d[1], int(same), 0 if opName.startswith("max") else extra, 0])
        return c.op(n, opName.replace("2d", "3dnew"), [x], iArgs=list(k) + list(s) + list(lo) + list(d) + [int(same), extra, 0])
    return rule


onnx_op("MaxPool")(_pool("maxpool2d"))
onnx_op("AveragePool")(_pool("avgpool2d"))


@onnx_op("GlobalAveragePool", "GlobalMaxPool")
def _gpool(c, n):
    x = c.inp(n.inputs[0])
    nd = len(c.shape_of(x) or [0, 0, 0, 0])
    return c.op(n, "reduce_mean" if n.op == "GlobalAveragePool" else "reduce_max", [x], iArgs=list(range(2, nd)), bArgs=[True])


@onnx_op("BatchNormalization")
def _bn(c, n):
    x, scale, bias, mean, var = [c.inp(i) for i in n.inputs[:5]]
    return c.op(n, "batchnorm", [x, mean, var, scale, bias], iArgs=[1, 1, 1], tArgs=[n.a_float("epsilon", 1e-5)])


@onnx_op("LayerNormalization")
def _ln(c, n):
    x, g = c.inp(n.inputs[0]), c.inp(n.inputs[1])
    b = c.opt(n, 2)
    axis = n.a_int("axis", -1)
    nd = len(c.shape_of(x))
    dims = list(range(axis % nd, nd))
    return c.op(n, "layer_norm", [x, g] + ([b] if b is not None else []), iArgs=dims)


@onnx_op("InstanceNormalization")
def _inorm(c, n):
    x, scale, bias = [c.inp(i) for i in n.inputs[:3]]
    nd = len(c.shape_of(x))
    dims = list(range(2, nd))
    m = c.op(n.name + "/m", "reduce_mean", [x], iArgs=dims, bArgs=[True])
    v = c.op(n.name + "/v", "reduce_variance", [x], iArgs=dims, bArgs=[True, False])
    sh = [1, -1] + [1] * (nd - 2)
    y = (x - m) / c.sd.math.sqrt(v + n.a_float("epsilon", 1e-5))
    return c.op(n, "add", [y * c.op(n.name + "/s", "reshape", [scale], iArgs=sh), c.op(n.name + "/b", "reshape", [bias], iArgs=sh)])


@onnx_op("LRN")
def _lrn(c, n):
    size = n.a_int("size")
    return c.op(n, "lrn", [c.inp(n.inputs[0])], iArgs=[(size - 1) // 2], tArgs=[n.a_float("bias", 1.0), n.a_float("alpha", 1e-4) / size, n.a_float("beta", 0.75)])


@onnx_op("Dropout")
def _dropout(c, n):
    y = c.op(n, "identity", [c.inp(n.inputs[0])])
    return [y, y]


@onnx_op("Flatten")
def _flatten(c, n):
    return c.op(n, "flatten_2d", [c.inp(n.inputs[0])], iArgs=[n.a_int("axis", 1)])


@onnx_op("Reshape")
def _reshape(c, n):
    x = c.inp(n.inputs[0])
    shp = c.const_of(n.inputs[1]) if len(n.inputs) > 1 else torch.tensor(n.a_ints("shape"))
    if shp is None:
        return c.op(n, "reshape", [x, c.inp(n.inputs[1])])
    shp = [int(v) for v in shp.tolist()]
    xs = c.shape_of(x)
    if xs:
        shp = [xs[k] if v == 0 and k < len(xs) else v for k, v in enumerate(shp)]
    return c.op(n, "reshape", [x], iArgs=shp)


@onnx_op("Transpose")
def _transpose(c, n):
    perm = n.a_ints("perm")
    return c.op(n, "permute", [c.inp(n.inputs[0])], iArgs=perm) if perm else c.op(n, "transpose", [c.inp(n.inputs[0])])


@onnx_op("Concat")
def _concat(c, n):
    return c.op(n, "concat", [c.inp(i) for i in n.inputs], iArgs=[n.a_int("axis")])


@onnx_op("Squeeze", "Unsqueeze")
def _sq(c, n):
    axes = n.a_ints("axes") if "axes" in n.attrs else (c.ints(n.inputs[1]) if len(n.inputs) > 1 else [])
    x = c.inp(n.inputs[0])
    if n.op == "Squeeze":
        return c.op(n, "squeeze", [x], iArgs=axes)
    y = x
    for k, a in enumerate(sorted(axes)):
        y = c.op(n.name if k == len(axes) - 1 else "%s/%d" % (n.name, k), "expand_dims", [y], iArgs=[a])
    return y


@onnx_op("Gather")
def _gather(c, n):
    return c.op(n, "gather", [c.inp(n.inputs[0]), c.inp(n.inputs[1])], iArgs=[n.a_int("axis")])


@onnx_op("Cast")
def _cast(c, n):
    return c.op(n, "cast", [c.inp(n.inputs[0])], dArgs=[_DTN[ONNX_DT[n.a_int("to")]]])


def _reduce(opName):
    def rule(c, n):
        x = c.inp(n.inputs[0])
        axes = n.a_ints("axes") if "axes" in n.attrs else (c.ints(n.inputs[1]) if len(n.inputs) > 1 and n.inputs[1] else [])
        return c.op(n, opName, [x], iArgs=axes, bArgs=[bool(n.a_int("keepdims", 1))])
    return rule


for _o, _s in (("ReduceMean", "reduce_mean"), ("ReduceSum", "reduce_sum"), ("ReduceMax", "reduce_max"), ("ReduceMin", "reduce_min"),
               ("ReduceProd", "reduce_prod"), ("ReduceL1", "reduce_norm1"), ("ReduceL2", "reduce_norm2"), ("ReduceSumSquare", "reduce_sqnorm"),
               ("ReduceLogSumExp", "reduce_logsumexp")):
    onnx_op(_o)(_reduce(_s))


@onnx_op("ArgMax", "ArgMin")
def _arg(c, n):
    return c.op(n, "argmax" if n.op == "ArgMax" else "argmin", [c.inp(n.inputs[0])], iArgs=[n.a_int("axis")], bArgs=[bool(n.a_int("keepdims", 1))])


@onnx_op("Slice")
def _slice(c, n):
    x = c.inp(n.inputs[0])
    if "starts" in n.attrs:
        starts, ends, axes, steps = n.a_ints("starts"), n.a_ints("ends"), n.a_ints("axes"), []
    else:
        starts, ends = c.ints(n.inputs[1]), c.ints(n.inputs[2])
        axes = c.ints(n.inputs[3]) if len(n.inputs) > 3 and n.inputs[3] else []
        steps = c.ints(n.inputs[4]) if len(n.inputs) > 4 and n.inputs[4] else []
    nd = len(c.shape_of(x))
    axes = axes or list(range(len(starts)))
    steps = steps or [1] * len(starts)
    b, e, s, bm, em = [0] * nd, [0] * nd, [1] * nd, (1 << nd) - 1, (1 << nd) - 1
    for a, st, en, sp in zip(axes, starts, ends, steps):
        a %= nd
        b[a], s[a] = st, sp
        bm &= ~(1 << a)
        if abs(en) < 2 ** 31 - 1:
            e[a] = en
            em &= ~(1 << a)
    return c.op(n, "strided_slice", [x], iArgs=[bm, 0, em, 0, 0] + b + e + s)


@onnx_op("Expand")
def _expand(c, n):
    x = c.inp(n.inputs[0])
    shape = c.ints(n.inputs[1])
    xs = c.shape_of(x) or []
    full = list(torch.broadcast_shapes(tuple(xs), tuple(shape)))
    return c.op(n, "broadcast_to", [x], iArgs=full)


@onnx_op("Tile")
def _tile(c, n):
    return c.op(n, "tile", [c.inp(n.inputs[0])], iArgs=c.ints(n.inputs[1]))


@onnx_op("Pad")
def _pad(c, n):
    x = c.inp(n.inputs[0])
    pads = n.a_ints("pads") if "pads" in n.attrs else c.ints(n.inputs[1])
    nd = len(pads) // 2
    val = n.a_float("value", 0.0) if "value" in n.attrs else (float(c.const_of(n.inputs[2])) if len(n.inputs) > 2 and n.inputs[2] else 0.0)
    mode = {"constant": 0, "reflect": 1, "edge": 2}[n.a_str("mode", "constant")]
    pt = torch.tensor([[pads[k], pads[nd + k]] for k in range(nd)], dtype=torch.int64)
    return c.op(n, "pad", [x, c.sd.constant(None, pt)], iArgs=[mode], tArgs=[val])


@onnx_op("ConstantOfShape")
def _cos(c, n):
    v = n.a_tensor("value")
    v = v if v is not None else torch.zeros(1)
    return c.op(n, "fill", [c.inp(n.inputs[0])], tArgs=[float(v.reshape(-1)[0])], dArgs=[_DTN[v.dtype]])


@onnx_op("Range")
def _range(c, n):
    return c.op(n, "range", [c.inp(i) for i in n.inputs])


@onnx_op("OneHot")
def _onehot(c, n):
    vals = c.const_of(n.inputs[2])
    return c.op(n, "onehot", [c.inp(n.inputs[0])], iArgs=[n.a_int("axis", -1), c.ints(n.inputs[1])[0]], tArgs=[float(vals[1]), float(vals[0])],
                dArgs=[_DTN[vals.dtype]])


@onnx_op("Split")
def _split(c, n):
    x = c.inp(n.inputs[0])
    axis = n.a_int("axis")
    sizes = n.a_ints("split") if "split" in n.attrs else (c.ints(n.inputs[1]) if len(n.inputs) > 1 and n.inputs[1] else [])
    k = len(n.outputs)
    if sizes:
        return c.op(n, "split_v", [x, c.sd.constant(None, torch.tensor(sizes, dtype=torch.int64))], nOut=k, iArgs=[axis])
    return c.op(n, "split", [x], nOut=k, iArgs=[k, axis])


@onnx_op("Resize", "Upsample")
def _resize(c, n):
    x = c.inp(n.inputs[0])
    xs = c.shape_of(x)
    mode = n.a_str("mode", "nearest")
    sizes = None
    if n.op == "Resize" and len(n.inputs) > 3 and n.inputs[3]:
        sizes = c.ints(n.inputs[3])[2:]
    else:
        sc = c.const_of(n.inputs[2] if n.op == "Resize" and len(n.inputs) > 2 else n.inputs[1])
        sizes = [int(xs[2 + k] * float(sc[2 + k])) for k in range(len(xs) - 2)]
    nhwc = c.op(n.name + "/nhwc", "permute", [x], iArgs=[0, 2, 3, 1])
    opn = "resize_nearest_neighbor" if mode == "nearest" else ("resize_bicubic" if mode == "cubic" else "resize_bilinear")
    half = n.a_str("coordinate_transformation_mode", "half_pixel") == "half_pixel" and mode != "nearest"
    r = c.op(n.name + "/r", opn, [nhwc], iArgs=sizes, bArgs=[n.a_str("coordinate_transformation_mode") == "align_corners", half])
    return c.op(n, "permute", [r], iArgs=[0, 3, 1, 2])


class OnnxImporter:
    @staticmethod
    def register(opType, rule):
        _RULES[opType] = rule

    @staticmethod
    def supportedOps():
        return sorted(_RULES)

    @staticmethod
    def importGraph(path_or_bytes, overrides: Optional[Dict[str, Callable]] = None, trainable=False, device=None) -> SameDiff:
        """Initializers become CONSTANTs (VARIABLEs with ``trainable=True``), graph inputs become placeholders."""
        buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(str(path_or_bytes), "rb").read()
        model = Msg(bytes(buf))
        opset = max([o.int(2) for o in model.msgs(8) if o.str(1) in ("", "ai.onnx")] or [13])
        g = model.msg(7)
        sd = SameDiff(device)
        c = _Ctx(sd, opset)
        for init in g.msgs(5):
            t = parse_tensor(init)
            name = init.str(8)
            c.consts[name] = t
            if trainable and t.is_floating_point():
                c.vars[name] = sd.var(name, t)
            else:
                c.vars[name] = sd.constant(name, t)
        for vi in g.msgs(11):
            name = vi.str(1)
            if name in c.vars:
                continue
            tt = vi.msg(2).msg(1) if vi.msg(2) is not None else None
            dt = ONNX_DT.get(tt.int(1), torch.float32) if tt is not None else torch.float32
            shape = []
            if tt is not None and tt.msg(2) is not None:
                shape = [d.int(1) if d.has(1) else -1 for d in tt.msg(2).msgs(1)]
            c.vars[name] = sd.placeHolder(name, _DTN.get(dt, "FLOAT"), *shape)
        for m in g.msgs(1):
            n = OnnxNode(m)
            rule = (overrides or {}).get(n.name) or _RULES.get(n.op)
            if rule is None:
                raise NotImplementedError("ONNX op '%s' (node %s) has no import rule" % (n.op, n.name))
            res = rule(c, n)
            res = res if isinstance(res, (list, tuple)) else [res]
            for oname, v in zip(n.outputs, res):
                if oname and v is not None:
                    c.vars[oname] = v
                    if v.name != oname and oname not in sd.variables:
                        sd.renameVariable(v.name, oname)
        sd.onnxOutputs = [o.str(1) for o in g.msgs(12)]
        return sd


from . import import_onnx_ext  # noqa: E402,F401  (registers the second tranche of rules)
