"""TensorFlow frozen-graph (.pb GraphDef) -> SameDiff importer.

Reference: nd4j/samediff-import/samediff-import-tensorflow (ImportGraph + per-op MappingProcess rules,
OpMappingRegistry), legacy imports/graphmapper/tf/TFGraphMapper.java, and the BERT recipe of SURVEY B.10
(placeholder overrides, node filter). Here a mapping rule is a Python function
``rule(ctx, node) -> SDVariable | list`` registered with ``@tf_op``; unknown TF ops fail loudly with
the op name so the registry can be extended (``TFGraphMapper.register``).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from .protowire import Msg
from .samediff import LOGIC_NUM, OpType, SameDiff, SDVariable

# tensorflow/core/framework/types.proto
TF_DT = {1: torch.float32, 2: torch.float64, 3: torch.int32, 4: torch.uint8, 5: torch.int16, 6: torch.int8, 9: torch.int64,
         10: torch.bool, 14: torch.bfloat16, 19: torch.float16, 17: torch.int32, 22: torch.int64, 23: torch.int64}
_NP = {1: "<f4", 2: "<f8", 3: "<i4", 4: "u1", 5: "<i2", 6: "i1", 9: "<i8", 10: "u1", 14: "<u2", 19: "<f2", 17: "<u2", 22: "<u4", 23: "<u8"}
_DTN = {torch.float32: "FLOAT", torch.float64: "DOUBLE", torch.int32: "INT", torch.int64: "LONG", torch.bool: "BOOL", torch.uint8: "UBYTE",
        torch.int16: "SHORT", torch.int8: "BYTE", torch.float16: "HALF", torch.bfloat16: "BFLOAT16"}


def parse_shape(m: Optional[Msg]):
    if m is None or m.int(3):
        return None
    return [d.int(1) for d in m.msgs(2)]


def parse_tensor(m: Msg) -> torch.Tensor:
    dt = m.int(1)
    shape = parse_shape(m.msg(2)) or []
    n = int(np.prod(shape)) if shape else 1
    td = TF_DT.get(dt)
    if td is None:
        if dt == 7:      # DT_STRING
            return None
        raise ValueError("unsupported TF dtype %d" % dt)
    raw = m.bytes(4)
    if raw:
        arr = np.frombuffer(raw, dtype=np.dtype(_NP[dt])).copy()
        if dt == 14:
            t = torch.from_numpy(arr.view(np.int16)).view(torch.bfloat16)
        elif dt in (17, 22, 23):
            t = torch.from_numpy(arr.astype(np.int64)).to(td)
        else:
            t = torch.from_numpy(arr)
            t = t.bool() if dt == 10 else t
    else:
        if dt == 1:
            vals = m.floats(5)
        elif dt == 2:
            vals = m.floats(6, True)
        elif dt in (3, 4, 5, 6, 17):
            vals = m.ints(7)
        elif dt in (9,):
            vals = m.ints(10)
        elif dt == 10:
            vals = m.ints(11)
        elif dt == 19 or dt == 14:
            hv = np.asarray(m.ints(13), dtype=np.uint16)
            t16 = torch.from_numpy(hv.view(np.int16).copy())
            vals = t16.view(torch.float16 if dt == 19 else torch.bfloat16).float().tolist()
        else:
            vals = m.ints(7)
        if len(vals) == 0:
            vals = [0]
        if len(vals) < n:                      # TF repeats the last value
            vals = list(vals) + [vals[-1]] * (n - len(vals))
        t = torch.tensor(vals[:n] if shape else vals[:1]).to(td)
    return t.reshape(shape).to(td)


class TFNode:
    def __init__(self, m: Msg):
        self.name, self.op = m.str(1), m.str(2)
        self.inputs = m.strs(3)
        self.attrs: Dict[str, Msg] = {}
        for e in m.msgs(5):
            self.attrs[e.str(1)] = e.msg(2)

    def a_int(self, k, default=0):
        a = self.attrs.get(k)
        return a.int(3, default) if a is not None else default

    def a_float(self, k, default=0.0):
        a = self.attrs.get(k)
        return a.float(4, default) if a is not None else default

    def a_bool(self, k, default=False):
        a = self.attrs.get(k)
        return bool(a.int(5, int(default))) if a is not None else default

    def a_str(self, k, default=""):
        a = self.attrs.get(k)
        return a.bytes(2).decode() if a is not None and a.has(2) else default

    def a_type(self, k, default=None):
        a = self.attrs.get(k)
        return TF_DT.get(a.int(6)) if a is not None and a.has(6) else default

    def a_ints(self, k):
        a = self.attrs.get(k)
        lst = a.msg(1) if a is not None else None
        return lst.ints(3) if lst is not None else []

    def a_shape(self, k):
        a = self.attrs.get(k)
        return parse_shape(a.msg(7)) if a is not None and a.has(7) else None

    def a_tensor(self, k):
        a = self.attrs.get(k)
        return parse_tensor(a.msg(8)) if a is not None and a.has(8) else None


_RULES: Dict[str, Callable] = {}


def tf_op(*names):
    def deco(fn):
        for n in names:
            _RULES[n] = fn
        return fn
    return deco


class _Ctx:
    def __init__(self, sd: SameDiff):
        self.sd = sd
        self.vars: Dict[str, SDVariable] = {}          # "node" / "node:k" -> variable
        self.consts: Dict[str, torch.Tensor] = {}

    def inp(self, ref: str) -> SDVariable:
        ref = ref[:-2] if ref.endswith(":0") else ref
        return self.vars[ref]

    def const_of(self, ref: str) -> Optional[torch.Tensor]:
        ref = ref[:-2] if ref.endswith(":0") else ref
        return self.consts.get(ref)

    def ints(self, ref) -> List[int]:
        c = self.const_of(ref)
        if c is None:
            raise ValueError("input '%s' must be a constant for static import" % ref)
        return [int(v) for v in c.reshape(-1).tolist()]

    def op(self, name, opName, inputs, nOut=1, **kw):
        r = self.sd._op(opName, inputs, name=name, nOut=nOut, **kw)
        return r[0] if nOut == 1 else r


def _data(n: TFNode):
    return [i for i in n.inputs if not i.startswith("^")]


# ----------------------------------------------------------------------------- rules
def _simple(opName, nin=None, **fixed):
    def rule(c, n):
        ins = [c.inp(i) for i in _data(n)][:nin]
        return c.op(n.name, opName, ins, **fixed)
    return rule


for _tf, _sd in (("Add", "add"), ("AddV2", "add"), ("Sub", "subtract"), ("Mul", "multiply"), ("RealDiv", "divide"), ("Div", "divide"),
                 ("Maximum", "maximum"), ("Minimum", "minimum"), ("Pow", "Pow"), ("SquaredDifference", "squaredsubtract"),
                 ("FloorDiv", "floordiv"), ("FloorMod", "floormod"), ("Less", "less"), ("LessEqual", "less_equal"), ("Greater", "greater"),
                 ("GreaterEqual", "greater_equal"), ("Equal", "equals"), ("NotEqual", "not_equals"), ("LogicalAnd", "boolean_and"),
                 ("LogicalOr", "boolean_or"), ("Relu", "relu"), ("Relu6", "relu6"), ("Sigmoid", "sigmoid"), ("Tanh", "tanh"), ("Elu", "elu"),
                 ("Selu", "selu"), ("Softplus", "softplus"), ("Softsign", "softsign"), ("Identity", "identity"), ("StopGradient", "stop_gradient"),
                 ("Square", "square"), ("BiasAdd", "biasadd"), ("AddN", "mergeadd"), ("ZerosLike", "zeros_as"), ("OnesLike", "ones_as"),
                 ("Shape", "shape_of"), ("Rank", "rank"), ("Size", "size"), ("Atan2", "tf_atan2"), ("Select", "select"), ("SelectV2", "Where"),
                 ("InvertPermutation", "invert_permutation"), ("MatrixDiag", "matrix_diag"), ("MatrixDiagPart", "matrix_diag_part"),
                 ("Cholesky", "cholesky"), ("MatrixInverse", "matrix_inverse"), ("MatrixDeterminant", "matrix_determinant"),
                 ("Lgamma", "lgamma"), ("Digamma", "digamma"), ("Rint", "rint"), ("Floor", "Floor"), ("Log1p", "Log1p"), ("PreventGradient", "identity"),
                 ("Snapshot", "identity"), ("LogicalNot", "boolean_not"), ("IdentityN", "identity_n")):
    if _tf == "IdentityN":
        tf_op(_tf)(lambda c, n: c.op(n.name, "identity_n", [c.inp(i) for i in _data(n)], nOut=len(_data(n))))
    else:
        tf_op(_tf)(_simple(_sd))


def _legacy_unary(fam, opn):
    def rule(c, n):
        return getattr(c.sd.math, opn[0].lower() + opn[1:])(c.inp(_data(n)[0]), name=n.name)
    return rule


for _tf, _opn in (("Exp", "Exp"), ("Log", "Log"), ("Sqrt", "Sqrt"), ("Rsqrt", "RSqrt"), ("Neg", "Neg"), ("Abs", "Abs"), ("Erf", "Erf"),
                  ("Erfc", "Erfc"), ("Sin", "Sin"), ("Cos", "Cosine"), ("Tan", "Tan"), ("Asin", "ASin"), ("Acos", "ACos"), ("Atan", "ATan"),
                  ("Sinh", "Sinh"), ("Cosh", "Cosh"), ("Asinh", "ASinh"), ("Acosh", "ACosh"), ("Atanh", "ATanh"), ("Ceil", "Ceiling"),
                  ("Sign", "Sign"), ("Reciprocal", "Reciprocal"), ("Round", "Round"), ("Expm1", "Expm1"), ("IsNan", "IsNan"), ("IsInf", "IsInf"),
                  ("IsFinite", "IsFinite"), ("Inv", "Reciprocal")):
    tf_op(_tf)(_legacy_unary(None, _opn))


@tf_op("Placeholder", "PlaceholderWithDefault")
def _placeholder(c, n):
    dt = n.a_type("dtype", torch.float32)
    shape = n.a_shape("shape") or []
    return c.sd.placeHolder(n.name, _DTN.get(dt, "FLOAT"), *[s if s is not None else -1 for s in shape])


@tf_op("Const")
def _const(c, n):
    t = n.a_tensor("value")
    if t is None:
        t = torch.zeros(0)
    c.consts[n.name] = t
    return c.sd.constant(n.name, t)


@tf_op("NoOp", "Assert", "CheckNumerics")
def _noop(c, n):
    ins = _data(n)
    if n.op == "CheckNumerics" and ins:
        return c.op(n.name, "identity", [c.inp(ins[0])])
    return None


@tf_op("MatMul")
def _matmul(c, n):
    a, b = [c.inp(i) for i in _data(n)]
    return c.op(n.name, "matmul", [a, b], iArgs=[int(n.a_bool("transpose_a")), int(n.a_bool("transpose_b")), 0])


@tf_op("BatchMatMul", "BatchMatMulV2", "BatchMatMulV3")
def _bmm(c, n):
    a, b = [c.inp(i) for i in _data(n)]
    return c.op(n.name, "matmul", [a, b], iArgs=[int(n.a_bool("adj_x")), int(n.a_bool("adj_y")), 0])


@tf_op("Softmax")
def _softmax(c, n):
    return c.op(n.name, "softmax", [c.inp(_data(n)[0])], iArgs=[-1])


@tf_op("LogSoftmax")
def _log_softmax(c, n):
    return c.op(n.name, "log_softmax", [c.inp(_data(n)[0])], iArgs=[-1])


@tf_op("LeakyRelu")
def _lrelu(c, n):
    return c.op(n.name, "lrelu", [c.inp(_data(n)[0])], tArgs=[n.a_float("alpha", 0.2)])


@tf_op("Reshape")
def _reshape(c, n):
    x, s = _data(n)
    shp = c.const_of(s)
    if shp is not None:
        return c.op(n.name, "reshape", [c.inp(x)], iArgs=[int(v) for v in shp.reshape(-1).tolist()])
    return c.op(n.name, "reshape", [c.inp(x), c.inp(s)])


@tf_op("Transpose")
def _transpose(c, n):
    x, p = _data(n)
    return c.op(n.name, "permute", [c.inp(x)], iArgs=c.ints(p))


@tf_op("Squeeze")
def _squeeze(c, n):
    return c.op(n.name, "squeeze", [c.inp(_data(n)[0])], iArgs=n.a_ints("squeeze_dims"))


@tf_op("ExpandDims")
def _expand(c, n):
    x, a = _data(n)
    return c.op(n.name, "expand_dims", [c.inp(x)], iArgs=c.ints(a)[:1])


@tf_op("ConcatV2")
def _concat(c, n):
    ins = _data(n)
    return c.op(n.name, "concat", [c.inp(i) for i in ins[:-1]], iArgs=c.ints(ins[-1])[:1])


@tf_op("Concat")
def _concat_v1(c, n):
    ins = _data(n)
    return c.op(n.name, "concat", [c.inp(i) for i in ins[1:]], iArgs=c.ints(ins[0])[:1])


@tf_op("Pack")
def _pack(c, n):
    return c.op(n.name, "stack", [c.inp(i) for i in _data(n)], iArgs=[n.a_int("axis")])


@tf_op("Unpack")
def _unpack(c, n):
    return c.op(n.name, "unstack", [c.inp(_data(n)[0])], nOut=n.a_int("num"), iArgs=[n.a_int("axis")])


@tf_op("Split")
def _split(c, n):
    ax, x = _data(n)
    k = n.a_int("num_split")
    return c.op(n.name, "split", [c.inp(x)], nOut=k, iArgs=[k, c.ints(ax)[0]])


@tf_op("SplitV")
def _splitv(c, n):
    x, sizes, ax = _data(n)
    return c.op(n.name, "split_v", [c.inp(x), c.inp(sizes)], nOut=len(c.ints(sizes)), iArgs=[c.ints(ax)[0]])


@tf_op("Slice")
def _slice(c, n):
    x, b, s = _data(n)
    return c.op(n.name, "slice", [c.inp(x)], iArgs=c.ints(b) + c.ints(s))


@tf_op("StridedSlice")
def _strided(c, n):
    x, b, e, s = _data(n)
    masks = [n.a_int("begin_mask"), n.a_int("ellipsis_mask"), n.a_int("end_mask"), n.a_int("new_axis_mask"), n.a_int("shrink_axis_mask")]
    if all(c.const_of(v) is not None for v in (b, e, s)):
        return c.op(n.name, "strided_slice", [c.inp(x)], iArgs=masks + c.ints(b) + c.ints(e) + c.ints(s))
    return c.op(n.name, "strided_slice", [c.inp(x), c.inp(b), c.inp(e), c.inp(s)], iArgs=masks)


@tf_op("Gather", "GatherV2", "ResourceGather")
def _gather(c, n):
    ins = _data(n)
    axis = c.ints(ins[2])[0] if len(ins) > 2 else 0
    return c.op(n.name, "gather", [c.inp(ins[0]), c.inp(ins[1])], iArgs=[axis])


@tf_op("GatherNd")
def _gather_nd(c, n):
    return c.op(n.name, "gather_nd", [c.inp(i) for i in _data(n)])


@tf_op("OneHot")
def _onehot(c, n):
    idx, depth, on, off = _data(n)
    onv, offv = c.const_of(on), c.const_of(off)
    return c.op(n.name, "onehot", [c.inp(idx)], iArgs=[n.a_int("axis", -1), c.ints(depth)[0]],
                tArgs=[float(onv), float(offv)], dArgs=[_DTN.get(onv.dtype, "FLOAT")])


@tf_op("Cast")
def _cast(c, n):
    return c.op(n.name, "cast", [c.inp(_data(n)[0])], dArgs=[_DTN.get(n.a_type("DstT", torch.float32), "FLOAT")])


def _reduce(opName, legacy=None):
    def rule(c, n):
        x, ax = _data(n)
        keep = n.a_bool("keep_dims")
        dims = c.ints(ax)
        if legacy:
            fam, opn = legacy
            return c.sd.math._legacy(fam, opn, [c.inp(x)], n.name, dimensions=dims, bArgs=[keep])
        return c.op(n.name, opName, [c.inp(x)], iArgs=dims, bArgs=[keep])
    return rule


tf_op("Mean")(_reduce("reduce_mean"))
tf_op("Sum")(_reduce("reduce_sum"))
tf_op("Max")(_reduce("reduce_max"))
tf_op("Min")(_reduce("reduce_min"))
tf_op("Prod")(_reduce("reduce_prod"))
tf_op("All")(_reduce(None, ("reduce_bool", "All")))
tf_op("Any")(_reduce(None, ("reduce_bool", "Any")))


@tf_op("ArgMax", "ArgMin")
def _argmax(c, n):
    x, ax = _data(n)
    return c.op(n.name, "argmax" if n.op == "ArgMax" else "argmin", [c.inp(x)], iArgs=c.ints(ax)[:1])


def _nhwc(n):
    return n.a_str("data_format", "NHWC") == "NHWC"


def _hw(v, nhwc):
    v = list(v) or [1, 1, 1, 1]
    return (v[1], v[2]) if nhwc else (v[2], v[3])


@tf_op("Conv2D")
def _conv2d(c, n):
    x, w = _data(n)
    nhwc = _nhwc(n)
    sH, sW = _hw(n.a_ints("strides"), nhwc)
    dH, dW = _hw(n.a_ints("dilations"), nhwc)
    wshape = c.sd.variables[c.inp(w).name].getShape()
    kH, kW = (wshape[0], wshape[1]) if wshape else (0, 0)
    same = n.a_str("padding", "VALID") == "SAME"
    return c.op(n.name, "conv2d", [c.inp(x), c.inp(w)], iArgs=[kH, kW, sH, sW, 0, 0, dH, dW, int(same), int(nhwc), 0])


@tf_op("DepthwiseConv2dNative")
def _dwconv(c, n):
    x, w = _data(n)
    nhwc = _nhwc(n)
    sH, sW = _hw(n.a_ints("strides"), nhwc)
    dH, dW = _hw(n.a_ints("dilations"), nhwc)
    wshape = c.sd.variables[c.inp(w).name].getShape()
    same = n.a_str("padding", "VALID") == "SAME"
    return c.op(n.name, "depthwise_conv2d", [c.inp(x), c.inp(w)], iArgs=[wshape[0], wshape[1], sH, sW, 0, 0, dH, dW, int(same), int(nhwc), 0])


def _pool(opName):
    def rule(c, n):
        nhwc = _nhwc(n)
        kH, kW = _hw(n.a_ints("ksize"), nhwc)
        sH, sW = _hw(n.a_ints("strides"), nhwc)
        same = n.a_str("padding", "VALID") == "SAME"
        extra = 1 if opName == "avgpool2d" else 0          # TF average excludes padding
        return c.op(n.name, opName, [c.inp(_data(n)[0])], iArgs=[kH, kW, sH, sW, 0, 0, 1, 1, int(same), extra, int(nhwc)])
    return rule


tf_op("MaxPool")(_pool("maxpool2d"))
tf_op("AvgPool")(_pool("avgpool2d"))


@tf_op("FusedBatchNorm", "FusedBatchNormV2", "FusedBatchNormV3")
def _fbn(c, n):
    x, scale, offset, mean, var = [c.inp(i) for i in _data(n)[:5]]
    nhwc = _nhwc(n)
    y = c.op(n.name, "batchnorm", [x, mean, var, scale, offset], iArgs=[1, 1, 3 if nhwc else 1], tArgs=[n.a_float("epsilon", 1e-3)])
    return [y, mean, var, mean, var, mean]


@tf_op("Pad", "PadV2", "MirrorPad")
def _pad(c, n):
    ins = _data(n)
    mode = {"REFLECT": 1, "SYMMETRIC": 2}.get(n.a_str("mode", ""), 0) if n.op == "MirrorPad" else 0
    val = float(c.const_of(ins[2])) if len(ins) > 2 else 0.0
    return c.op(n.name, "pad", [c.inp(ins[0]), c.inp(ins[1])], iArgs=[mode], tArgs=[val])


@tf_op("Tile")
def _tile(c, n):
    x, m = _data(n)
    return c.op(n.name, "tile", [c.inp(x)], iArgs=c.ints(m))


@tf_op("Fill")
def _fill(c, n):
    s, v = _data(n)
    cv = c.const_of(v)
    if cv is not None:
        return c.op(n.name, "fill", [c.inp(s)], tArgs=[float(cv)], dArgs=[_DTN.get(cv.dtype, "FLOAT")])
    return c.op(n.name, "fill", [c.inp(s), c.inp(v)])


@tf_op("Range")
def _range(c, n):
    return c.op(n.name, "range", [c.inp(i) for i in _data(n)])


@tf_op("Where")
def _where(c, n):
    return c.op(n.name, "Where", [c.inp(_data(n)[0])])


@tf_op("ReverseV2")
def _reverse(c, n):
    x, ax = _data(n)
    return c.op(n.name, "reverse", [c.inp(x)], iArgs=c.ints(ax))


@tf_op("TopKV2")
def _topk(c, n):
    x, k = _data(n)
    return c.op(n.name, "top_k", [c.inp(x)], nOut=2, iArgs=[c.ints(k)[0]], bArgs=[n.a_bool("sorted", True)])


@tf_op("Cumsum", "Cumprod")
def _cum(c, n):
    x, ax = _data(n)
    return c.op(n.name, "cumsum" if n.op == "Cumsum" else "cumprod", [c.inp(x)], iArgs=[int(n.a_bool("exclusive")), int(n.a_bool("reverse"))] + c.ints(ax))


@tf_op("ClipByValue")
def _clip(c, n):
    x, lo, hi = _data(n)
    return c.op(n.name, "clipbyvalue", [c.inp(x)], tArgs=[float(c.const_of(lo)), float(c.const_of(hi))])


@tf_op("ResizeBilinear", "ResizeNearestNeighbor", "ResizeBicubic")
def _resize(c, n):
    x, s = _data(n)
    opn = {"ResizeBilinear": "resize_bilinear", "ResizeNearestNeighbor": "resize_nearest_neighbor", "ResizeBicubic": "resize_bicubic"}[n.op]
    return c.op(n.name, opn, [c.inp(x)], iArgs=c.ints(s), bArgs=[n.a_bool("align_corners"), n.a_bool("half_pixel_centers")])


@tf_op("SpaceToDepth", "DepthToSpace")
def _s2d(c, n):
    return c.op(n.name, "space_to_depth" if n.op == "SpaceToDepth" else "depth_to_space", [c.inp(_data(n)[0])],
                iArgs=[n.a_int("block_size"), int(_nhwc(n))])


@tf_op("L2Loss")
def _l2loss(c, n):
    return c.op(n.name, "l2_loss", [c.inp(_data(n)[0])])


@tf_op("SoftmaxCrossEntropyWithLogits")
def _sce(c, n):
    lg, lb = [c.inp(i) for i in _data(n)]
    loss = c.op(n.name, "softmax_cross_entropy_loss_with_logits", [lg, lb], iArgs=[-1])
    return [loss, loss]


@tf_op("SparseSoftmaxCrossEntropyWithLogits")
def _ssce(c, n):
    lg, lb = [c.inp(i) for i in _data(n)]
    loss = c.op(n.name, "sparse_softmax_cross_entropy_loss_with_logits", [lb, lg])
    return [loss, loss]


# -- control flow
def _logic(kind, nOut=1):
    def rule(c, n):
        ins = [c.inp(i) for i in _data(n)]
        r = c.sd._op(kind, ins, name=n.name, nOut=nOut, opType=OpType.LOGIC, opNum=LOGIC_NUM[kind])
        if kind == "enter":
            o = c.sd.ops[c.sd._producer[r[0].name]]
            o.frameName = n.a_str("frame_name", n.name)
            o.isConstantEnter = n.a_bool("is_constant")
        return r[0] if nOut == 1 else r
    return rule


tf_op("Switch")(_logic("switch", 2))
tf_op("Enter")(_logic("enter"))
tf_op("Exit")(_logic("exit"))
tf_op("NextIteration")(_logic("next_iteration"))
tf_op("LoopCond")(_logic("loop_cond"))


@tf_op("Merge")
def _merge(c, n):
    """Merge may reference a NextIteration that is defined later: unresolved inputs are patched after the pass."""
    ins, pending = [], []
    for k, i in enumerate(_data(n)):
        key = i[:-2] if i.endswith(":0") else i
        if key in c.vars:
            ins.append(c.vars[key])
        else:
            ins.append(ins[0] if ins else None)
            pending.append((k, key))
    ins = [v if v is not None else next(x for x in ins if x is not None) for v in ins]
    r = c.sd._op("merge", ins, name=n.name, nOut=2, opType=OpType.LOGIC, opNum=LOGIC_NUM["merge"])
    c.pending_merges.append((c.sd._producer[r[0].name], pending))
    return r


class TFGraphMapper:
    @staticmethod
    def register(opName, rule):
        _RULES[opName] = rule

    @staticmethod
    def supportedOps():
        return sorted(_RULES)

    @staticmethod
    def importGraph(path_or_bytes, overrides: Optional[Dict[str, Callable]] = None, opFilter: Optional[Callable] = None,
                    device=None) -> SameDiff:
        """``overrides``: node name -> rule(ctx, node) replacing the default mapping (e.g. IteratorGetNext ->
        placeholders). ``opFilter(node) -> bool``: True = skip the node."""
        buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(str(path_or_bytes), "rb").read()
        g = Msg(bytes(buf))
        nodes = [TFNode(m) for m in g.msgs(1)]
        sd = SameDiff(device)
        c = _Ctx(sd)
        c.pending_merges = []
        by_name = {n.name: n for n in nodes}
        done = set()

        def emit(n: TFNode):
            if n.name in done:
                return
            done.add(n.name)
            if opFilter is not None and opFilter(n):
                return
            if n.op != "Merge":
                for i in _data(n):
                    dep = i.split(":")[0]
                    if dep in by_name and dep not in done:
                        emit(by_name[dep])
            else:                               # break loop cycles: only the first (Enter) input is required up front
                first = _data(n)[0].split(":")[0]
                if first in by_name:
                    emit(by_name[first])
            rule = (overrides or {}).get(n.name) or _RULES.get(n.op)
            if rule is None:
                raise NotImplementedError("TF op '%s' (node %s) has no import rule" % (n.op, n.name))
            res = rule(c, n)
            if res is None:
                return
            res = res if isinstance(res, (list, tuple)) else [res]
            for k, v in enumerate(res):
                c.vars[n.name if k == 0 else "%s:%d" % (n.name, k)] = v

        import sys
        sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
        for n in nodes:
            emit(n)
        for op_name, pend in c.pending_merges:
            o = sd.ops[op_name]
            for k, key in pend:
                v = c.vars[key]
                old = o.inputs[k]
                o.inputs[k] = v.name
                if o.name in sd._consumers.get(old, []) and old not in o.inputs:
                    sd._consumers[old].remove(o.name)
                sd._consumers[v.name].append(o.name)
        return sd
from . import import_tf_ext  # noqa: E402,F401  (registers the second tranche of rules)
