"""Test-case table for the op-validation sweep: for (nearly) every registered custom op one or more invocations with an
INDEPENDENT reference (NumPy / closed-form / fp64 PyTorch functional written here, not the op's implementation module).

Reference: platform-tests/src/test/java/org/eclipse/deeplearning4j/nd4j/autodiff/opvalidation/* (12 suites: TransformOpValidation,
ReductionOpValidation, ShapeOpValidation, LayerOpValidation, LossOpValidation, MiscOpValidation, RandomOpValidation, …) and
nd4j-api autodiff/validation/OpValidation.java (coverage accounting). ``cases()`` returns {op name: [Case]}; the sweep itself is
:func:`deeplearning4j_amd.autodiff.validation.OpValidation.sweep`.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


class Case:
    """ins: tensors (fp32 / ints); i/t/b: iArgs / tArgs / bArgs; ref(*ins) -> tensor | list of tensors | None (= only execution,
    shape function and determinism are checked); check(outs, ins) -> None | error string for property-style validation;
    grad: numeric fp64 gradient check of ``<op>_bp`` against central differences of this forward case; gin: indices of the
    inputs that are differentiated."""

    def __init__(self, ins, i=(), t=(), b=(), d=(), ref=None, check=None, grad=False, gin=None, rtol=1e-5, atol=1e-6, tag="",
                 bp_out=False, canon=None):
        self.ins, self.i, self.t, self.b, self.d = list(ins), list(i), list(t), list(b), list(d)
        self.ref, self.check, self.grad, self.gin, self.rtol, self.atol, self.tag = ref, check, grad, gin, rtol, atol, tag
        self.bp_out = bp_out          # the _bp op also takes the forward output after the incoming gradient (softmax_bp)
        self.canon = canon            # outputs -> canonical form, applied to the op's and the reference's outputs alike (eigenvector
        #                               order / phase, ...) before they are compared


def _g(seed=0):
    return torch.Generator().manual_seed(seed)


def R(*shape, seed=0, lo=None, hi=None):
    x = torch.randn(*shape, generator=_g(seed))
    if lo is not None:
        x = torch.rand(*shape, generator=_g(seed)) * (hi - lo) + lo
    return x


def I(*vals, dtype=torch.int64):
    return torch.tensor(list(vals), dtype=dtype)


def d64(x):
    return x.double() if isinstance(x, torch.Tensor) and x.is_floating_point() else x


# ------------------------------------------------------------------------------------------------ elementwise families
def _sel(x, alpha=1.6732632423543772, lam=1.0507009873554805):
    return lam * torch.where(x > 0, x, alpha * (torch.exp(x) - 1))


UNARY: Dict[str, Callable] = {
    "relu": lambda x: torch.clamp_min(x, 0), "sigmoid": lambda x: 1 / (1 + torch.exp(-x)), "tanh": torch.tanh,
    "elu": lambda x: torch.where(x > 0, x, torch.exp(x) - 1), "selu": _sel, "softplus": lambda x: torch.log1p(torch.exp(x)),
    "softsign": lambda x: x / (1 + x.abs()), "cube": lambda x: x ** 3, "hardtanh": lambda x: x.clamp(-1, 1),
    "hardsigmoid": lambda x: (0.2 * x + 0.5).clamp(0, 1), "lrelu": lambda x: torch.where(x > 0, x, 0.01 * x),
    "relu6": lambda x: x.clamp(0, 6),
    "rationaltanh": lambda x: 1.7159 * torch.sign(2 * x / 3) * (1 - 1 / (1 + (2 * x / 3).abs() + (2 * x / 3) ** 2 + 1.41645 * (2 * x / 3) ** 4)),
    "rectifiedtanh": lambda x: torch.tanh(x).clamp_min(0), "identity": lambda x: x, "square": lambda x: x * x,
    "rint": lambda x: torch.from_numpy(np.rint(x.numpy())), "Floor": torch.floor, "Log1p": None, "stop_gradient": lambda x: x,
    "zeros_as": torch.zeros_like, "ones_as": torch.ones_like, "to_double": lambda x: x.double(), "to_float32": lambda x: x.float(),
    "to_float16": lambda x: x.half(), "to_int32": lambda x: x.to(torch.int32), "to_int64": lambda x: x.to(torch.int64),
    "flatten_2d": None, "noop": None,
}
UNARY_POS = {"Log1p": lambda x: torch.log1p(x), "lgamma": torch.lgamma, "digamma": torch.digamma}

BINARY: Dict[str, Callable] = {
    "add": lambda a, b: a + b, "subtract": lambda a, b: a - b, "multiply": lambda a, b: a * b, "divide": lambda a, b: a / b,
    "realdiv": lambda a, b: a / b, "reversesubtract": lambda a, b: b - a, "reversedivide": lambda a, b: b / a,
    "squaredsubtract": lambda a, b: (a - b) ** 2, "maximum": torch.maximum, "minimum": torch.minimum,
    "floordiv": lambda a, b: torch.floor(a / b), "floormod": lambda a, b: a - torch.floor(a / b) * b,
    "mod": lambda a, b: torch.fmod(a, b), "truncatediv": lambda a, b: torch.trunc(a / b), "tf_atan2": lambda a, b: torch.atan2(a, b),
    "divide_no_nan": lambda a, b: torch.where(b == 0, torch.zeros_like(a), a / b),
    "reversemod": lambda a, b: torch.fmod(b, a),
    "equals": lambda a, b: a == b, "not_equals": lambda a, b: a != b, "greater": lambda a, b: a > b,
    "greater_equal": lambda a, b: a >= b, "less": lambda a, b: a < b, "less_equal": lambda a, b: a <= b,
}
BINARY_GRAD = {"add", "subtract", "multiply", "divide", "realdiv", "reversesubtract", "reversedivide", "squaredsubtract", "maximum", "minimum"}

REDUCE: Dict[str, Callable] = {
    "reduce_sum": lambda x, d, k: x.sum(d, keepdim=k), "reduce_mean": lambda x, d, k: x.mean(d, keepdim=k),
    "reduce_max": lambda x, d, k: x.amax(d, keepdim=k), "reduce_min": lambda x, d, k: x.amin(d, keepdim=k),
    "reduce_prod": lambda x, d, k: x.prod(d[0], keepdim=k), "reduce_norm1": lambda x, d, k: x.abs().sum(d, keepdim=k),
    "reduce_norm2": lambda x, d, k: (x * x).sum(d, keepdim=k).sqrt(), "reduce_sqnorm": lambda x, d, k: (x * x).sum(d, keepdim=k),
    "reduce_norm_max": lambda x, d, k: x.abs().amax(d, keepdim=k),
    "reduce_variance": lambda x, d, k: ((x - x.mean(d, keepdim=True)) ** 2).mean(d, keepdim=k),
    "reduce_stdev": lambda x, d, k: ((x - x.mean(d, keepdim=True)) ** 2).mean(d, keepdim=k).sqrt(),
    "reduce_logsumexp": lambda x, d, k: torch.log(torch.exp(x).sum(d, keepdim=k)),
}


def _np_conv2d(x, w, b, s, p):
    """Direct NCHW convolution, w [oC, iC, kH, kW] — seven nested loops written out, the literal definition."""
    x, w = x.double().numpy(), w.double().numpy()
    N, C, H, W = x.shape
    O, _, KH, KW = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)))
    OH, OW = (H + 2 * p - KH) // s + 1, (W + 2 * p - KW) // s + 1
    y = np.zeros((N, O, OH, OW))
    for kh in range(KH):
        for kw in range(KW):
            patch = xp[:, :, kh:kh + s * OH:s, kw:kw + s * OW:s]
            y += np.einsum("nchw,oc->nohw", patch, w[:, :, kh, kw])
    if b is not None:
        y += b.double().numpy().reshape(1, -1, 1, 1)
    return torch.from_numpy(y)


def _pool_ref(x, k, s, mode):
    x = x.double()
    N, C, H, W = x.shape
    OH, OW = (H - k) // s + 1, (W - k) // s + 1
    cols = torch.stack([x[:, :, i:i + s * OH:s, j:j + s * OW:s] for i in range(k) for j in range(k)], 0)
    return cols.amax(0) if mode == "max" else (cols.mean(0) if mode == "avg" else (cols.abs() ** 2).sum(0).sqrt())


def _lstm_cell_ref(x, h, c, Wx, Wr, b):
    z = x @ Wx + h @ Wr + b
    n = h.shape[1]
    i, f, g, o = torch.sigmoid(z[:, :n]), torch.sigmoid(z[:, n:2 * n]), torch.tanh(z[:, 2 * n:3 * n]), torch.sigmoid(z[:, 3 * n:])
    c2 = f * c + i * g
    return o * torch.tanh(c2), c2


def cases() -> Dict[str, List[Case]]:
    C: Dict[str, List[Case]] = {}

    def add(name, *cs):
        C.setdefault(name, []).extend(cs)

    x34, y34 = R(3, 4, seed=1), R(3, 4, seed=2)
    pos = R(3, 4, seed=3, lo=0.5, hi=2.5)
    # ---- transforms
    for n, fn in UNARY.items():
        ref = (lambda f: (lambda x: f(d64(x))))(fn) if fn is not None else None
        add(n, Case([x34], t=[0.0] if n in ("relu", "relu6") else [], ref=ref, grad=True))
    for n, fn in UNARY_POS.items():
        add(n, Case([pos], ref=(lambda f: (lambda x: f(d64(x))))(fn), rtol=1e-4))
    add("thresholdedrelu", Case([x34], t=[0.3], ref=lambda x: torch.where(d64(x) > 0.3, d64(x), torch.zeros_like(d64(x))), grad=True))
    add("crelu", Case([x34], ref=lambda x: torch.cat([d64(x).clamp_min(0), (-d64(x)).clamp_min(0)], -1), grad=True))
    add("prelu", Case([x34, torch.full((4,), 0.25)], i=[1], ref=lambda x, a: torch.where(d64(x) > 0, d64(x), 0.25 * d64(x)), grad=True, gin=[0]))
    for n, fn in BINARY.items():
        a, b = (pos, R(3, 4, seed=4, lo=0.5, hi=2.5)) if n in ("floordiv", "floormod", "mod", "truncatediv", "reversemod") else (x34, y34 + 0.1)
        add(n, Case([a, b], ref=(lambda f: (lambda p, q: f(d64(p), d64(q))))(fn), grad=n in BINARY_GRAD))
        add(n, Case([a, b[0:1]], ref=(lambda f: (lambda p, q: f(d64(p), d64(q))))(fn), tag="broadcast"))
    add("Pow", Case([pos, R(3, 4, seed=5, lo=0.5, hi=2.0)], ref=lambda a, b: d64(a) ** d64(b), grad=True, rtol=1e-4))
    for n, fn in (("boolean_and", torch.logical_and), ("boolean_or", torch.logical_or), ("boolean_xor", torch.logical_xor)):
        add(n, Case([x34 > 0, y34 > 0], ref=fn))
    add("boolean_not", Case([x34 > 0], ref=torch.logical_not))
    ia, ib = torch.randint(0, 255, (3, 4), generator=_g(6), dtype=torch.int32), torch.randint(0, 255, (3, 4), generator=_g(7), dtype=torch.int32)
    for n, fn in (("bitwise_and", torch.bitwise_and), ("bitwise_or", torch.bitwise_or), ("bitwise_xor", torch.bitwise_xor)):
        add(n, Case([ia, ib], ref=fn))
    sh = torch.randint(0, 5, (3, 4), generator=_g(8), dtype=torch.int32)
    add("shift_bits", Case([ia, sh], ref=lambda a, s: a << s))
    add("rshift_bits", Case([ia, sh], ref=lambda a, s: a >> s))
    add("toggle_bits", Case([ia], ref=lambda a: ~a))
    add("bits_hamming_distance", Case([ia, ib], ref=lambda a, b: torch.tensor(sum(bin(int(v)).count("1") for v in (a ^ b).reshape(-1)))))
    for n, cmp in (("eq_scalar", "eq"), ("neq_scalar", "ne"), ("gt_scalar", "gt"), ("gte_scalar", "ge"), ("lt_scalar", "lt"), ("lte_scalar", "le")):
        add(n, Case([torch.tensor(0.2), torch.tensor(0.2)], ref=(lambda c: (lambda x, s: getattr(torch, c)(x, s)))(cmp)))
    add("check_numerics", Case([x34, torch.zeros(1)]))
    add("clipbyvalue", Case([x34], t=[-0.5, 0.5], ref=lambda x: d64(x).clamp(-0.5, 0.5)))
    add("clipbynorm", Case([x34], t=[1.0], ref=lambda x: d64(x) * min(1.0, 1.0 / float(d64(x).norm())), grad=True))
    add("clipbyavgnorm", Case([x34], t=[0.05], ref=lambda x: d64(x) * min(1.0, 0.05 / (float(d64(x).norm()) / 12)), grad=True))
    add("clip_by_global_norm", Case([x34, y34], t=[1.0], ref=lambda a, b: [d64(a) * min(1.0, 1.0 / float(torch.sqrt((d64(a) ** 2).sum() + (d64(b) ** 2).sum()))),
                                                                        d64(b) * min(1.0, 1.0 / float(torch.sqrt((d64(a) ** 2).sum() + (d64(b) ** 2).sum())))]))
    rot = lambda a, k: ((a.to(torch.int64) << k) | (a.to(torch.int64) >> (32 - k))) & 0xFFFFFFFF
    add("cyclic_shift_bits", Case([ia, torch.full((3, 4), 3, dtype=torch.int32)], ref=lambda a, k: rot(a, 3).to(torch.int32)))
    add("cyclic_rshift_bits", Case([ia, torch.full((3, 4), 3, dtype=torch.int32)], ref=lambda a, k: (((a.to(torch.int64) >> 3) | (a.to(torch.int64) << 29)) & 0xFFFFFFFF).to(torch.int32)))
    for nm in ("floordiv_bp", "floormod_bp", "mod_bp", "reversemod_bp", "assign_bp"):
        add(nm, Case([pos, R(3, 4, seed=4, lo=0.5, hi=2.5), y34], check=lambda o, ins: None if [list(t.shape) for t in o[:2]] == [[3, 4], [3, 4]] else "gradient shapes"))
    add("dynamic_partition_bp", Case([I(1.0, 2.0, 3.0, 4.0, dtype=torch.float32), I(0, 1, 0, 1), I(10.0, 30.0, dtype=torch.float32), I(20.0, 40.0, dtype=torch.float32)], i=[2],
                                     ref=lambda x, p, g0, g1: torch.tensor([10.0, 20.0, 30.0, 40.0], dtype=torch.float64)))
    add("to_uint32", Case([pos]))
    add("to_uint64", Case([pos]))
    add("axpy", Case([x34, y34], t=[2.5], ref=lambda a, b: 2.5 * d64(a) + d64(b)))
    add("assign", Case([x34, y34], ref=lambda a, b: d64(b)))
    add("cast", Case([x34], i=[6]))
    add("zero_fraction", Case([torch.tensor([0.0, 1.0, 0.0, 2.0])], ref=lambda x: torch.tensor(0.5)))
    add("is_non_decreasing", Case([torch.tensor([1.0, 1.0, 2.0])], ref=lambda x: torch.tensor(True)))
    add("is_strictly_increasing", Case([torch.tensor([1.0, 1.0, 2.0])], ref=lambda x: torch.tensor(False)))
    add("is_numeric_tensor", Case([x34], ref=lambda x: torch.tensor(True)))
    add("betainc", Case([I(2.0, 0.5, dtype=torch.float32), I(3.0, 0.5, dtype=torch.float32), I(0.4, 0.3, dtype=torch.float32)],
                        ref=lambda a, b, x: torch.tensor([0.5248, 0.36901], dtype=torch.float64), atol=1e-3))
    add("igamma", Case([pos, pos + 0.3], ref=lambda a, x: torch.special.gammainc(d64(a), d64(x)), rtol=1e-4))
    add("igammac", Case([pos, pos + 0.3], ref=lambda a, x: torch.special.gammaincc(d64(a), d64(x)), rtol=1e-4))
    add("zeta", Case([pos + 1.1, pos], ref=lambda s, q: torch.special.zeta(d64(s), d64(q)), rtol=1e-4))
    add("polygamma", Case([torch.ones(3, 4), pos], ref=lambda n, x: torch.special.polygamma(1, d64(x)), rtol=1e-4))
    # ---- reductions
    x234 = R(2, 3, 4, seed=9)
    for n, fn in REDUCE.items():
        add(n, Case([x234], i=[1], ref=(lambda f: (lambda x: f(d64(x), [1], False)))(fn), grad=n != "reduce_logsumexp", rtol=1e-5))
        add(n, Case([x234], i=[0, 2], b=[True], ref=(lambda f: (lambda x: f(d64(x), [0, 2], True)))(fn), tag="keepdims") if n != "reduce_prod"
            else Case([x234], i=[2], b=[True], ref=lambda x: d64(x).prod(2, keepdim=True), tag="keepdims"))
    add("argmax", Case([x234], i=[2], ref=lambda x: x.argmax(2)))
    add("argmin", Case([x234], i=[1], ref=lambda x: x.argmin(1)))
    add("argamax", Case([x234], i=[2], ref=lambda x: x.abs().argmax(2)))
    add("argamin", Case([x234], i=[2], ref=lambda x: x.abs().argmin(2)))
    add("moments", Case([x234], i=[0, 1], ref=lambda x: [d64(x).mean((0, 1)), d64(x).var((0, 1), unbiased=False)]))
    add("norm", Case([x34], t=[2.0]))
    add("l2_loss", Case([x34], ref=lambda x: (d64(x) ** 2).sum() / 2))
    add("trace", Case([R(4, 4, seed=10)], ref=lambda x: d64(x).diagonal().sum()))
    add("cumsum", Case([x34], i=[0, 0, 1], ref=lambda x: d64(x).cumsum(1), grad=True))
    add("cumprod", Case([pos], i=[0, 0, 1], ref=lambda x: d64(x).cumprod(1), grad=True))
    add("standardize", Case([x34], i=[1], ref=lambda x: (d64(x) - d64(x).mean(1, keepdim=True)) / d64(x).std(1, unbiased=False, keepdim=True), grad=True, rtol=1e-4))
    add("reduce_dot_bp", Case([x34, y34, R(3, seed=11)], i=[1]))
    add("percentile", Case([x34], t=[50.0, 2.0]))
    add("nth_element", Case([x34, I(1)], i=[0], ref=lambda x, n: d64(x).sort(-1)[0][:, 1]))
    add("top_k", Case([x34], i=[2], ref=lambda x: list(d64(x).topk(2, -1))))
    add("in_top_k", Case([x34, I(0, 1, 2)], i=[2], ref=lambda p, t: torch.tensor([int(t[r]) in p[r].topk(2)[1].tolist() for r in range(3)])))
    add("histogram", Case([x34], i=[4]))
    add("histogram_fixed_width", Case([x34, I(-2.0, 2.0, dtype=torch.float32), I(4)], ref=lambda x, r, n: torch.histc(x.clamp(-2, 1.9999), 4, -2, 2).long()))
    add("bincount", Case([I(1, 1, 3, 0, dtype=torch.int32)], ref=lambda x: torch.tensor([1, 2, 0, 1])))
    add("confusion_matrix", Case([I(0, 1, 2, 1), I(0, 2, 2, 1)], i=[3], ref=lambda a, b: torch.tensor([[1, 0, 0], [0, 1, 1], [0, 0, 1]])))
    add("sufficient_statistics", Case([x34, I(0)], ref=None))
    add("normalize_moments", Case([torch.tensor(12.0), R(4, seed=12), R(4, seed=13, lo=1, hi=2)], t=[0.0]))
    # ---- shape / data movement
    add("reshape", Case([x34], i=[99, 2, 6], ref=lambda x: d64(x).reshape(2, 6), grad=False),
        Case([x34, I(4, 3)], ref=lambda x, s: d64(x).reshape(4, 3), tag="shape-input"))
    add("reshapeas", Case([x34, R(2, 6)], ref=lambda x, y: d64(x).reshape(2, 6)))
    add("permute", Case([x234], i=[2, 0, 1], ref=lambda x: d64(x).permute(2, 0, 1)))
    add("transpose", Case([x34], ref=lambda x: d64(x).t()))
    add("expand_dims", Case([x34], i=[1], ref=lambda x: d64(x).unsqueeze(1)))
    add("squeeze", Case([R(3, 1, 4)], i=[1], ref=lambda x: d64(x).squeeze(1)))
    add("flatten", Case([x34, y34], i=[ord("c")], ref=lambda a, b: torch.cat([d64(a).reshape(-1), d64(b).reshape(-1)])))
    add("concat", Case([x34, y34], i=[1], ref=lambda a, b: torch.cat([d64(a), d64(b)], 1), grad=True))
    add("stack", Case([x34, y34], i=[0], ref=lambda a, b: torch.stack([d64(a), d64(b)], 0)))
    add("parallel_stack", Case([x34, y34], ref=lambda a, b: torch.stack([d64(a), d64(b)], 0)))
    add("unstack", Case([x34], i=[0], ref=lambda x: list(d64(x).unbind(0))))
    add("split", Case([x34], i=[2, 1], ref=lambda x: list(d64(x).chunk(2, 1))))
    add("split_v", Case([x34, I(1, 3)], i=[1], ref=lambda x, s: list(d64(x).split([1, 3], 1))))
    add("tile", Case([x34], i=[2, 1], ref=lambda x: d64(x).repeat(2, 1), grad=True))
    add("tile_to_shape", Case([R(1, 4)], i=[3, 4], ref=lambda x: d64(x).expand(3, 4), grad=True))
    add("repeat", Case([x34], i=[2, 1], ref=lambda x: d64(x).repeat_interleave(2, 1)))
    add("broadcast_to", Case([R(1, 4), I(3, 4)], ref=lambda x, s: d64(x).expand(3, 4)))
    add("reverse", Case([x34], i=[1], ref=lambda x: d64(x).flip(1), grad=True))
    add("reverse_sequence", Case([x234, I(2, 3)], i=[1, 0], ref=lambda x, l: torch.stack([torch.cat([d64(x)[b, :int(l[b])].flip(0), d64(x)[b, int(l[b]):]]) for b in range(2)])))
    add("roll", Case([x34], i=[1, 1], ref=lambda x: d64(x).roll(1, 1)))
    add("slice", Case([x34], i=[1, 0, 2, 3], ref=lambda x: d64(x)[1:3, 0:3], grad=True))
    add("strided_slice", Case([x34], i=[0, 0, 0, 0, 0, 1, 0, 3, 4, 1, 2], ref=lambda x: d64(x)[1:3, 0:4:2], grad=True))
    add("gather", Case([x34, I(2, 0)], i=[1], ref=lambda x, ix: d64(x)[:, [2, 0]]))
    add("gather_nd", Case([x34, torch.tensor([[0, 1], [2, 3]])], ref=lambda x, ix: torch.stack([d64(x)[0, 1], d64(x)[2, 3]])))
    add("embedding_lookup", Case([x34, I(2, 0)], i=[0], ref=lambda x, ix: d64(x)[[2, 0]]))
    z32 = torch.zeros(3, 2)
    for n, fn in (("scatter_add", lambda o, u: o + u), ("scatter_sub", lambda o, u: o - u), ("scatter_upd", lambda o, u: u)):
        add(n, Case([z32 + 1, I(0, 2), R(2, 2, seed=14)], ref=(lambda f: (lambda z, ix, u: torch.stack([f(d64(z)[0], d64(u)[0]), d64(z)[1], f(d64(z)[2], d64(u)[1])])))(fn) if fn else None))
    for n, fn in (("scatter_mul", lambda o, u: o * u), ("scatter_div", lambda o, u: o / u), ("scatter_max", torch.maximum), ("scatter_min", torch.minimum)):
        add(n, Case([z32 + 1, I(0, 2), R(2, 2, seed=15, lo=0.5, hi=2)], ref=(lambda f: (lambda z, ix, u: torch.stack([f(d64(z)[0], d64(u)[0]), d64(z)[1], f(d64(z)[2], d64(u)[1])])))(fn)))
    add("scatter_nd", Case([torch.tensor([[1], [3]]), I(9.0, 10.0, dtype=torch.float32), I(5)], ref=lambda ix, u, s: torch.tensor([0, 9.0, 0, 10.0, 0], dtype=torch.float64)))
    for n, sgn in (("scatter_nd_add", 1.0), ("scatter_nd_sub", -1.0)):
        add(n, Case([torch.ones(5), torch.tensor([[1], [3]]), I(9.0, 10.0, dtype=torch.float32)],
                    ref=(lambda s: (lambda x, ix, u: torch.tensor([1, 1 + s * 9.0, 1, 1 + s * 10.0, 1], dtype=torch.float64)))(sgn)))
    add("scatter_nd_update", Case([torch.ones(5), torch.tensor([[1], [3]]), I(9.0, 10.0, dtype=torch.float32)], ref=lambda x, ix, u: torch.tensor([1, 9.0, 1, 10.0, 1], dtype=torch.float64)))
    add("pad", Case([torch.ones(2, 2), torch.tensor([[1, 0], [0, 2]])], i=[0], t=[5.0], ref=lambda x, p: F.pad(d64(x), [0, 2, 1, 0], value=5.0)))
    add("mirror_pad", Case([x34, torch.tensor([[1, 1], [1, 1]])], i=[0], ref=lambda x, p: F.pad(d64(x)[None, None], [1, 1, 1, 1], mode="reflect")[0, 0]))
    add("onehot", Case([I(0, 2)], i=[-1, 3], t=[1.0, 0.0], ref=lambda x: torch.eye(3, dtype=torch.float64)[[0, 2]]))
    add("eye", Case([], i=[99, 3, 3] if False else [3], ref=None))
    add("fill", Case([I(2, 3)], t=[1.5], ref=lambda s: torch.full((2, 3), 1.5, dtype=torch.float64)))
    add("fill_as", Case([x34], t=[2.0], ref=lambda x: torch.full((3, 4), 2.0, dtype=torch.float64)))
    add("range", Case([], t=[0.0, 5.0, 1.0], ref=lambda: torch.arange(0.0, 5.0, 1.0, dtype=torch.float64)))
    add("lin_space", Case([torch.tensor(0.0), torch.tensor(1.0), torch.tensor(5)], ref=lambda a, b, n: torch.linspace(0, 1, 5, dtype=torch.float64)))
    add("meshgrid", Case([I(1.0, 2.0, dtype=torch.float32), I(3.0, 4.0, 5.0, dtype=torch.float32)], i=[0], ref=None))
    add("diag", Case([I(1.0, 2.0, 3.0, dtype=torch.float32)], ref=lambda x: torch.diag(d64(x))))
    add("diag_part", Case([R(3, 3, seed=16)], ref=lambda x: d64(x).diagonal()))
    add("matrix_diag", Case([x34], ref=lambda x: torch.diag_embed(d64(x))))
    add("matrix_diag_part", Case([R(2, 3, 3, seed=17)], ref=lambda x: d64(x).diagonal(dim1=-2, dim2=-1)))
    add("matrix_set_diag", Case([R(3, 3, seed=18), I(1.0, 2.0, 3.0, dtype=torch.float32)], ref=lambda x, d: d64(x) - torch.diag(d64(x).diagonal()) + torch.diag(d64(d))))
    add("matrix_band_part", Case([R(4, 4, seed=19)], i=[1, 0], ref=lambda x: torch.tril(d64(x)) - torch.tril(d64(x), -2)))
    add("triu", Case([R(4, 4, seed=20)], i=[0], ref=lambda x: torch.triu(d64(x)), grad=True))
    add("tri", Case([], i=[3, 4, 0], ref=lambda: torch.tril(torch.ones(3, 4, dtype=torch.float64))))
    add("shape_of", Case([x234], ref=lambda x: torch.tensor([2, 3, 4])))
    add("shapes_of", Case([x34, x234], ref=lambda a, b: [torch.tensor([3, 4]), torch.tensor([2, 3, 4])]))
    add("size", Case([x234], ref=lambda x: torch.tensor(24)))
    add("size_at", Case([x234], i=[1], ref=lambda x: torch.tensor(3)))
    add("rank", Case([x234], ref=lambda x: torch.tensor(3)))
    add("order", Case([x34], i=[ord("c")]))
    add("invert_permutation", Case([I(2, 0, 1)], ref=lambda p: torch.tensor([1, 2, 0])))
    add("sequence_mask", Case([I(1, 3)], i=[4], ref=lambda l: torch.tensor([[1, 0, 0, 0], [1, 1, 1, 0]]).bool()))
    add("unique", Case([I(3, 1, 3, 2, 1)], ref=lambda x: [torch.tensor([3, 1, 2]), torch.tensor([0, 1, 0, 2, 1])]))
    add("unique_with_counts", Case([I(3, 1, 3, 2, 1)], ref=lambda x: [torch.tensor([3, 1, 2]), torch.tensor([0, 1, 0, 2, 1]), torch.tensor([2, 2, 1])]))
    add("listdiff", Case([I(1, 2, 3, 4), I(2, 4)], ref=lambda a, b: [torch.tensor([1, 3]), torch.tensor([0, 2])]))
    add("dynamic_partition", Case([I(1.0, 2.0, 3.0, 4.0, dtype=torch.float32), I(0, 1, 0, 1)], i=[2], ref=lambda x, p: [torch.tensor([1.0, 3.0], dtype=torch.float64), torch.tensor([2.0, 4.0], dtype=torch.float64)]))
    add("dynamic_stitch", Case([I(0, 2), I(1, 3), I(1.0, 3.0, dtype=torch.float32), I(2.0, 4.0, dtype=torch.float32)], i=[2], ref=lambda a, b, c, d: torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64)))
    add("where_np", Case([x34 > 0, x34, y34], ref=lambda c, a, b: torch.where(c, d64(a), d64(b))))
    add("select", Case([x34 > 0, x34, y34], ref=lambda c, a, b: torch.where(c, d64(a), d64(b))))
    add("Where", Case([x34 > 0, x34, y34], ref=lambda c, a, b: torch.where(c, d64(a), d64(b))))
    add("choose", Case([x34], i=[3], t=[0.0]))
    add("space_to_depth", Case([R(1, 4, 4, 2, seed=21)], i=[2, 1], ref=None))
    add("depth_to_space", Case([R(1, 2, 2, 8, seed=22)], i=[2, 1], ref=None))
    add("space_to_batch", Case([torch.arange(16.0).reshape(1, 4, 4, 1), torch.zeros(2, 2, dtype=torch.int64)], i=[2], ref=None))
    add("batch_to_space", Case([torch.arange(16.0).reshape(4, 2, 2, 1), torch.zeros(2, 2, dtype=torch.int64)], i=[2], ref=None))
    add("space_to_batch_nd", Case([torch.arange(16.0).reshape(1, 4, 4, 1), I(2, 2), torch.zeros(2, 2, dtype=torch.int64)], ref=None))
    add("batch_to_space_nd", Case([torch.arange(16.0).reshape(4, 2, 2, 1), I(2, 2), torch.zeros(2, 2, dtype=torch.int64)], ref=None))
    add("ismax", Case([x34], i=[1], ref=lambda x: (d64(x) == d64(x).amax(1, keepdim=True))))
    add("mergeadd", Case([x34, y34, pos], ref=lambda a, b, c: d64(a) + d64(b) + d64(c), grad=True))
    add("mergeavg", Case([x34, y34, pos], ref=lambda a, b, c: (d64(a) + d64(b) + d64(c)) / 3, grad=True))
    add("mergemax", Case([x34, y34, pos], ref=lambda a, b, c: torch.maximum(torch.maximum(d64(a), d64(b)), d64(c)), grad=True))
    add("mergemaxindex", Case([x34, y34, pos], ref=lambda a, b, c: torch.stack([a, b, c]).argmax(0)))
    add("identity_n", Case([x34, y34], ref=lambda a, b: [d64(a), d64(b)]))
    add("cross", Case([R(5, 3, seed=23), R(5, 3, seed=24)], ref=lambda a, b: torch.cross(d64(a), d64(b), dim=-1)))
    add("bitcast", Case([x34], i=[7]))
    add("broadcast_dynamic_shape", Case([I(3, 1), I(1, 4)], ref=lambda a, b: torch.tensor([3, 4])))
    add("evaluate_reduction_shape", Case([I(2, 3, 4), I(1)], b=[False, False], ref=lambda s, d: torch.tensor([2, 4])))
    add("create", Case([I(2, 3)], i=[ord("c"), 5]))
    add("set_shape", Case([x34, I(2, 6)]))
    add("tear", Case([x34], i=[1]))
    # segments
    seg_x, seg_i = I(1.0, 2.0, 3.0, 4.0, dtype=torch.float32), I(0, 0, 1, 1)
    for n, fn in (("segment_sum", lambda a: a.sum()), ("segment_mean", lambda a: a.mean()), ("segment_max", lambda a: a.max()),
                  ("segment_min", lambda a: a.min()), ("segment_prod", lambda a: a.prod())):
        add(n, Case([seg_x, seg_i], ref=(lambda f: (lambda x, s: torch.stack([f(d64(x)[:2]), f(d64(x)[2:])])))(fn), grad=True, gin=[0]))
        add("unsorted_" + n, Case([seg_x, I(1, 0, 1, 0)], i=[2], ref=(lambda f: (lambda x, s: torch.stack([f(d64(x)[[1, 3]]), f(d64(x)[[0, 2]])])))(fn), grad=True, gin=[0]))
    add("unsorted_segment_sqrt_n", Case([seg_x, I(1, 0, 1, 0)], i=[2], ref=lambda x, s: torch.stack([d64(x)[[1, 3]].sum(), d64(x)[[0, 2]].sum()]) / math.sqrt(2), grad=True, gin=[0]))
    # ---- linalg
    a34, b54 = R(3, 4, seed=25), R(5, 4, seed=26)
    add("matmul", Case([a34, b54], i=[0, 1, 0], ref=lambda a, b: d64(a) @ d64(b).t(), grad=True),
        Case([R(2, 3, 4, seed=27), R(2, 4, 5, seed=28)], ref=lambda a, b: d64(a) @ d64(b), tag="batched"))
    add("tensormmul", Case([R(2, 3, 4, seed=29), R(4, 3, 5, seed=30)], i=[2, 1, 2, 2, 1, 0], ref=lambda a, b: torch.tensordot(d64(a), d64(b), dims=([1, 2], [1, 0]))))
    add("batched_gemm", Case([torch.tensor(1.0), torch.tensor(0.0), R(3, 4, seed=31), R(3, 4, seed=32), R(4, 2, seed=33), R(4, 2, seed=34)],
                             i=[0, 0, 3, 2, 4, 3, 4, 3, 2], ref=lambda al, be, a0, a1, b0, b1: [d64(a0) @ d64(b0), d64(a1) @ d64(b1)]))
    add("xw_plus_b", Case([a34, R(4, 5, seed=35), R(5, seed=36)], ref=lambda x, w, b: d64(x) @ d64(w) + d64(b), grad=True))
    add("relu_layer", Case([a34, R(4, 5, seed=35), R(5, seed=36)], ref=lambda x, w, b: (d64(x) @ d64(w) + d64(b)).clamp_min(0)))
    add("biasadd", Case([a34, R(4, seed=37)], ref=lambda x, b: d64(x) + d64(b), grad=True))
    m4 = R(4, 4, seed=38)
    spd = m4 @ m4.t() + 4 * torch.eye(4)
    add("matrix_inverse", Case([spd], ref=lambda a: torch.linalg.inv(d64(a)), rtol=1e-3, atol=1e-4))
    add("matrix_determinant", Case([spd], ref=lambda a: torch.linalg.det(d64(a)), rtol=1e-4))
    add("log_matrix_determinant", Case([spd], ref=None))
    add("logdet", Case([spd], ref=lambda a: torch.logdet(d64(a)), rtol=1e-4))
    add("cholesky", Case([spd], ref=lambda a: torch.linalg.cholesky(d64(a)), rtol=1e-4, atol=1e-5))
    add("sqrtm", Case([spd], check=lambda o, ins: None if torch.allclose(o[0].double() @ o[0].double(), ins[0].double(), atol=1e-3) else "sqrtm(a)^2 != a"))
    add("solve", Case([spd, R(4, 2, seed=39)], ref=lambda a, b: torch.linalg.solve(d64(a), d64(b)), rtol=1e-3, atol=1e-4))
    add("triangular_solve", Case([torch.tril(spd), R(4, 2, seed=39)], b=[True, False], ref=lambda a, b: torch.linalg.solve_triangular(d64(a), d64(b), upper=False), rtol=1e-3, atol=1e-4))
    add("lstsq", Case([R(5, 3, seed=40), R(5, 2, seed=41)], t=[0.0], b=[True], ref=lambda a, b: torch.linalg.lstsq(d64(a), d64(b)).solution, rtol=1e-3, atol=1e-4))
    add("solve_ls", Case([R(5, 3, seed=40), R(5, 2, seed=41)], t=[0.0], b=[True], ref=lambda a, b: torch.linalg.lstsq(d64(a), d64(b)).solution, rtol=1e-3, atol=1e-4))
    add("svd", Case([a34], i=[0, 1, 16], check=lambda o, ins: None if torch.allclose((o[1] * o[0]) @ o[2].t(), ins[0], atol=1e-4) else "U S V^T != a"))
    add("qr", Case([R(4, 3, seed=42)], b=[False], check=lambda o, ins: None if torch.allclose(o[0] @ o[1], ins[0], atol=1e-4) else "QR != a"))
    add("lu", Case([spd], ref=None))
    add("eig", Case([spd], ref=None))
    # ---- nn
    img = R(2, 3, 8, 8, seed=43)
    w_khwio = R(3, 3, 3, 5, seed=44) * 0.3          # wFormat 0: [kH, kW, iC, oC]
    bias5 = R(5, seed=45)
    add("conv2d", Case([img, w_khwio, bias5], i=[3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0], ref=lambda x, w, b: _np_conv2d(x, w.permute(3, 2, 0, 1), b, 1, 1), grad=True, rtol=1e-4, atol=1e-5),
        Case([img, w_khwio], i=[3, 3, 2, 2, 0, 0, 1, 1, 0, 0, 0], ref=lambda x, w: _np_conv2d(x, w.permute(3, 2, 0, 1), None, 2, 0), tag="stride2", rtol=1e-4, atol=1e-5))
    add("pointwise_conv2d", Case([img, R(1, 1, 3, 5, seed=46)], i=[0, 0], ref=lambda x, w: _np_conv2d(x, w.permute(3, 2, 0, 1), None, 1, 0), rtol=1e-4, atol=1e-5))
    add("conv1d", Case([R(2, 3, 9, seed=47), R(3, 3, 4, seed=48)], i=[3, 1, 0, 1, 0, 0, 0], ref=lambda x, w: F.conv1d(d64(x), d64(w).permute(2, 1, 0)), grad=True, rtol=1e-4, atol=1e-5))
    add("conv3dnew", Case([R(1, 2, 5, 5, 5, seed=49), R(2, 2, 2, 2, 3, seed=50)], i=[2, 2, 2, 1, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0, 0], ref=lambda x, w: F.conv3d(d64(x), d64(w).permute(4, 3, 0, 1, 2)), grad=True, rtol=1e-4, atol=1e-5))
    add("deconv2d", Case([R(1, 3, 4, 4, seed=51), R(2, 2, 5, 3, seed=52)], i=[2, 2, 2, 2, 0, 0, 1, 1, 0, 0, 0], ref=lambda x, w: F.conv_transpose2d(d64(x), d64(w).permute(3, 2, 0, 1), stride=2), grad=True, rtol=1e-4, atol=1e-5))
    add("deconv3d", Case([R(1, 3, 3, 3, 3, seed=53), R(2, 2, 2, 4, 3, seed=54)], i=[2, 2, 2, 1, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0, 0], ref=lambda x, w: F.conv_transpose3d(d64(x), d64(w).permute(4, 3, 0, 1, 2)), grad=True, rtol=1e-4, atol=1e-5))
    add("depthwise_conv2d", Case([img, R(3, 3, 3, 2, seed=55)], i=[3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0], ref=lambda x, w: F.conv2d(d64(x), d64(w).permute(2, 3, 0, 1).reshape(6, 1, 3, 3), padding=1, groups=3), grad=True, rtol=1e-4, atol=1e-5))
    add("sconv2d", Case([img, R(3, 3, 3, 2, seed=55), R(1, 1, 6, 4, seed=56)], i=[3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0],
                        ref=lambda x, wd, wp: F.conv2d(F.conv2d(d64(x), d64(wd).permute(2, 3, 0, 1).reshape(6, 1, 3, 3), padding=1, groups=3), d64(wp).permute(3, 2, 0, 1)), grad=True, rtol=1e-4, atol=1e-5))
    add("maxpool2d", Case([img], i=[2, 2, 2, 2, 0, 0, 1, 1, 0, 0, 0], ref=lambda x: _pool_ref(x, 2, 2, "max"), grad=True))
    add("avgpool2d", Case([img], i=[2, 2, 2, 2, 0, 0, 1, 1, 0, 0, 0], ref=lambda x: _pool_ref(x, 2, 2, "avg"), grad=True))
    add("pnormpool2d", Case([img], i=[2, 2, 2, 2, 0, 0, 1, 1, 0, 2, 0], t=[1e-12], ref=lambda x: _pool_ref(x, 2, 2, "pnorm"), grad=True, rtol=1e-4))
    vol = R(1, 2, 4, 4, 4, seed=57)
    add("maxpool3dnew", Case([vol], i=[2, 2, 2, 2, 2, 2, 0, 0, 0, 1, 1, 1, 0, 1, 0], ref=lambda x: F.max_pool3d(d64(x), 2), grad=True))
    add("avgpool3dnew", Case([vol], i=[2, 2, 2, 2, 2, 2, 0, 0, 0, 1, 1, 1, 0, 1, 0], ref=lambda x: F.avg_pool3d(d64(x), 2), grad=True))
    add("max_pool_with_argmax", Case([img], i=[2, 2, 2, 2, 0, 0, 1, 1, 0, 0, 0], ref=None))
    add("upsampling2d", Case([img], i=[2, 2, 1], ref=lambda x: d64(x).repeat_interleave(2, 2).repeat_interleave(2, 3), grad=True))
    add("upsampling3d", Case([vol], i=[2, 2, 2, 1], ref=lambda x: d64(x).repeat_interleave(2, 2).repeat_interleave(2, 3).repeat_interleave(2, 4), grad=True))
    add("im2col", Case([img], i=[2, 2, 2, 2, 0, 0, 1, 1, 0], t=[0.0], ref=None, grad=True))
    add("col2im", Case([R(2, 3, 2, 2, 4, 4, seed=58)], i=[2, 2, 0, 0, 8, 8, 1, 1, 0], ref=None))
    add("extract_image_patches", Case([R(1, 6, 6, 2, seed=59)], i=[2, 2, 2, 2, 1, 1, 0], ref=None))
    add("dilation2d", Case([R(1, 6, 6, 2, seed=60), R(2, 2, 2, seed=61)], i=[1, 1, 1, 1, 1, 1, 1, 1, 1], ref=None))
    add("batchnorm", Case([img, R(3, seed=62), R(3, seed=63, lo=0.5, hi=2), R(3, seed=64), R(3, seed=65)], i=[1, 1, 1], t=[1e-5],
                          ref=lambda x, m, v, g, b: (d64(x) - d64(m).view(1, 3, 1, 1)) / torch.sqrt(d64(v).view(1, 3, 1, 1) + 1e-5) * d64(g).view(1, 3, 1, 1) + d64(b).view(1, 3, 1, 1), grad=True, rtol=1e-4))
    add("fused_batch_norm", Case([R(2, 4, 4, 3, seed=66), R(3, seed=64), R(3, seed=65), torch.zeros(3), torch.ones(3)], i=[0, 1], t=[1e-3], ref=None))
    add("layer_norm", Case([x34, R(4, seed=67), R(4, seed=68)], i=[1], ref=lambda x, g, b: F.layer_norm(d64(x), (4,), d64(g), d64(b), 1e-5), grad=True, rtol=1e-4))
    add("lrn", Case([img], t=[1.0, 1e-4, 0.75], i=[2], ref=None, grad=True))
    add("softmax", Case([x34], i=[1], ref=lambda x: torch.exp(d64(x)) / torch.exp(d64(x)).sum(1, keepdim=True), grad=True, bp_out=True))
    add("log_softmax", Case([x34], i=[1], ref=lambda x: d64(x) - torch.log(torch.exp(d64(x)).sum(1, keepdim=True)), grad=True, bp_out=True))
    add("dropout", Case([x34], t=[1.0], i=[0], ref=lambda x: d64(x)))
    add("alpha_dropout_bp", Case([x34, y34], t=[0.5, 1.0, 1.0, 0.0], i=[0]))
    add("dot_product_attention", Case([R(2, 4, 3, seed=69), R(2, 4, 5, seed=70), R(2, 6, 5, seed=71)], i=[1, 0],
                                      ref=lambda q, k, v: d64(v) @ torch.softmax(d64(k).transpose(1, 2) @ d64(q) / 2.0, 1), grad=True, rtol=1e-4))
    add("dot_product_attention_v2", Case([R(2, 3, 4, seed=72), R(2, 5, 4, seed=73), R(2, 5, 4, seed=74)], ref=None))
    add("multi_head_dot_product_attention", Case([R(2, 4, 3, seed=75), R(2, 4, 5, seed=76), R(2, 4, 5, seed=77), R(2, 3, 4, seed=78), R(2, 3, 4, seed=79), R(2, 3, 4, seed=80), R(6, 4, seed=81)], i=[1, 0], ref=None, grad=True, rtol=1e-4))
    n, nIn, B = 5, 4, 2
    add("lstmCell", Case([R(B, nIn, seed=82), R(B, n, seed=83), R(B, n, seed=84), R(nIn, 4 * n, seed=85) * 0.3, R(n, 4 * n, seed=86) * 0.3, torch.zeros(3 * n), torch.eye(n), R(4 * n, seed=87) * 0.1],
                         i=[0, 0], t=[0.0, 0.0, 0.0], ref=None))
    add("lstmLayerCell", Case([R(B, nIn, seed=82), R(nIn, 4 * n, seed=85) * 0.3, R(n, 4 * n, seed=86) * 0.3, R(4 * n, seed=87) * 0.1, R(B, n, seed=83), R(B, n, seed=84)],
                              i=[0, 2, 0, 1], t=[0.0], b=[True, False], ref=None))
    add("lstmLayer", Case([R(3, B, nIn, seed=88), R(nIn, 4 * n, seed=85) * 0.3, R(n, 4 * n, seed=86) * 0.3, R(4 * n, seed=87) * 0.1],
                          i=[0, 0, 2, 0, 1], t=[0.0], b=[True, False, False, False, False, True, True, True], ref=None))
    add("gruCell", Case([R(B, nIn, seed=89), R(B, n, seed=90), R(nIn + n, 2 * n, seed=91) * 0.3, R(nIn + n, n, seed=92) * 0.3, R(2 * n, seed=93) * 0.1, R(n, seed=94) * 0.1], ref=None))
    add("gru", Case([R(3, B, nIn, seed=95), R(B, n, seed=90), R(nIn, 3 * n, seed=96) * 0.3, R(n, 3 * n, seed=97) * 0.3, R(3 * n, seed=98) * 0.1], ref=None, grad=True, gin=[0, 1, 2, 3, 4], rtol=1e-4))
    add("sruCell", Case([R(B, n, seed=99), R(B, n, seed=100), R(n, 3 * n, seed=101) * 0.3, R(2 * n, seed=102) * 0.1], ref=None))
    add("sru", Case([R(B, n, 3, seed=103), R(3 * n, n, seed=104) * 0.3, R(2 * n, seed=105) * 0.1, R(B, n, seed=106), torch.ones(B, n)], ref=None))
    add("static_rnn", Case([R(3, B, nIn, seed=107), R(nIn, n, seed=108) * 0.3, R(n, n, seed=109) * 0.3, R(2 * n, seed=110) * 0.1], ref=None))
    add("dynamic_rnn", Case([R(3, B, nIn, seed=107), R(nIn, n, seed=108) * 0.3, R(n, n, seed=109) * 0.3, R(2 * n, seed=110) * 0.1], i=[1], ref=None))
    # ---- losses (weights = scalar 1, reduction mode in iArgs[0]: 0 none, 1 sum, 2 mean by weight, 3 mean by non-zero weights)
    lab, pred, one = R(3, 4, seed=111, lo=0, hi=1), R(3, 4, seed=112, lo=0.05, hi=0.95), torch.ones(1)
    add("absolute_difference_loss", Case([pred, one, lab], i=[1], ref=lambda p, w, l: (d64(p) - d64(l)).abs().sum(), grad=False))
    add("mean_sqerr_loss", Case([pred, one, lab], i=[1], ref=lambda p, w, l: ((d64(p) - d64(l)) ** 2).sum()))
    add("huber_loss", Case([pred, one, lab], i=[1], t=[0.3], ref=lambda p, w, l: torch.where((d64(p) - d64(l)).abs() <= 0.3, 0.5 * (d64(p) - d64(l)) ** 2, 0.3 * (d64(p) - d64(l)).abs() - 0.5 * 0.09).sum(), rtol=1e-4))
    add("log_loss", Case([pred, one, lab], i=[1], t=[1e-7], ref=lambda p, w, l: -(d64(l) * torch.log(d64(p) + 1e-7) + (1 - d64(l)) * torch.log(1 - d64(p) + 1e-7)).sum(), rtol=1e-4))
    add("hinge_loss", Case([x34, one, (lab > 0.5).float()], i=[1], ref=lambda p, w, l: torch.clamp_min(1 - (2 * d64(l) - 1) * d64(p), 0).sum()))
    add("log_poisson_loss", Case([x34, one, lab], i=[1, 0], ref=lambda p, w, l: (torch.exp(d64(p)) - d64(l) * d64(p)).sum(), rtol=1e-4))
    add("cosine_distance_loss", Case([F.normalize(pred, dim=1), one, F.normalize(lab, dim=1)], i=[1, 1], ref=lambda p, w, l: (1 - (d64(p) * d64(l)).sum(1)).sum(), rtol=1e-4))
    add("mean_pairwssqerr_loss", Case([pred, one, lab], i=[1], ref=None))
    add("sigm_cross_entropy_loss", Case([x34, one, lab], i=[1], t=[0.0], ref=lambda z, w, l: (torch.clamp_min(d64(z), 0) - d64(z) * d64(l) + torch.log1p(torch.exp(-d64(z).abs()))).sum(), rtol=1e-4))
    oh = torch.eye(4)[[0, 2, 3]]
    add("softmax_cross_entropy_loss", Case([x34, one, oh], i=[1], t=[0.0], ref=lambda z, w, l: -(d64(l) * torch.log_softmax(d64(z), 1)).sum(), rtol=1e-4))
    add("softmax_cross_entropy_loss_with_logits", Case([x34, oh], i=[1] if False else [], ref=lambda z, l: -(d64(l) * torch.log_softmax(d64(z), 1)).sum(1), rtol=1e-4))
    add("sparse_softmax_cross_entropy_loss_with_logits", Case([I(0, 2, 3), x34], ref=lambda l, z: -torch.log_softmax(d64(z), 1)[torch.arange(3), l], rtol=1e-4))
    add("weighted_cross_entropy_with_logits", Case([lab, x34, torch.ones(4) * 2], ref=None))
    add("ctc_loss", Case([torch.tensor([[1, 2]]), R(1, 5, 4, seed=113).log_softmax(-1), I(2), I(5)], i=[0], ref=lambda t, lp, tl, il: F.ctc_loss(d64(lp).transpose(0, 1), t, il, tl, blank=0, reduction="none"), rtol=1e-4))
    for nm in ("absolute_difference_loss", "mean_sqerr_loss", "huber_loss", "log_loss", "hinge_loss", "log_poisson_loss", "cosine_distance_loss",
               "mean_pairwssqerr_loss", "sigm_cross_entropy_loss", "softmax_cross_entropy_loss"):
        c0 = C[nm][0]
        add(nm + "_grad", Case(c0.ins, i=c0.i, t=c0.t, ref=None, tag="of:" + nm))
    add("softmax_cross_entropy_loss_with_logits_grad", Case([x34, oh], ref=None))
    add("sparse_softmax_cross_entropy_loss_with_logits_grad", Case([I(0, 2, 3), x34], ref=None))
    add("ctc_loss_grad", Case([torch.tensor([[1, 2]]), R(1, 5, 4, seed=113).log_softmax(-1), I(2), I(5)], i=[0], ref=None))
    # ---- updaters (closed forms of SURVEY appendix B.5)
    gr, z2 = I(0.1, -0.2, dtype=torch.float32), torch.zeros(2)
    add("sgd_updater", Case([gr], t=[0.1], ref=lambda g: 0.1 * d64(g)))
    add("apply_sgd", Case([torch.ones(2), gr], t=[0.1], ref=lambda p, g: 1 - 0.1 * d64(g)))
    add("adam_updater", Case([gr, z2, z2], t=[0.01, 0.9, 0.999, 1e-8], i=[0],
                             ref=lambda g, v, m: [0.1 * d64(g) * (0.01 * math.sqrt(1 - 0.999) / (1 - 0.9)) / (torch.sqrt(0.001 * d64(g) ** 2) + 1e-8), 0.001 * d64(g) ** 2, 0.1 * d64(g)], rtol=1e-4))
    add("rms_prop_updater", Case([gr, z2], t=[0.1, 0.95, 1e-8], ref=lambda g, s: [0.1 * d64(g) / (torch.sqrt(0.05 * d64(g) ** 2) + 1e-8), 0.05 * d64(g) ** 2], rtol=1e-4))
    add("ada_grad_updater", Case([gr, z2], t=[0.1, 1e-6], ref=lambda g, s: [0.1 * d64(g) / (torch.sqrt(d64(g) ** 2) + 1e-6), d64(g) ** 2], rtol=1e-4))
    add("nesterovs_updater", Case([gr, z2], t=[0.1, 0.9], ref=lambda g, v: [-(0 - (1 + 0.9) * (-0.1 * d64(g))) * -1 if False else (1 + 0.9) * 0.1 * d64(g) * 1.0 - 0, -0.1 * d64(g)], rtol=1e-4))
    add("ada_delta_updater", Case([gr, z2, z2], t=[0.95, 1e-6], ref=None))
    add("ada_max_updater", Case([gr, z2, z2], t=[0.002, 0.9, 0.999, 1e-8], i=[0], ref=None))
    add("nadam_updater", Case([gr, z2, z2], t=[0.002, 0.9, 0.999, 1e-8], i=[0], ref=None))
    add("ams_grad_updater", Case([gr, z2, z2, z2], t=[0.001, 0.9, 0.999, 1e-8], i=[0], ref=None))
    add("adabelief_updater", Case([gr, z2, z2], t=[0.001, 0.9, 0.999, 1e-8], i=[0], ref=None))
    # ---- random (statistical properties)
    def stat(mean=None, var=None, lo=None, hi=None, tol=0.08):
        def chk(o, ins):
            v = o[0].double()
            if mean is not None and abs(float(v.mean()) - mean) > tol * max(1.0, abs(mean)):
                return "mean %.4f, expected %.4f" % (float(v.mean()), mean)
            if var is not None and abs(float(v.var()) - var) > 2 * tol * max(1.0, var):
                return "variance %.4f, expected %.4f" % (float(v.var()), var)
            if lo is not None and float(v.min()) < lo:
                return "minimum %.4f below %.4f" % (float(v.min()), lo)
            if hi is not None and float(v.max()) > hi:
                return "maximum %.4f above %.4f" % (float(v.max()), hi)
            return None
        return chk
    shp = I(4000)
    add("randomuniform", Case([shp], t=[1.0, 3.0], check=stat(2.0, 1 / 3, 1.0, 3.0)))
    add("random_normal", Case([shp], t=[1.0, 2.0], check=stat(1.0, 4.0)))
    add("random_bernoulli", Case([shp], t=[0.3], check=stat(0.3, 0.21, 0.0, 1.0)))
    add("random_exponential", Case([shp], t=[2.0], check=stat(0.5, 0.25, 0.0)))
    add("random_gamma", Case([shp, torch.tensor([2.0])], check=stat(2.0, 2.0, 0.0, tol=0.12)))
    add("random_poisson", Case([shp, torch.tensor([3.0])], check=stat(3.0, 3.0, 0.0, tol=0.12)))
    add("random_multinomial", Case([torch.log(torch.tensor([[0.2, 0.8]])), I(2000)], i=[1], check=lambda o, ins: None if abs(float(o[0].double().mean()) - 0.8) < 0.05 else "class-1 frequency %.3f" % float(o[0].double().mean())))
    add("random_shuffle", Case([torch.arange(50.0)], check=lambda o, ins: None if sorted(o[0].tolist()) == list(range(50)) else "not a permutation"))
    add("random_crop", Case([R(6, 6, 3, seed=114), I(4, 4, 3)], check=lambda o, ins: None if list(o[0].shape) == [4, 4, 3] else "shape"))
    add("get_seed", Case([]))
    add("set_seed", Case([I(7)]))
    # ---- image
    rgb = R(2, 4, 4, 3, seed=115, lo=0, hi=1)
    for a, b in (("rgb_to_hsv", "hsv_to_rgb"), ("rgb_to_yiq", "yiq_to_rgb"), ("rgb_to_yuv", "yuv_to_rgb")):
        add(a, Case([rgb], ref=None))
        add(b, Case([rgb], ref=None))
    add("rgb_to_grs", Case([rgb], ref=lambda x: (d64(x) * torch.tensor([0.2989, 0.5870, 0.1140], dtype=torch.float64)).sum(-1, keepdim=True), rtol=1e-3, atol=1e-3))
    add("adjust_contrast", Case([rgb], t=[2.0], ref=lambda x: (d64(x) - d64(x).mean((1, 2), keepdim=True)) * 2 + d64(x).mean((1, 2), keepdim=True)))
    add("adjust_contrast_v2", Case([rgb], t=[2.0], ref=lambda x: (d64(x) - d64(x).mean((1, 2), keepdim=True)) * 2 + d64(x).mean((1, 2), keepdim=True)))
    add("adjust_hue", Case([rgb], t=[0.1], ref=None))
    add("adjust_saturation", Case([rgb], t=[0.5], ref=None))
    for nm in ("resize_bilinear", "resize_nearest_neighbor", "resize_bicubic", "resize_area"):
        add(nm, Case([rgb, I(8, 8, dtype=torch.int32)], b=[False, False], check=lambda o, ins: None if list(o[0].shape) == [2, 8, 8, 3] else "shape %s" % (list(o[0].shape),)))
    add("resize_images", Case([rgb, I(8, 8, dtype=torch.int32)], i=[0], b=[False, False]))
    add("image_resize", Case([rgb, I(8, 8, dtype=torch.int32)], i=[1], b=[False, False]))
    add("crop_and_resize", Case([rgb, torch.tensor([[0.0, 0.0, 1.0, 1.0]]), I(0, dtype=torch.int32), I(4, 4, dtype=torch.int32)], i=[0], t=[0.0],
                                ref=lambda im, bx, bi, sz: d64(im)[0:1], rtol=1e-4, atol=1e-5))
    boxes, scores = torch.tensor([[0, 0, 1, 1], [0, 0, 0.9, 0.9], [2, 2, 3, 3.0]]), I(0.9, 0.8, 0.7, dtype=torch.float32)
    add("non_max_suppression", Case([boxes, scores], i=[3], t=[0.5], ref=lambda b, s: torch.tensor([0, 2])))
    add("non_max_suppression_v3", Case([boxes, scores], i=[3], t=[0.5, 0.0], ref=lambda b, s: torch.tensor([0, 2])))
    add("non_max_suppression_overlaps", Case([torch.tensor([[1.0, 0.9, 0.0], [0.9, 1.0, 0.0], [0.0, 0.0, 1.0]]), scores], i=[3], t=[0.5, 0.0], ref=lambda o, s: torch.tensor([0, 2])))
    add("draw_bounding_boxes", Case([rgb, torch.tensor([[[0.1, 0.1, 0.8, 0.8]], [[0.2, 0.2, 0.6, 0.6]]]), torch.tensor([[1.0, 0.0, 0.0]])], ref=None))
    add("compare_and_bitpack", Case([R(2, 8, seed=116), torch.tensor(0.0)], ref=lambda x, t: torch.tensor([[sum(int(x[r, c] > 0) << (7 - c) for c in range(8))] for r in range(2)], dtype=torch.uint8)))
    add("conv2d_input_bp", Case([I(2, 3, 8, 8), w_khwio, R(2, 5, 8, 8, seed=123)], i=[3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0],
                                ref=lambda s_, w, gy: F.conv_transpose2d(d64(gy), d64(w).permute(3, 2, 0, 1), padding=1), rtol=1e-4, atol=1e-5))
    add("deconv2d_tf", Case([I(1, 5, 8, 8), R(2, 2, 5, 3, seed=52), R(1, 3, 4, 4, seed=51)], i=[2, 2, 2, 2, 0, 0, 1, 1, 0, 0, 0],
                            ref=lambda s_, w, x: F.conv_transpose2d(d64(x), d64(w).permute(3, 2, 0, 1), stride=2), rtol=1e-4, atol=1e-5))
    add("ctc_beam", Case([R(1, 5, 4, seed=113).log_softmax(-1), I(5)], i=[0, 3, 2], b=[True], ref=None))
    # ---- misc
    add("hashcode", Case([x34]))
    add("min_max_datatype", Case([], i=[5, 1]))
    add("knn_mindistance", Case([R(4, seed=117), R(4, seed=118) - 2, R(4, seed=119) + 2], ref=None))
    add("cell_contains", Case([R(2, seed=120), R(2, seed=121, lo=1, hi=2), R(2, seed=122) * 0.1], i=[2]))
    add("barnes_gains", Case([x34.abs(), x34, y34], ref=None))
    add("firas_sparse", Case([x34], i=[1, 2]))
    add("test_scalar", Case([torch.tensor([1.0])]))
    add("testop2i2o", Case([x34, y34]))
    add("testreduction", Case([x34], i=[1]))
    add("testcustom", Case([x34]))
    add("test_output_reshape", Case([x34]))
    add("print_variable", Case([x34]))
    add("print_affinity", Case([x34]))
    add("expose", Case([x34]))
    from .op_cases_extra import extend
    extend(C, add)
    return C


# ------------------------------------------------------------------------------------------------ multi-op scenarios
def _list_scenario(device):
    """TensorArray ops as one sequence (reference: ListOperationsTests.cpp / TensorArray in ShapeOpValidation)."""
    from ..nd4j.ops.custom import exec_op
    a, b, c = (torch.full((2,), float(v), device=device) for v in (1, 2, 3))
    lst = exec_op("create_list", [], iArgs=[0, 1])[0]
    for k, v in enumerate((a, b, c)):
        lst = exec_op("write_list", [lst, v], iArgs=[k])[0]
    if int(exec_op("size_list", [lst])[0]) != 3:
        return "size_list"
    if not torch.equal(exec_op("read_list", [lst], iArgs=[1])[0], b):
        return "read_list"
    if not torch.equal(exec_op("stack_list", [lst])[0], torch.stack([a, b, c])):
        return "stack_list"
    if not torch.equal(exec_op("gather_list", [lst, torch.tensor([2, 0])])[0], torch.stack([c, a])):
        return "gather_list"
    if not torch.equal(exec_op("pick_list", [lst], iArgs=[1, 2])[0], torch.stack([b, c])):
        return "pick_list"
    cl = exec_op("clone_list", [lst])[0]
    lst = exec_op("delete_list", [lst], iArgs=[0])[0]
    if not torch.equal(exec_op("stack_list", [cl])[0], torch.stack([a, b, c])) or torch.equal(exec_op("stack_list", [lst])[0][0], a):
        return "clone_list / delete_list"
    ul = exec_op("unstack_list", [torch.stack([a, b])])[0]
    if not torch.equal(exec_op("read_list", [ul], iArgs=[1])[0], b):
        return "unstack_list"
    sl = exec_op("scatter_list", [exec_op("create_list", [], iArgs=[0, 1])[0], torch.tensor([1, 0]), torch.stack([a, b])])[0]
    if not torch.equal(exec_op("read_list", [sl], iArgs=[0])[0], b):
        return "scatter_list"
    sp = exec_op("split_list", [exec_op("create_list", [], iArgs=[0, 1])[0], torch.arange(5.0, device=device), torch.tensor([2, 3])])[0]
    if exec_op("read_list", [sp], iArgs=[1])[0].tolist() != [2.0, 3.0, 4.0]:
        return "split_list"
    return None


SCENARIOS = [
    ({"create_list", "write_list", "read_list", "stack_list", "size_list", "gather_list", "pick_list", "clone_list", "delete_list",
      "unstack_list", "scatter_list", "split_list"}, _list_scenario),
]

# ops whose meaning only exists inside a graph / a training loop: validated by the named tests, not by a single-op case
EXTERNAL = {
    "Switch": "tests/test_samediff_cpu.py (ifCond / whileLoop frames)", "While": "tests/test_samediff_cpu.py", "Scope": "tests/test_samediff_cpu.py",
    "Conditional": "tests/test_samediff_cpu.py", "Return": "tests/test_samediff_cpu.py", "invoke": "tests/test_samediff_cpu.py",
    "skipgram": "tests/test_nlp_cpu.py (Word2Vec skip-gram training)", "cbow": "tests/test_nlp_cpu.py", "skipgram_inference": "tests/test_nlp_cpu.py",
    "cbow_inference": "tests/test_nlp_cpu.py", "barnes_edge_forces": "tests/test_helpers2_gpu.py", "barnes_symmetrized": "tests/test_helpers2_gpu.py",
}
