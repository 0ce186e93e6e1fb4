"""Legacy op families: opNum tables (ABI) + executioner.

Reference: libnd4j/include/loops/legacy_ops.h:26-132 (opNum tables, SURVEY Appendix E — numbers have
gaps and must not be renumbered), legacy/cuda/NativeOpExecutioner.cu (exec* entry points),
nd4j-api linalg/api/ops/executioner/OpExecutioner.java.

GPU tensors run on the hand-written gfx950 kernels in native/csrc/legacy_ops.hip (strides passed by
value; broadcast = zero strides; two-stage reductions). CPU tensors use the PyTorch formulation of
the same functor — it is also the fp32/fp64 oracle the GPU numerics tests compare against.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Sequence

import torch

from ...native import lib as L

# ----------------------------------------------------------------------------- opNum tables (verbatim ABI)
PAIRWISE = {0: "Add", 1: "CopyPws", 2: "Divide", 3: "Multiply", 4: "Pow", 5: "ReverseSubtract", 6: "Subtract",
            7: "MaxPairwise", 8: "MinPairwise", 9: "Copy2", 10: "Axpy", 11: "ReverseDivide", 12: "CompareAndSet",
            13: "CompareAndReplace", 14: "Remainder", 15: "FMod", 16: "Atan2", 17: "TruncateDiv", 18: "FloorDiv",
            19: "FloorMod", 20: "SquaredSubtract", 21: "ReverseMod", 22: "SafeDivide", 23: "Mod", 24: "RelativeError",
            25: "BinaryRelativeError", 26: "BinaryMinimumAbsoluteRelativeError", 27: "LogicalOr", 28: "LogicalXor",
            29: "LogicalNot", 30: "LogicalAnd", 31: "PowDerivative", 32: "LogPoissonLoss", 33: "LogPoissonLossFull",
            34: "AMaxPairwise", 35: "AMinPairwise", 36: "TruncateMod", 37: "ReplaceNans", 38: "DivideNoNan",
            39: "IGamma", 40: "IGammac"}
PAIRWISE_BOOL = {0: "EqualTo", 1: "GreaterThan", 2: "LessThan", 3: "Epsilon", 4: "GreaterThanOrEqual",
                 5: "MatchCondition", 6: "NotEqualTo", 7: "And", 8: "Or", 9: "Xor", 10: "Not", 11: "LessThanOrEqual"}
BROADCAST_BOOL = SCALAR_BOOL = PAIRWISE_BOOL
PAIRWISE_INT = {0: "ShiftLeft", 1: "ShiftRight", 2: "CyclicShiftLeft", 3: "CyclicShiftRight", 4: "IntAnd", 5: "IntOr",
                6: "IntXor"}
BROADCAST_INT = SCALAR_INT = PAIRWISE_INT
BROADCAST = {0: "Add", 1: "Subtract", 2: "Multiply", 3: "Divide", 4: "ReverseDivide", 5: "ReverseSubtract", 6: "CopyPws",
             7: "Pow", 13: "MinPairwise", 14: "MaxPairwise", 15: "AMinPairwise", 16: "AMaxPairwise", 17: "SquaredSubtract",
             18: "FloorMod", 19: "FloorDiv", 20: "ReverseMod", 21: "SafeDivide", 22: "Mod", 23: "TruncateDiv", 26: "Atan2",
             27: "LogicalOr", 28: "LogicalXor", 29: "LogicalNot", 30: "LogicalAnd", 31: "DivideNoNan", 32: "IGamma",
             33: "IGammac", 34: "PowDerivative"}
SCALAR = {0: "Add", 1: "Subtract", 2: "Multiply", 3: "Divide", 4: "ReverseDivide", 5: "ReverseSubtract", 6: "MaxPairwise",
          7: "ELU", 8: "ELUDerivative", 13: "MinPairwise", 14: "CopyPws", 15: "Mod", 16: "ReverseMod", 17: "Remainder",
          18: "FMod", 19: "TruncateDiv", 20: "FloorDiv", 21: "FloorMod", 22: "SquaredSubtract", 23: "SafeDivide",
          24: "AMaxPairwise", 25: "AMinPairwise", 26: "Atan2", 27: "LogicalOr", 28: "LogicalXor", 29: "LogicalNot",
          30: "LogicalAnd", 31: "Pow", 32: "PowDerivative", 33: "CompareAndSet", 34: "SXELogitsSmoother", 35: "LeakyRELU",
          36: "LeakyRELUDerivative", 37: "ReplaceNans", 38: "LogX", 39: "RELU", 40: "RELU6", 41: "Step", 42: "LstmClip",
          43: "TruncateMod", 44: "SquaredReverseSubtract", 45: "ReversePow", 46: "DivideNoNan", 47: "IGamma",
          48: "IGammac", 49: "RELUDerivative"}
TRANSFORM_SAME = {0: "Abs", 1: "Sign", 2: "Ones", 3: "Neg", 4: "Round", 5: "TimesOneMinus", 6: "Cube", 7: "OneMinus",
                  11: "Reciprocal", 12: "Square", 13: "CompareAndSetTransform", 15: "Identity", 17: "Ceiling", 18: "Floor",
                  19: "ClipByValue", 21: "Copy"}
TRANSFORM_ANY = {0: "Assign"}
TRANSFORM_FLOAT = {1: "Sqrt", 3: "RSqrt"}
TRANSFORM_BOOL = {1: "IsInf", 2: "IsNan", 3: "IsFinite", 4: "IsInfOrNan", 5: "MatchConditionBool", 6: "IsPositive",
                  7: "Not", 8: "IsNegative"}
TRANSFORM_STRICT = {2: "ScaledTanh", 3: "Affine", 4: "TanhDerivative", 5: "HardTanhDerivative", 6: "SigmoidDerivative",
                    7: "SoftSignDerivative", 8: "TanDerivative", 9: "SELUDerivative", 10: "HardSigmoidDerivative",
                    11: "RationalTanhDerivative", 12: "RectifiedTanhDerivative", 13: "SwishDerivative",
                    14: "ACoshDerivative", 15: "ASinhDerivative", 16: "SinhDerivative", 17: "LogSigmoidDerivative",
                    18: "SpecialDerivative", 19: "Stabilize", 20: "StabilizeFP16", 21: "CubeDerivative", 22: "Cosine",
                    23: "Exp", 24: "Log", 25: "SetRange", 26: "Sigmoid", 27: "Sin", 28: "SoftPlus", 29: "Tanh", 30: "ACos",
                    31: "ASin", 32: "ATan", 33: "HardTanh", 34: "SoftSign", 36: "HardSigmoid", 37: "RationalTanh",
                    38: "RectifiedTanh", 39: "Sinh", 40: "Cosh", 41: "Tan", 42: "SELU", 43: "Swish", 44: "Log1p", 45: "Erf",
                    46: "ACosh", 47: "ASinh", 48: "Rint", 49: "LogSigmoid", 50: "Erfc", 51: "Expm1", 52: "ATanh", 53: "GELU",
                    54: "GELUDerivative", 55: "PreciseGELU", 56: "PreciseGELUDerivative", 57: "Mish", 58: "MishDerivative"}
REDUCE_FLOAT = {0: "Mean", 1: "AMean", 2: "Norm1", 3: "Norm2", 4: "NormMax", 5: "NormFrobenius", 6: "NormP",
                7: "SquaredNorm", 8: "Entropy", 9: "LogEntropy", 10: "ShannonEntropy"}
REDUCE_SAME = {0: "Sum", 1: "Max", 2: "Min", 3: "Prod", 4: "ASum", 5: "AMax", 6: "AMin"}
REDUCE_BOOL = {0: "Any", 1: "All", 2: "IsFinite", 3: "IsInfOrNan", 4: "IsNan", 5: "IsInf", 6: "IsPositive", 7: "IsNegative"}
REDUCE_LONG = {0: "CountNonZero", 1: "CountZero", 2: "MatchCondition"}
REDUCE3 = {0: "ManhattanDistance", 1: "EuclideanDistance", 2: "CosineSimilarity", 3: "Dot", 4: "EqualsWithEps",
           5: "CosineDistance", 6: "JaccardDistance", 7: "SimpleHammingDistance"}
INDEX_REDUCE = {0: "IndexMax", 1: "IndexMin", 2: "IndexAbsoluteMax", 3: "IndexAbsoluteMin", 4: "FirstIndex", 5: "LastIndex"}
SUMMARY_STATS = {0: "Variance", 1: "StandardDeviation"}
RANDOM = {0: "UniformDistribution", 1: "DropOut", 2: "DropOutInverted", 3: "ProbablisticMerge", 4: "Linspace", 5: "Choice",
          6: "GaussianDistribution", 7: "BernoulliDistribution", 8: "BinomialDistribution", 9: "BinomialDistributionEx",
          10: "LogNormalDistribution", 11: "TruncatedNormalDistribution", 12: "AlphaDropOut", 13: "ExponentialDistribution",
          14: "ExponentialDistributionInv", 15: "PoissonDistribution", 16: "GammaDistribution"}
AGGREGATE = {0: "HierarchicSoftmax", 1: "Dot", 2: "Axpy", 3: "SkipGram", 4: "CBOW", 5: "GEMM"}

FAMILIES = {"pairwise": PAIRWISE, "pairwise_bool": PAIRWISE_BOOL, "pairwise_int": PAIRWISE_INT, "broadcast": BROADCAST,
            "broadcast_bool": BROADCAST_BOOL, "broadcast_int": BROADCAST_INT, "scalar": SCALAR, "scalar_bool": SCALAR_BOOL,
            "scalar_int": SCALAR_INT, "transform_same": TRANSFORM_SAME, "transform_any": TRANSFORM_ANY,
            "transform_float": TRANSFORM_FLOAT, "transform_bool": TRANSFORM_BOOL, "transform_strict": TRANSFORM_STRICT,
            "reduce_float": REDUCE_FLOAT, "reduce_same": REDUCE_SAME, "reduce_bool": REDUCE_BOOL, "reduce_long": REDUCE_LONG,
            "reduce3": REDUCE3, "index_reduce": INDEX_REDUCE, "summary_stats": SUMMARY_STATS, "random": RANDOM,
            "aggregate": AGGREGATE}


def op_num(family: str, name: str) -> int:
    for k, v in FAMILIES[family].items():
        if v == name:
            return k
    raise KeyError("%s has no op %s" % (family, name))


# ----------------------------------------------------------------------------- functor ids (legacy_ops.hip enums)
_BIN_ORDER = ["Add", "Subtract", "Multiply", "Divide", "ReverseSubtract", "ReverseDivide", "CopyPws", "Pow", "MaxPairwise",
              "MinPairwise", "AMaxPairwise", "AMinPairwise", "Axpy", "CompareAndSet", "CompareAndReplace", "Remainder",
              "FMod", "Atan2", "TruncateDiv", "FloorDiv", "FloorMod", "SquaredSubtract", "ReverseMod", "SafeDivide", "Mod",
              "RelativeError", "BinaryRelativeError", "BinaryMinimumAbsoluteRelativeError", "LogicalOr", "LogicalXor",
              "LogicalNot", "LogicalAnd", "PowDerivative", "LogPoissonLoss", "LogPoissonLossFull", "TruncateMod",
              "ReplaceNans", "DivideNoNan", "IGamma", "IGammac",
              "ELU", "ELUDerivative", "SXELogitsSmoother", "LeakyRELU", "LeakyRELUDerivative", "LogX", "RELU", "RELU6",
              "Step", "LstmClip", "SquaredReverseSubtract", "ReversePow", "RELUDerivative",
              "EqualTo", "GreaterThan", "LessThan", "Epsilon", "GreaterThanOrEqual", "MatchCondition", "NotEqualTo", "And",
              "Or", "Xor", "Not", "LessThanOrEqual",
              "ShiftLeft", "ShiftRight", "CyclicShiftLeft", "CyclicShiftRight", "IntAnd", "IntOr", "IntXor"]
BIN = {n: i for i, n in enumerate(_BIN_ORDER)}
BIN["Copy2"] = BIN["CopyPws"]
_UN_ORDER = ["Abs", "Sign", "Ones", "Neg", "Round", "TimesOneMinus", "Cube", "OneMinus", "Reciprocal", "Square",
             "CompareAndSetTransform", "Identity", "Ceiling", "Floor", "ClipByValue", "Copy", "Sqrt", "RSqrt",
             "IsInf", "IsNan", "IsFinite", "IsInfOrNan", "MatchConditionBool", "IsPositive", "Not", "IsNegative",
             "ScaledTanh", "Affine", "TanhDerivative", "HardTanhDerivative", "SigmoidDerivative", "SoftSignDerivative",
             "TanDerivative", "SELUDerivative", "HardSigmoidDerivative", "RationalTanhDerivative",
             "RectifiedTanhDerivative", "SwishDerivative", "ACoshDerivative", "ASinhDerivative", "SinhDerivative",
             "LogSigmoidDerivative", "SpecialDerivative", "Stabilize", "CubeDerivative", "Cosine", "Exp", "Log", "SetRange",
             "Sigmoid", "Sin", "SoftPlus", "Tanh", "ACos", "ASin", "ATan", "HardTanh", "SoftSign", "HardSigmoid",
             "RationalTanh", "RectifiedTanh", "Sinh", "Cosh", "Tan", "SELU", "Swish", "Log1p", "Erf", "ACosh", "ASinh",
             "Rint", "LogSigmoid", "Erfc", "Expm1", "ATanh", "GELU", "GELUDerivative", "PreciseGELU",
             "PreciseGELUDerivative", "Mish", "MishDerivative"]
UN = {n: i for i, n in enumerate(_UN_ORDER)}
UN["StabilizeFP16"] = UN["Stabilize"]
UN["Assign"] = UN["Copy"]
_RED_ORDER = ["Mean", "AMean", "Norm1", "Norm2", "NormMax", "NormFrobenius", "NormP", "SquaredNorm", "Entropy", "LogEntropy",
              "ShannonEntropy", "Sum", "Max", "Min", "Prod", "ASum", "AMax", "AMin", "Any", "All", "IsFinite", "IsInfOrNan",
              "IsNan", "IsInf", "IsPositive", "IsNegative", "CountNonZero", "CountZero", "MatchCondition",
              "ManhattanDistance", "EuclideanDistance", "CosineSimilarity", "Dot", "EqualsWithEps", "CosineDistance",
              "JaccardDistance", "SimpleHammingDistance", "Variance", "StandardDeviation", "IndexMax", "IndexMin",
              "IndexAbsoluteMax", "IndexAbsoluteMin", "FirstIndex", "LastIndex"]
RED = {n: i for i, n in enumerate(_RED_ORDER)}

MAXR = 8
_FORCE_TORCH = [False]


class force_torch:
    """Context manager: run legacy ops through their ATen formulation even on GPU tensors."""

    def __enter__(self):
        self._prev = _FORCE_TORCH[0]
        _FORCE_TORCH[0] = True

    def __exit__(self, *exc):
        _FORCE_TORCH[0] = self._prev
        return False


class _Strides3(ctypes.Structure):
    _fields_ = [("rank", ctypes.c_int), ("contiguous", ctypes.c_int), ("shape", ctypes.c_int64 * MAXR),
                ("sx", ctypes.c_int64 * MAXR), ("sy", ctypes.c_int64 * MAXR), ("sz", ctypes.c_int64 * MAXR)]


class _TadDesc(ctypes.Structure):
    _fields_ = [("irank", ctypes.c_int), ("orank", ctypes.c_int),
                ("ishape", ctypes.c_int64 * MAXR), ("istride", ctypes.c_int64 * MAXR), ("iystride", ctypes.c_int64 * MAXR),
                ("oshape", ctypes.c_int64 * MAXR), ("ostride", ctypes.c_int64 * MAXR), ("oystride", ctypes.c_int64 * MAXR),
                ("tad_len", ctypes.c_int64), ("num_tads", ctypes.c_int64)]


def _coalesce(shape, *strides):
    """Drop unit dims and merge dims that are contiguous in every operand (c-order iteration)."""
    dims = [(s, tuple(st[i] for st in strides)) for i, s in enumerate(shape) if s != 1]
    if not dims:
        return [1], [[0] for _ in strides]
    out = [dims[0]]
    for s, st in dims[1:]:
        ps, pst = out[-1]
        if all(pst[k] == st[k] * s for k in range(len(strides))):
            out[-1] = (ps * s, st)
        else:
            out.append((s, st))
    return [d[0] for d in out], [[d[1][k] for d in out] for k in range(len(strides))]


def _strides3(x, y, z):
    shape = list(z.shape)
    xs = list(x.expand(z.shape).stride()) if x is not None else [0] * len(shape)
    ys = list(y.expand(z.shape).stride()) if y is not None else [0] * len(shape)
    zs = list(z.stride())
    cs, (cx, cy, cz) = _coalesce(shape, xs, ys, zs)
    if len(cs) > MAXR:
        return None
    s = _Strides3()
    s.rank = len(cs)
    s.contiguous = int(len(cs) == 1 and cx[0] == 1 and cz[0] == 1 and (y is None or cy[0] == 1))
    for i, v in enumerate(cs):
        s.shape[i], s.sx[i], s.sy[i], s.sz[i] = v, cx[i], cy[i], cz[i]
    return s


def _extra(e):
    arr = (ctypes.c_double * 4)(0, 0, 0, 0)
    if e is not None:
        for i, v in enumerate(list(e)[:4]):
            arr[i] = float(v)
    return arr


_FLOAT_T = (torch.float32, torch.float64, torch.bfloat16, torch.float16)


# ----------------------------------------------------------------------------- CPU (PyTorch) formulations
def _match(x, e):
    cmp, eps, mode = (list(e) + [0, 0, 0])[:3] if e is not None else (0, 0, 0)
    mode = int(mode)
    fx = x.double() if not x.is_floating_point() else x
    return [lambda: (fx - cmp).abs() <= eps, lambda: (fx - cmp).abs() > eps, lambda: fx < cmp, lambda: fx > cmp,
            lambda: fx <= cmp, lambda: fx >= cmp, lambda: fx.abs() < cmp, lambda: fx.abs() > cmp, lambda: torch.isinf(fx),
            lambda: torch.isnan(fx), lambda: fx == cmp, lambda: fx != cmp, lambda: fx.abs() >= cmp, lambda: fx.abs() <= cmp,
            lambda: torch.isfinite(fx), lambda: ~torch.isfinite(fx)][mode]()


def _relerr(x, y):
    d = (x - y).abs() / (x.abs() + y.abs())
    return torch.where((x == 0) & (y == 0), torch.zeros_like(d), d)


def _imod(a, b):
    bi = b.long()
    return torch.fmod(a.long(), torch.where(bi == 0, torch.ones_like(bi), bi))


def _cshift(x, y, left):
    v = x.long() & 0xFFFFFFFF
    s = y.long() & 31
    r = ((v << s) | (v >> ((32 - s) & 31))) if left else ((v >> s) | (v << ((32 - s) & 31)))
    r = r & 0xFFFFFFFF
    return torch.where(r >= 2 ** 31, r - 2 ** 32, r)


def _bin_cpu(name, x, y, e):
    T = torch
    one, zero = torch.ones_like(x), torch.zeros_like(x)
    b = lambda m: m.to(x.dtype)
    ev = list(e) if e is not None else [0.0, 0.0, 0.0, 0.0]
    tbl = {
        "Add": lambda: x + y, "Subtract": lambda: x - y, "Multiply": lambda: x * y, "Divide": lambda: x / y,
        "ReverseSubtract": lambda: y - x, "ReverseDivide": lambda: y / x, "CopyPws": lambda: y + zero, "Copy2": lambda: y + zero,
        "Pow": lambda: T.pow(x, y), "ReversePow": lambda: T.pow(y, x), "MaxPairwise": lambda: T.maximum(x, y),
        "MinPairwise": lambda: T.minimum(x, y), "AMaxPairwise": lambda: T.where(x.abs() > y.abs(), x, y),
        "AMinPairwise": lambda: T.where(x.abs() < y.abs(), x, y), "Axpy": lambda: ev[0] * x + y,
        "CompareAndSet": lambda: T.where(_match(x, e), y, x), "CompareAndReplace": lambda: T.where(_match(y, e), y, x),
        "Remainder": lambda: T.remainder(x, y) if False else x - y * T.round(x / y), "FMod": lambda: T.fmod(x, y),
        "Atan2": lambda: T.atan2(x, y), "TruncateDiv": lambda: T.trunc(x / y), "FloorDiv": lambda: T.floor(x / y),
        "FloorMod": lambda: T.remainder(x, y), "SquaredSubtract": lambda: (x - y) ** 2,
        "SquaredReverseSubtract": lambda: (y - x) ** 2, "ReverseMod": lambda: _imod(y, x).to(x.dtype),
        "SafeDivide": lambda: T.where(y == 0, zero, x / T.where(y == 0, one, y)), "Mod": lambda: _imod(x, y).to(x.dtype),
        "TruncateMod": lambda: _imod(x, y).to(x.dtype), "RelativeError": lambda: _relerr(x, y),
        "BinaryRelativeError": lambda: b(_relerr(x, y) > ev[0]),
        "BinaryMinimumAbsoluteRelativeError": lambda: b((_relerr(x, y) > ev[0]) & ~((x - y).abs() < ev[1])),
        "LogicalOr": lambda: b((x != 0) | (y != 0)), "LogicalXor": lambda: b((x != 0) ^ (y != 0)),
        "LogicalNot": lambda: b(((x != 0) & (y == 0)) | (x == 0)), "LogicalAnd": lambda: b((x != 0) & (y != 0)),
        "PowDerivative": lambda: y * T.pow(x, y - 1), "LogPoissonLoss": lambda: T.exp(y) - x * y,
        "LogPoissonLossFull": lambda: T.exp(y) - x * y + (x * T.log(x) - x + 0.5 * T.log(2 * math.pi * x)),
        "ReplaceNans": lambda: T.where(T.isnan(x), y + zero, x), "DivideNoNan": lambda: T.where(y == 0, zero, x / T.where(y == 0, one, y)),
        "IGamma": lambda: T.special.gammainc(x, y + zero), "IGammac": lambda: T.special.gammaincc(x, y + zero),
        "ELU": lambda: T.where(x >= 0, x, y * (T.exp(x) - 1)), "ELUDerivative": lambda: T.where(x >= 0, one, y * T.exp(x)),
        "SXELogitsSmoother": lambda: x * (1 - y) + 0.5 * y, "LeakyRELU": lambda: T.where(x < 0, y * x, x),
        "LeakyRELUDerivative": lambda: T.where(x >= 0, one, y + zero), "LogX": lambda: T.log(x) / T.log(y + zero),
        "RELU": lambda: T.maximum(x, y + zero), "RELU6": lambda: T.clamp(T.maximum(x, y + zero), max=6),
        "Step": lambda: b(x > y), "LstmClip": lambda: T.maximum(T.minimum(x, y + zero), -(y + zero)),
        "RELUDerivative": lambda: b(x > y),
        "EqualTo": lambda: x == y, "GreaterThan": lambda: x > y, "LessThan": lambda: x < y,
        "Epsilon": lambda: (x.double() - y.double()).abs() <= (ev[0] if ev[0] > 0 else 1e-5), "GreaterThanOrEqual": lambda: x >= y,
        "MatchCondition": lambda: _match(x, e), "NotEqualTo": lambda: x != y, "And": lambda: (x != 0) & (y != 0),
        "Or": lambda: (x != 0) | (y != 0), "Xor": lambda: (x != 0) ^ (y != 0), "Not": lambda: x == 0,
        "LessThanOrEqual": lambda: x <= y,
        "ShiftLeft": lambda: (x.long() << (y.long() & 63)).to(x.dtype), "ShiftRight": lambda: (x.long() >> (y.long() & 63)).to(x.dtype),
        "CyclicShiftLeft": lambda: _cshift(x, y, True).to(x.dtype), "CyclicShiftRight": lambda: _cshift(x, y, False).to(x.dtype),
        "IntAnd": lambda: x & y, "IntOr": lambda: x | y, "IntXor": lambda: x ^ y,
    }
    return tbl[name]()


def _un_cpu(name, x, e):
    T = torch
    ev = list(e) if e is not None else [0.0, 0.0, 0.0, 0.0]
    SA, SL = 1.6732632423543772, 1.0507009873554805
    sig = T.sigmoid
    b = lambda m: m

    def rat_tanh():
        d = x * (2.0 / 3.0)
        return 1.7159 * T.sign(d) * (1 - 1 / (1 + d.abs() + d * d + 1.41645 * d ** 4))

    def rat_tanh_d():
        d = x * (2.0 / 3.0)
        den = 1 + d.abs() + d * d + 1.41645 * d ** 4
        sg = T.where(d < 0, -T.ones_like(d), T.ones_like(d))
        return 1.7159 * (2.0 / 3.0) * (1 + sg * (2 * d + 4 * 1.41645 * d ** 3)) / (den * den)

    def setrange():
        mn, mx = ev[0], ev[1]
        inside = (x >= mn) & (x <= mx)
        alt = T.floor((sig(x) if (mn == 0 and mx == 1) else x) * (mx - mn)) + mn
        return T.where(inside, x, alt)

    u = 0.7978845608028654 * (x + 0.044715 * x ** 3) if x.is_floating_point() else None
    sp = T.nn.functional.softplus(x) if x.is_floating_point() else None
    tbl = {
        "Abs": lambda: x.abs(), "Sign": lambda: T.sign(x), "Ones": lambda: T.ones_like(x), "Neg": lambda: -x,
        "Round": lambda: T.where(x >= 0, T.floor(x + 0.5), T.ceil(x - 0.5)), "TimesOneMinus": lambda: x * (1 - x),
        "Cube": lambda: x ** 3, "OneMinus": lambda: 1 - x, "Reciprocal": lambda: 1 / x, "Square": lambda: x * x,
        "CompareAndSetTransform": lambda: T.where(_match(x, e), T.full_like(x, ev[3]), x), "Identity": lambda: x.clone(),
        "Copy": lambda: x.clone(), "Assign": lambda: x.clone(), "Ceiling": lambda: T.ceil(x), "Floor": lambda: T.floor(x),
        "ClipByValue": lambda: T.clamp(x, ev[0], ev[1]), "Sqrt": lambda: T.sqrt(x), "RSqrt": lambda: T.rsqrt(x),
        "IsInf": lambda: T.isinf(x), "IsNan": lambda: T.isnan(x), "IsFinite": lambda: T.isfinite(x),
        "IsInfOrNan": lambda: ~T.isfinite(x), "MatchConditionBool": lambda: _match(x, e), "IsPositive": lambda: x > 0,
        "Not": lambda: x == 0, "IsNegative": lambda: x < 0,
        "ScaledTanh": lambda: ev[0] * T.tanh(ev[1] * x), "Affine": lambda: ev[0] * x + ev[1],
        "TanhDerivative": lambda: 1 - T.tanh(x) ** 2, "HardTanhDerivative": lambda: ((x >= -1) & (x <= 1)).to(x.dtype),
        "SigmoidDerivative": lambda: sig(x) * (1 - sig(x)), "SoftSignDerivative": lambda: 1 / (1 + x.abs()) ** 2,
        "TanDerivative": lambda: 1 / T.cos(x) ** 2, "SELUDerivative": lambda: T.where(x > 0, T.full_like(x, SL), SL * SA * T.exp(x)),
        "HardSigmoidDerivative": lambda: T.where((x < -2.5) | (x > 2.5), T.zeros_like(x), T.full_like(x, 0.2)),
        "RationalTanhDerivative": rat_tanh_d,
        "RectifiedTanhDerivative": lambda: T.where(x > 0, 1 - T.tanh(x) ** 2, T.zeros_like(x)),
        "SwishDerivative": lambda: sig(x) + x * sig(x) * (1 - sig(x)),
        "ACoshDerivative": lambda: 1 / (T.sqrt(x - 1) * T.sqrt(x + 1)), "ASinhDerivative": lambda: 1 / T.sqrt(x * x + 1),
        "SinhDerivative": lambda: T.cosh(x), "LogSigmoidDerivative": lambda: 1 / (T.exp(x) + 1),
        "SpecialDerivative": lambda: x * (1 - x),
        "Stabilize": lambda: T.where(x * ev[0] > 88, 88.0 / ev[0], T.where(x * ev[0] < -88, -88.0 / ev[0], x)),
        "StabilizeFP16": lambda: T.where(x * ev[0] > 88, 88.0 / ev[0], T.where(x * ev[0] < -88, -88.0 / ev[0], x)),
        "CubeDerivative": lambda: 3 * x * x, "Cosine": lambda: T.cos(x), "Exp": lambda: T.exp(x), "Log": lambda: T.log(x),
        "SetRange": setrange, "Sigmoid": lambda: sig(x), "Sin": lambda: T.sin(x), "SoftPlus": lambda: sp, "Tanh": lambda: T.tanh(x),
        "ACos": lambda: T.acos(x), "ASin": lambda: T.asin(x), "ATan": lambda: T.atan(x), "HardTanh": lambda: T.clamp(x, -1, 1),
        "SoftSign": lambda: x / (1 + x.abs()), "HardSigmoid": lambda: T.clamp(0.2 * x + 0.5, 0, 1), "RationalTanh": rat_tanh,
        "RectifiedTanh": lambda: T.clamp_min(T.tanh(x), 0), "Sinh": lambda: T.sinh(x), "Cosh": lambda: T.cosh(x),
        "Tan": lambda: T.tan(x), "SELU": lambda: T.where(x > 0, SL * x, SL * SA * (T.exp(x) - 1)), "Swish": lambda: x * sig(x),
        "Log1p": lambda: T.log1p(x), "Erf": lambda: T.erf(x), "ACosh": lambda: T.acosh(x), "ASinh": lambda: T.asinh(x),
        "Rint": lambda: T.round(x), "LogSigmoid": lambda: T.nn.functional.logsigmoid(x), "Erfc": lambda: T.erfc(x),
        "Expm1": lambda: T.expm1(x), "ATanh": lambda: T.atanh(x), "GELU": lambda: x * sig(1.702 * x),
        "GELUDerivative": lambda: sig(1.702 * x) + 1.702 * x * sig(1.702 * x) * (1 - sig(1.702 * x)),
        "PreciseGELU": lambda: 0.5 * x * (1 + T.tanh(u)),
        "PreciseGELUDerivative": lambda: 0.5 * (1 + T.tanh(u)) + 0.5 * x * (1 - T.tanh(u) ** 2) * 0.7978845608028654 * (1 + 3 * 0.044715 * x * x),
        "Mish": lambda: x * T.tanh(sp), "MishDerivative": lambda: T.tanh(sp) + x * (1 - T.tanh(sp) ** 2) * sig(x),
    }
    return tbl[name]()


def _norm_dims(x, dims):
    if dims is None or (isinstance(dims, (list, tuple)) and len(dims) == 0):
        return list(range(x.dim()))
    if isinstance(dims, int):
        dims = [dims]
    return sorted(set(d % max(x.dim(), 1) for d in dims))


def _red_cpu(name, x, dims, e, y=None):
    T = torch
    ev = list(e) if e is not None else [0.0, 0.0, 0.0, 0.0]
    d = tuple(dims)
    if x.dim() == 0:
        x = x.reshape(1)
        y = None if y is None else y.reshape(1)
        d = (0,)
    fx = x if x.is_floating_point() else x.double()
    n = 1
    for k in d:
        n *= x.shape[k]
    xlogx = lambda v: T.where(v > 0, v * T.log(v.clamp_min(1e-300)), T.zeros_like(v))

    def flat_inner(t):
        keep = [k for k in range(t.dim()) if k not in d]
        return t.permute(*keep, *d).reshape([t.shape[k] for k in keep] + [-1])

    def idx(fn, v):
        return fn(flat_inner(v), dim=-1)

    def first_last(last):
        m = flat_inner(_match(x, e))
        L_ = m.shape[-1]
        ar = T.arange(L_, device=x.device).expand_as(m)
        if last:
            r = T.where(m, ar, T.full_like(ar, -1)).amax(-1)
        else:
            r = T.where(m, ar, T.full_like(ar, L_)).amin(-1)
            r = T.where(r == L_, T.full_like(r, -1), r)
        return r

    tbl = {
        "Mean": lambda: fx.mean(d), "AMean": lambda: fx.abs().mean(d), "Norm1": lambda: fx.abs().sum(d),
        "Norm2": lambda: T.sqrt((fx * fx).sum(d)), "NormFrobenius": lambda: T.sqrt((fx * fx).sum(d)),
        "NormMax": lambda: fx.abs().amax(d), "NormP": lambda: (fx.abs() ** ev[0]).sum(d) ** (1.0 / ev[0]),
        "SquaredNorm": lambda: (fx * fx).sum(d), "Entropy": lambda: -xlogx(fx).sum(d),
        "LogEntropy": lambda: T.log(-xlogx(fx).sum(d)),
        "ShannonEntropy": lambda: -(T.where(fx != 0, fx * fx * T.log2((fx * fx).clamp_min(1e-300)), T.zeros_like(fx))).sum(d),
        "Sum": lambda: x.sum(d), "Max": lambda: x.amax(d), "Min": lambda: x.amin(d), "Prod": lambda: _prod(x, d),
        "ASum": lambda: x.abs().sum(d), "AMax": lambda: x.abs().amax(d), "AMin": lambda: x.abs().amin(d),
        "Any": lambda: (x != 0).sum(d) > 0, "All": lambda: (x != 0).sum(d) == n, "IsFinite": lambda: T.isfinite(fx).sum(d) == n,
        "IsInfOrNan": lambda: (~T.isfinite(fx)).sum(d) > 0, "IsNan": lambda: T.isnan(fx).sum(d) > 0,
        "IsInf": lambda: T.isinf(fx).sum(d) > 0, "IsPositive": lambda: (x > 0).sum(d) == n, "IsNegative": lambda: (x < 0).sum(d) == n,
        "CountNonZero": lambda: (x != 0).sum(d), "CountZero": lambda: (x == 0).sum(d), "MatchCondition": lambda: _match(x, e).sum(d),
        "ManhattanDistance": lambda: (fx - y).abs().sum(d), "EuclideanDistance": lambda: T.sqrt(((fx - y) ** 2).sum(d)),
        "CosineSimilarity": lambda: (fx * y).sum(d) / (T.sqrt((fx * fx).sum(d)) * T.sqrt((y * y).sum(d))),
        "CosineDistance": lambda: 1 - (fx * y).sum(d) / (T.sqrt((fx * fx).sum(d)) * T.sqrt((y * y).sum(d))),
        "Dot": lambda: (fx * y).sum(d), "EqualsWithEps": lambda: ((fx - y).abs() > (ev[0] if ev[0] > 0 else 1e-5)).to(fx.dtype).sum(d),
        "JaccardDistance": lambda: 1 - T.minimum(fx, y).sum(d) / T.maximum(fx, y).sum(d),
        "SimpleHammingDistance": lambda: (fx != y).to(fx.dtype).sum(d) / n,
        "Variance": lambda: fx.var(d, unbiased=bool(ev[0])) if n > 1 or not ev[0] else T.zeros_like(fx.sum(d)),
        "StandardDeviation": lambda: fx.std(d, unbiased=bool(ev[0])) if n > 1 or not ev[0] else T.zeros_like(fx.sum(d)),
        "IndexMax": lambda: idx(T.argmax, x), "IndexMin": lambda: idx(T.argmin, x),
        "IndexAbsoluteMax": lambda: idx(T.argmax, x.abs()), "IndexAbsoluteMin": lambda: idx(T.argmin, x.abs()),
        "FirstIndex": lambda: first_last(False), "LastIndex": lambda: first_last(True),
    }
    return tbl[name]()


def _prod(x, d):
    for k in sorted(d, reverse=True):
        x = x.prod(k)
    return x


# GPU operands that do NOT reach the HIP legacy kernels are counted here by reason (OpExecutioner.helperStatistics()['legacy'])
# and, with DL4J_AMD_LOG_FALLBACKS=1, logged once per reason — there is no unseen fallback.
import collections as _collections
import logging as _logging
import os as _os

FALLBACKS = _collections.Counter()
_LOG_FB = _os.environ.get("DL4J_AMD_LOG_FALLBACKS", "0") == "1"


def _fallback(reason):
    if _LOG_FB and reason not in FALLBACKS:
        _logging.getLogger("deeplearning4j_amd.ops").warning("legacy op on GPU operands runs on ATen: %s", reason)
    FALLBACKS[reason] += 1


# ----------------------------------------------------------------------------- executioner
class LegacyOpExecutioner:
    """exec* entry points by (family, opNum). Mirrors NativeOpExecutioner::exec* (legacy/cuda/NativeOpExecutioner.cu)."""

    # -- helpers
    @staticmethod
    def _gpu_ok(*ts):
        on_gpu = any(t is not None and t.is_cuda for t in ts)
        if _FORCE_TORCH[0] or any(t is not None and t.requires_grad for t in ts):
            if on_gpu:      # autograd sessions (SameDiff training) need the differentiable ATen formulation
                _fallback("forced ATen (gradient recording)" if _FORCE_TORCH[0] else "autograd-tracked operand")
            return False
        ok = L.use_hip(*ts) and all(t is None or L.dtype_code_all(t) >= 0 for t in ts)
        if on_gpu and not ok:
            _fallback("dtype without a HIP instantiation: %s" % sorted({str(t.dtype) for t in ts if t is not None and L.dtype_code_all(t) < 0})
                      if L.use_hip(*ts) else "operands on different devices / eager forced")
        return ok

    def _pairwise(self, name, x, y, z, extra, scalar=None, zbool=False):
        if scalar is None:
            shape = torch.broadcast_shapes(x.shape, y.shape)
        else:
            shape = x.shape
        gpu = self._gpu_ok(x, y if scalar is None else None) and (scalar is not None or y.dtype == x.dtype)
        if z is None:
            z = torch.empty(shape, device=x.device, dtype=torch.bool if zbool else x.dtype)
        if gpu:
            s = _strides3(x, y if scalar is None else None, z)
            if s is not None and z.numel() > 0:
                zz = z.view(torch.uint8) if zbool else z
                L.check(L.hip().dl4j_legacy_pairwise(
                    L.c_int(BIN[name]), L.ptr(x), L.ptr(y if scalar is None else None), L.ptr(zz), L.c_int64(z.numel()),
                    ctypes.byref(s), _extra(extra), L.c_int(0 if scalar is None else 1),
                    ctypes.c_double(0.0 if scalar is None else float(scalar)), L.c_int(L.dtype_code_all(x)),
                    L.c_int(int(zbool)), L.stream()), "legacy_pairwise(%s)" % name)
                return z
        yy = y if scalar is None else torch.as_tensor(scalar, device=x.device, dtype=x.dtype)
        r = _bin_cpu(name, x, yy, extra)
        z.copy_(r.to(z.dtype) if r.shape == z.shape else r.expand(z.shape).to(z.dtype))
        return z

    def _transform(self, name, x, z, extra, zbool=False):
        if z is None:
            z = torch.empty(x.shape, device=x.device, dtype=torch.bool if zbool else x.dtype)
        if self._gpu_ok(x) and x.numel() > 0 and (zbool or z.dtype == x.dtype):
            s = _strides3(x, None, z)
            if s is not None:
                zz = z.view(torch.uint8) if zbool else z
                L.check(L.hip().dl4j_legacy_transform(L.c_int(UN[name]), L.ptr(x), L.ptr(zz), L.c_int64(z.numel()),
                                                      ctypes.byref(s), _extra(extra), L.c_int(L.dtype_code_all(x)),
                                                      L.c_int(int(zbool)), L.stream()), "legacy_transform(%s)" % name)
                return z
        z.copy_(_un_cpu(name, x, extra).to(z.dtype))
        return z

    @staticmethod
    def _tad(x, dims, y=None):
        inner = dims
        outer = [k for k in range(x.dim()) if k not in inner]
        ish, (ist, iyst) = _coalesce([x.shape[k] for k in inner], [x.stride(k) for k in inner],
                                     [y.stride(k) for k in inner] if y is not None else [0] * len(inner))
        if outer:
            osh, (ost, oyst) = _coalesce([x.shape[k] for k in outer], [x.stride(k) for k in outer],
                                         [y.stride(k) for k in outer] if y is not None else [0] * len(outer))
        else:
            osh, ost, oyst = [1], [0], [0]
        if len(ish) > MAXR or len(osh) > MAXR:
            return None
        d = _TadDesc()
        d.irank, d.orank = len(ish), len(osh)
        for i in range(len(ish)):
            d.ishape[i], d.istride[i], d.iystride[i] = ish[i], ist[i], iyst[i]
        for i in range(len(osh)):
            d.oshape[i], d.ostride[i], d.oystride[i] = osh[i], ost[i], oyst[i]
        d.tad_len = math.prod(ish)
        d.num_tads = math.prod(osh)
        return d

    def _reduce(self, name, x, dims, keepDims, extra, zdtype, y=None):
        dims = _norm_dims(x, dims)
        out_shape = [x.shape[k] for k in range(x.dim()) if k not in dims]
        zk = {torch.bool: 1, torch.int64: 2, torch.float32: 3, torch.float64: 4}.get(zdtype, 0)
        if zdtype == x.dtype and zk in (3, 4):
            zk = 0
        tri = RED[name] >= RED["ManhattanDistance"]
        if self._gpu_ok(x, y) and x.numel() > 0 and x.dim() > 0 and (y is None or y.dtype == x.dtype) and not (tri and zk == 1):
            d = self._tad(x, dims, y)
            if d is not None:
                z = torch.empty(out_shape, device=x.device, dtype=zdtype)
                zz = z.view(torch.uint8) if zdtype == torch.bool else z
                if tri:
                    segs = int(L.hip().dl4j_legacy_reduce_segs(L.c_int64(d.num_tads), L.c_int64(d.tad_len)))
                    part = torch.empty(3 * d.num_tads * segs, device=x.device, dtype=torch.float64) if segs > 1 else None
                    L.check(L.hip().dl4j_legacy_reduce_tri_p(L.c_int(RED[name]), L.ptr(x), L.ptr(y), L.ptr(zz), L.ptr(part), ctypes.byref(d),
                                                             _extra(extra), L.c_int(L.dtype_code_all(x)), L.c_int(zk), L.stream()),
                            "legacy_reduce_tri(%s)" % name)
                else:
                    L.hip().dl4j_legacy_reduce_partial_len.restype = ctypes.c_int64
                    plen = int(L.hip().dl4j_legacy_reduce_partial_len(ctypes.byref(d), L.c_int(x.element_size())))
                    part = torch.empty(plen, device=x.device, dtype=torch.float64) if plen > 0 else None
                    L.check(L.hip().dl4j_legacy_reduce(L.c_int(RED[name]), L.ptr(x), L.ptr(zz), L.ptr(part), ctypes.byref(d),
                                                       _extra(extra), L.c_int(L.dtype_code_all(x)), L.c_int(zk), L.stream()),
                            "legacy_reduce(%s)" % name)
                return self._keep(z, x, dims, keepDims)
        r = _red_cpu(name, x, dims, extra, y).to(zdtype)
        return self._keep(r.reshape(out_shape), x, dims, keepDims)

    @staticmethod
    def _keep(z, x, dims, keepDims):
        if keepDims:
            shape = [1 if k in dims else x.shape[k] for k in range(x.dim())]
            return z.reshape(shape)
        return z

    # -- public exec* API
    def execPairwiseTransform(self, opNum, x, y, z=None, extraParams=None):
        return self._pairwise(PAIRWISE[opNum], x, y, z, extraParams)

    def execPairwiseTransformBool(self, opNum, x, y, z=None, extraParams=None):
        return self._pairwise(PAIRWISE_BOOL[opNum], x, y, z, extraParams, zbool=True)

    def execPairwiseTransformInt(self, opNum, x, y, z=None):
        return self._pairwise(PAIRWISE_INT[opNum], x, y, z, None)

    def execBroadcast(self, opNum, x, y, z=None, dimensions=None):
        return self._pairwise(BROADCAST[opNum], x, self._bcast_view(x, y, dimensions), z, None)

    def execBroadcastBool(self, opNum, x, y, z=None, dimensions=None, extraParams=None):
        return self._pairwise(BROADCAST_BOOL[opNum], x, self._bcast_view(x, y, dimensions), z, extraParams, zbool=True)

    def execBroadcastInt(self, opNum, x, y, z=None, dimensions=None):
        return self._pairwise(BROADCAST_INT[opNum], x, self._bcast_view(x, y, dimensions), z, None)

    @staticmethod
    def _bcast_view(x, y, dimensions):
        """DL4J broadcast ops name the x-dimensions that y spans (BaseBroadcastOp.dimension)."""
        if dimensions is None or y.dim() == x.dim():
            return y
        dims = [d % x.dim() for d in ([dimensions] if isinstance(dimensions, int) else dimensions)]
        shape = [1] * x.dim()
        ysq = y.reshape([s for s in y.shape if s != 1] or [1]) if y.dim() != len(dims) else y
        for i, d in enumerate(sorted(dims)):
            shape[d] = ysq.shape[i]
        return ysq.reshape(shape)

    def execScalar(self, opNum, x, scalar, z=None, extraParams=None):
        return self._pairwise(SCALAR[opNum], x, None, z, extraParams, scalar=float(scalar))

    def execScalarBool(self, opNum, x, scalar, z=None, extraParams=None):
        return self._pairwise(SCALAR_BOOL[opNum], x, None, z, extraParams, scalar=float(scalar), zbool=True)

    def execScalarInt(self, opNum, x, scalar, z=None):
        return self._pairwise(SCALAR_INT[opNum], x, None, z, None, scalar=int(scalar))

    def execTransformSame(self, opNum, x, z=None, extraParams=None):
        return self._transform(TRANSFORM_SAME[opNum], x, z, extraParams)

    def execTransformStrict(self, opNum, x, z=None, extraParams=None):
        return self._transform(TRANSFORM_STRICT[opNum], x, z, extraParams)

    def execTransformFloat(self, opNum, x, z=None, extraParams=None):
        if not x.is_floating_point():
            x = x.float()
        return self._transform(TRANSFORM_FLOAT[opNum], x, z, extraParams)

    def execTransformBool(self, opNum, x, z=None, extraParams=None):
        return self._transform(TRANSFORM_BOOL[opNum], x, z, extraParams, zbool=True)

    def execTransformAny(self, opNum, x, z=None, extraParams=None):
        if z is not None and z.dtype != x.dtype:
            z.copy_(x)
            return z
        return self._transform(TRANSFORM_ANY[opNum], x, z, extraParams)

    def execReduceFloat(self, opNum, x, dimensions=None, keepDims=False, extraParams=None):
        zd = x.dtype if x.dtype in _FLOAT_T else torch.float32
        return self._reduce(REDUCE_FLOAT[opNum], x, dimensions, keepDims, extraParams, zd)

    def execReduceSame(self, opNum, x, dimensions=None, keepDims=False, extraParams=None):
        return self._reduce(REDUCE_SAME[opNum], x, dimensions, keepDims, extraParams, x.dtype)

    def execReduceBool(self, opNum, x, dimensions=None, keepDims=False, extraParams=None):
        return self._reduce(REDUCE_BOOL[opNum], x, dimensions, keepDims, extraParams, torch.bool)

    def execReduceLong(self, opNum, x, dimensions=None, keepDims=False, extraParams=None):
        return self._reduce(REDUCE_LONG[opNum], x, dimensions, keepDims, extraParams, torch.int64)

    def execReduce3(self, opNum, x, y, dimensions=None, keepDims=False, extraParams=None):
        zd = x.dtype if x.dtype in _FLOAT_T else torch.float32
        return self._reduce(REDUCE3[opNum], x, dimensions, keepDims, extraParams, zd, y=y)

    def execReduce3All(self, opNum, x, y, dimensions, extraParams=None):
        """All-pairs of TADs: result [numTadsX, numTadsY] (NativeOps.h execReduce3All)."""
        dims = _norm_dims(x, dimensions)
        keep = [k for k in range(x.dim()) if k not in dims]
        xm = x.permute(*keep, *dims).reshape(math.prod([x.shape[k] for k in keep]) if keep else 1, -1)
        keep_y = [k for k in range(y.dim()) if k not in dims]
        ym = y.permute(*keep_y, *dims).reshape(math.prod([y.shape[k] for k in keep_y]) if keep_y else 1, -1)
        xe = xm.unsqueeze(1).expand(xm.shape[0], ym.shape[0], xm.shape[1])
        ye = ym.unsqueeze(0).expand(xm.shape[0], ym.shape[0], ym.shape[1])
        return self.execReduce3(opNum, xe, ye, [2], False, extraParams)

    def execIndexReduce(self, opNum, x, dimensions=None, keepDims=False, extraParams=None):
        return self._reduce(INDEX_REDUCE[opNum], x, dimensions, keepDims, extraParams, torch.int64)

    def execSummaryStats(self, opNum, x, dimensions=None, biasCorrected=True, keepDims=False):
        zd = x.dtype if x.dtype in _FLOAT_T else torch.float32
        return self._reduce(SUMMARY_STATS[opNum], x, dimensions, keepDims, [1.0 if biasCorrected else 0.0], zd)

    def execRandom(self, opNum, rng, z, x=None, y=None, extraArguments=None):
        from ..rng import exec_random
        return exec_random(RANDOM[opNum], rng, z, x, y, extraArguments)


def _profiled(fn):
    """OpProfiler hook of the legacy executioner (OpProfiler.java processOpCall is invoked from DefaultOpExecutioner.exec for
    EVERY op class, not only custom ops): invocation counts, optional timing / strided-operand penalties, NaN / Inf panics."""
    import functools

    @functools.wraps(fn)
    def run(self, name, x, *a, **kw):
        from ..environment import OpProfiler
        prof = OpProfiler._inst
        if prof is None:
            return fn(self, name, x, *a, **kw)
        with prof.op(name, x, *[t for t in a[:1] if isinstance(t, torch.Tensor)]):
            res = fn(self, name, x, *a, **kw)
        cfg = prof.config
        if cfg.checkForNAN or cfg.checkForINF:
            prof.checkOutputs(name, res)
        return res
    return run


for _m in ("_pairwise", "_transform", "_reduce"):
    setattr(LegacyOpExecutioner, _m, _profiled(getattr(LegacyOpExecutioner, _m)))

_EXEC = LegacyOpExecutioner()


def executioner() -> LegacyOpExecutioner:
    return _EXEC


def exec_legacy(family: str, opName: str, inputs, tArgs=None, dimensions=None, scalar=None, keepDims=False):
    """Run one legacy op by (family, name) on tensors — the eager counterpart of SameDiff's legacy nodes, used by the generated
    ND* namespaces (codegen/generator.py)."""
    lg = executioner()
    n = op_num(family, opName)
    x = inputs[0] if inputs else None
    ext = list(tArgs) if tArgs else None
    dims = list(dimensions) if dimensions else None
    if family == "transform_same": return lg.execTransformSame(n, x, None, ext)
    if family == "transform_strict": return lg.execTransformStrict(n, x, None, ext)
    if family == "transform_float": return lg.execTransformFloat(n, x, None, ext)
    if family == "transform_bool": return lg.execTransformBool(n, x, None, ext)
    if family == "transform_any": return lg.execTransformAny(n, x, None, ext)
    if family == "pairwise": return lg.execPairwiseTransform(n, x, inputs[1], None, ext)
    if family == "pairwise_bool": return lg.execPairwiseTransformBool(n, x, inputs[1], None, ext)
    if family == "scalar": return lg.execScalar(n, x, float(scalar if scalar is not None else (ext[0] if ext else 0.0)))
    if family == "scalar_bool": return lg.execScalarBool(n, x, float(scalar if scalar is not None else (ext[0] if ext else 0.0)))
    if family == "reduce_float": return lg.execReduceFloat(n, x, dims, keepDims, ext)
    if family == "reduce_same": return lg.execReduceSame(n, x, dims, keepDims, ext)
    if family == "reduce_bool": return lg.execReduceBool(n, x, dims, keepDims, ext)
    if family == "reduce_long": return lg.execReduceLong(n, x, dims, keepDims, ext)
    if family == "reduce3": return lg.execReduce3(n, x, inputs[1], dims, keepDims, ext)
    if family == "index_reduce": return lg.execIndexReduce(n, x, dims, keepDims, ext)
    if family == "summary_stats": return lg.execSummaryStats(n, x, dims, True, keepDims)
    raise KeyError("unknown legacy family %r" % family)
