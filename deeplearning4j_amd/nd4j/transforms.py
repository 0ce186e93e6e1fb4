"""``Transforms``: the static elementwise / distance helpers user code calls on INDArrays (reference: nd4j-api
org/nd4j/linalg/ops/transforms/Transforms.java). Every function takes INDArrays or tensors; ``dup=True`` (the reference's
default) returns a new array, ``dup=False`` writes the result into the argument. On the GPU these are single ATen / HIP
elementwise launches on the array's device."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .ndarray import INDArray


def _t(x):
    return x.tensor() if isinstance(x, INDArray) else torch.as_tensor(x)


def _ret(x, out, dup):
    if dup or not isinstance(x, (INDArray, torch.Tensor)):
        return INDArray(out)
    _t(x).copy_(out)
    return x if isinstance(x, INDArray) else INDArray(x)


def _unary(fn):
    def f(x, dup: bool = True):
        return _ret(x, fn(_t(x)), dup)
    return staticmethod(f)


class Transforms:
    abs = _unary(torch.abs)
    acos = _unary(torch.acos)
    asin = _unary(torch.asin)
    atan = _unary(torch.atan)
    ceil = _unary(torch.ceil)
    ceiling = ceil
    cos = _unary(torch.cos)
    cosh = _unary(torch.cosh)
    exp = _unary(torch.exp)
    expm1 = _unary(torch.expm1)
    floor = _unary(torch.floor)
    log = _unary(torch.log)
    log1p = _unary(torch.log1p)
    neg = _unary(torch.neg)
    round = _unary(torch.round)
    sign = _unary(torch.sign)
    sin = _unary(torch.sin)
    sinh = _unary(torch.sinh)
    sqrt = _unary(torch.sqrt)
    tan = _unary(torch.tan)
    tanh = _unary(torch.tanh)
    atanh = _unary(torch.atanh)
    sigmoid = _unary(torch.sigmoid)
    sigmoidDerivative = _unary(lambda t: torch.sigmoid(t) * (1 - torch.sigmoid(t)))
    hardTanh = _unary(lambda t: t.clamp(-1.0, 1.0))
    hardTanhDerivative = _unary(lambda t: ((t >= -1) & (t <= 1)).to(t.dtype))
    relu = _unary(F.relu)
    relu6 = _unary(F.relu6)
    elu = _unary(F.elu)
    eluDerivative = _unary(lambda t: torch.where(t > 0, torch.ones_like(t), torch.exp(t)))
    softPlus = _unary(F.softplus)
    softsign = _unary(F.softsign)
    softsignDerivative = _unary(lambda t: 1.0 / (1.0 + t.abs()) ** 2)
    identity = _unary(lambda t: t.clone())
    stabilize = staticmethod(lambda x, k=1.0, dup=True: _ret(x, _t(x).clamp(-k * 88.0, k * 88.0), dup))
    isMax = _unary(lambda t: (t == t.max()).to(t.dtype))
    not_ = _unary(lambda t: (t == 0).to(t.dtype))
    reverse = _unary(lambda t: t.flip(tuple(range(t.dim()))))

    @staticmethod
    def log_base(x, base: float, dup: bool = True):
        t = _t(x)
        return _ret(x, torch.log(t) / torch.log(torch.tensor(float(base), dtype=t.dtype, device=t.device)), dup)

    @staticmethod
    def pow(x, power, dup: bool = True):
        return _ret(x, torch.pow(_t(x), _t(power) if isinstance(power, (INDArray, torch.Tensor)) else power), dup)

    @staticmethod
    def leakyRelu(x, cutoff: float = 0.01, dup: bool = True):
        return _ret(x, F.leaky_relu(_t(x), cutoff), dup)

    @staticmethod
    def leakyReluDerivative(x, cutoff: float = 0.01, dup: bool = True):
        t = _t(x)
        return _ret(x, torch.where(t >= 0, torch.ones_like(t), torch.full_like(t, cutoff)), dup)

    @staticmethod
    def softmax(x, dup: bool = True):
        return _ret(x, torch.softmax(_t(x), -1), dup)

    @staticmethod
    def max(x, k, dup: bool = True):
        return _ret(x, torch.maximum(_t(x), _t(k).to(_t(x).dtype)) if isinstance(k, (INDArray, torch.Tensor)) else _t(x).clamp(min=k), dup)

    @staticmethod
    def min(x, k, dup: bool = True):
        return _ret(x, torch.minimum(_t(x), _t(k).to(_t(x).dtype)) if isinstance(k, (INDArray, torch.Tensor)) else _t(x).clamp(max=k), dup)

    @staticmethod
    def clip(x, lo, hi, dup: bool = True):
        return _ret(x, _t(x).clamp(lo, hi), dup)

    @staticmethod
    def unitVec(x):
        t = _t(x)
        n = torch.linalg.vector_norm(t.float())
        return INDArray(t / n.to(t.dtype)) if float(n) > 0 else INDArray(t.clone())

    @staticmethod
    def normalizeZeroMeanAndUnitVariance(x):
        t = _t(x)
        mu, sd = t.mean(0, keepdim=True), t.std(0, keepdim=True, unbiased=True)
        t.sub_(mu).div_(torch.where(sd == 0, torch.ones_like(sd), sd))
        return x if isinstance(x, INDArray) else INDArray(t)

    # -- distances / similarities (scalars)
    @staticmethod
    def cosineSim(a, b) -> float:
        x, y = _t(a).double().reshape(-1), _t(b).double().reshape(-1)
        d = torch.linalg.vector_norm(x) * torch.linalg.vector_norm(y)
        return float(torch.dot(x, y) / d) if float(d) > 0 else 0.0

    @staticmethod
    def cosineDistance(a, b) -> float:
        return 1.0 - Transforms.cosineSim(a, b)

    @staticmethod
    def euclideanDistance(a, b) -> float:
        return float(torch.linalg.vector_norm(_t(a).double() - _t(b).double()))

    @staticmethod
    def manhattanDistance(a, b) -> float:
        return float((_t(a).double() - _t(b).double()).abs().sum())

    @staticmethod
    def hammingDistance(a, b) -> float:
        return float((_t(a) != _t(b)).sum())

    @staticmethod
    def jaccardDistance(a, b) -> float:
        x, y = _t(a).double(), _t(b).double()
        mx = torch.maximum(x, y).sum()
        return float(1.0 - torch.minimum(x, y).sum() / mx) if float(mx) != 0 else 0.0

    @staticmethod
    def allCosineSimilarities(a, b, *dims):
        x, y = _t(a).float(), _t(b).float()
        return INDArray(F.normalize(x, dim=-1) @ F.normalize(y, dim=-1).transpose(-1, -2))

    @staticmethod
    def allEuclideanDistances(a, b, *dims):
        return INDArray(torch.cdist(_t(a).float(), _t(b).float(), p=2))

    @staticmethod
    def allManhattanDistances(a, b, *dims):
        return INDArray(torch.cdist(_t(a).float(), _t(b).float(), p=1))

    # -- comparisons / logic (0/1 arrays of the input dtype, as in the reference)
    @staticmethod
    def greaterThanOrEqual(a, b, dup: bool = True):
        return _ret(a, (_t(a) >= _t(b)).to(_t(a).dtype), dup)

    @staticmethod
    def lessThanOrEqual(a, b, dup: bool = True):
        return _ret(a, (_t(a) <= _t(b)).to(_t(a).dtype), dup)

    @staticmethod
    def and_(a, b):
        return INDArray((_t(a) != 0) & (_t(b) != 0))

    @staticmethod
    def or_(a, b):
        return INDArray((_t(a) != 0) | (_t(b) != 0))

    @staticmethod
    def xor(a, b):
        return INDArray((_t(a) != 0) ^ (_t(b) != 0))

    @staticmethod
    def atan2(y, x):
        return INDArray(torch.atan2(_t(y), _t(x)))

    @staticmethod
    def dot(a, b) -> float:
        return float(torch.dot(_t(a).double().reshape(-1), _t(b).double().reshape(-1)))

    @staticmethod
    def cross(a, b):
        return INDArray(torch.linalg.cross(_t(a), _t(b)))

    @staticmethod
    def mpow(matrix, n: int, dup: bool = True):
        return _ret(matrix, torch.linalg.matrix_power(_t(matrix), int(n)), dup)
