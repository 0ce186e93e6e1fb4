"""Random number generation: ``Random`` (NativeRandom analogue) + legacy random-op execution.

Reference: nd4j-api linalg/api/rng/{Random,DefaultRandom}.java, nd4j-native CpuNativeRandom /
nd4j-cuda CudaNativeRandom (root/node state mirrored to native), distributions under
linalg/api/rng/distribution/impl/*, legacy random ops (libnd4j loops/cuda/random.cu; opNums in
nd4j/ops/legacy.py RANDOM).

Own design: a counter-based Philox4x32-10 stream keyed by (rootSeed, nodeSeed) (native/csrc/philox.h).
Element i of a fill uses counter ``offset + i`` — host (rt_random.cpp) and device (random.hip)
produce identical bits for uniform / gaussian / bernoulli, so CPU tests pin the GPU stream.
"""
from __future__ import annotations

import ctypes
import threading
import time

import torch

from ..native import lib as L

_OPS = {"UniformDistribution": 0, "DropOut": 1, "DropOutInverted": 2, "ProbablisticMerge": 3, "Linspace": 4, "Choice": 5,
        "GaussianDistribution": 6, "BernoulliDistribution": 7, "BinomialDistribution": 8, "BinomialDistributionEx": 9,
        "LogNormalDistribution": 10, "TruncatedNormalDistribution": 11, "AlphaDropOut": 12, "ExponentialDistribution": 13,
        "ExponentialDistributionInv": 14, "PoissonDistribution": 15, "GammaDistribution": 16}


class Random:
    def __init__(self, seed=None, nodeSeed=0):
        if seed is None:
            seed = time.time_ns() & 0x7FFFFFFFFFFFFFFF
        self._rt = L.rt()
        self._h = self._rt.rng_create(ctypes.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), ctypes.c_uint64(nodeSeed))
        self._seed = int(seed)
        self._lock = threading.Lock()

    def __del__(self):
        try:
            self._rt.rng_destroy(self._h)
        except Exception:
            pass

    # -- state
    def setSeed(self, seed):
        self._seed = int(seed)
        self._rt.rng_set_states(self._h, ctypes.c_uint64(self._seed & 0xFFFFFFFFFFFFFFFF), ctypes.c_uint64(0))

    def getSeed(self):
        return self._seed

    def rootState(self):
        return int(self._rt.rng_root_state(self._h))

    def nodeState(self):
        return int(self._rt.rng_node_state(self._h))

    def setStates(self, root, node):
        self._rt.rng_set_states(self._h, ctypes.c_uint64(root), ctypes.c_uint64(node))

    def position(self):
        return int(self._rt.rng_offset(self._h))

    def reserve(self, n):
        """Claim ``n`` draws: returns (key, stream, offset) for a fill and advances the stream."""
        with self._lock:
            off = self.position()
            self._rt.rng_rewind(self._h, ctypes.c_uint64(int(n)))
        return self.rootState(), self.nodeState(), off

    # -- scalar draws
    def nextInt(self, bound=None):
        _, _, off = self.reserve(1)
        self._rt.rng_rewind(self._h, ctypes.c_uint64(0))
        v = int(philox_bits(self.rootState(), self.nodeState(), off))
        return v if bound is None else v % int(bound)

    def nextLong(self):
        return (self.nextInt() << 32) | self.nextInt()

    def nextFloat(self):
        return ((self.nextInt() >> 8) + 0.5) / 16777216.0

    def nextDouble(self):
        return ((self.nextLong() >> 11) + 0.5) / 9007199254740992.0

    def nextBoolean(self):
        return bool(self.nextInt() & 1)

    def nextGaussian(self):
        import math
        u1, u2 = self.nextFloat(), self.nextFloat()
        return math.sqrt(-2.0 * math.log(u1)) * math.cos(2 * math.pi * u2)

    # -- array fills (Nd4j.rand / randn)
    def nextFloatArray(self, shape, device=None, dtype=torch.float32):
        z = torch.empty(shape, device=device, dtype=dtype)
        return exec_random("UniformDistribution", self, z, None, None, [0.0, 1.0])

    def nextGaussianArray(self, shape, device=None, dtype=torch.float32):
        z = torch.empty(shape, device=device, dtype=dtype)
        return exec_random("GaussianDistribution", self, z, None, None, [0.0, 1.0])


def philox_bits(key, stream, index):
    """Pure-python Philox4x32-10 (reference for tests and scalar draws)."""
    M0, M1, W0, W1, MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    ctr = index >> 2
    c = [ctr & MASK, (ctr >> 32) & MASK, stream & MASK, (stream >> 32) & MASK]
    k0, k1 = key & MASK, (key >> 32) & MASK
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c[index & 3]


_DEFAULT = None


def getRandom() -> Random:
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = Random(119)
    return _DEFAULT


def setDefaultSeed(seed):
    getRandom().setSeed(seed)


def _draws_per_element(op):
    return {"GaussianDistribution": 2, "LogNormalDistribution": 2, "TruncatedNormalDistribution": 2}.get(op, 1)


def exec_random(op: str, rng: Random, z: torch.Tensor, x=None, y=None, args=None):
    """Fill ``z`` in place. args by op: Uniform [from,to]; Gaussian/LogNormal/TruncatedNormal [mean,std];
    Bernoulli/DropOut/DropOutInverted/ProbablisticMerge [p]; Binomial [trials,p]; Exponential [lambda];
    Poisson [lambda]; Gamma [alpha,beta]; AlphaDropOut [p, alpha', a, b]; Linspace [from,to]."""
    rng = rng or getRandom()
    a = [float(v) for v in (args or [])]
    n = z.numel()
    key, stream, off = rng.reserve(n * _draws_per_element(op) + 2)
    off += off & 1          # keep Box-Muller pairs aligned
    trials = 0
    if op in ("BinomialDistribution", "BinomialDistributionEx"):
        trials, a = int(a[0]), [a[1]] + a[2:]
    a = (a + [0.0] * 4)[:4]
    if z.is_cuda:
        code = L.dtype_code_all(z)
        if code not in (0, 1, 2):
            raise TypeError("random ops fill floating-point arrays")
        if op == "Choice":
            src = x.float().contiguous()
            cdf = torch.cumsum(y.float().reshape(-1), 0).contiguous()
            tmp = z if z.dtype == torch.float32 and z.is_contiguous() else torch.empty(z.shape, device=z.device)
            L.check(L.hip().dl4j_random_choice(L.ptr(src), L.ptr(cdf), L.c_int(cdf.numel()), L.ptr(tmp), L.c_int64(n),
                                               ctypes.c_uint64(key), ctypes.c_uint64(stream), ctypes.c_uint64(off), L.stream()),
                    "random_choice")
            if tmp is not z:
                z.copy_(tmp)
            return z
        zz = z if z.is_contiguous() else torch.empty_like(z, memory_format=torch.contiguous_format)
        xx = None if x is None else x.to(z.dtype).contiguous()
        yy = None if y is None else y.to(z.dtype).contiguous()
        L.check(L.hip().dl4j_random_fill(L.c_int(_OPS[op]), L.ptr(xx), L.ptr(yy), L.ptr(zz), L.c_int64(n),
                                         ctypes.c_uint64(key), ctypes.c_uint64(stream), ctypes.c_uint64(off),
                                         (ctypes.c_float * 4)(*a), L.c_int(trials), L.c_int(code), L.stream()),
                "random_fill(%s)" % op)
        if zz is not z:
            z.copy_(zz)
        return z
    return _exec_random_cpu(op, key, stream, off, z, x, y, a, trials)


def _exec_random_cpu(op, key, stream, off, z, x, y, a, trials):
    rt = L.rt()
    n = z.numel()
    buf = torch.empty(n, dtype=torch.float32)
    u64 = ctypes.c_uint64

    def uniform01():
        rt.rng_fill_uniform_f32(buf.data_ptr(), n, u64(key), u64(stream), u64(off), 0.0, 1.0)
        return buf.reshape(z.shape)

    def gauss():
        rt.rng_fill_gaussian_f32(buf.data_ptr(), n, u64(key), u64(stream), u64(off), 0.0, 1.0)
        return buf.reshape(z.shape)

    g = torch.Generator().manual_seed((key ^ (stream * 0x9E3779B97F4A7C15) ^ off) & 0x7FFFFFFFFFFFFFFF)
    if op == "UniformDistribution":
        r = a[0] + (a[1] - a[0]) * uniform01()
    elif op == "GaussianDistribution":
        r = a[0] + a[1] * gauss()
    elif op == "LogNormalDistribution":
        r = torch.exp(a[0] + a[1] * gauss())
    elif op == "TruncatedNormalDistribution":
        gz = gauss().clone()
        for _ in range(16):
            bad = gz.abs() > 2
            if not bad.any():
                break
            gz[bad] = torch.randn(int(bad.sum()), generator=g)
        r = a[0] + a[1] * gz.clamp(-2, 2)
    elif op == "BernoulliDistribution":
        r = (uniform01() < a[0]).float()
    elif op == "DropOut":
        r = torch.where(uniform01() < a[0], x.float(), torch.zeros(()))
    elif op == "DropOutInverted":
        r = torch.where(uniform01() < a[0], x.float() / a[0], torch.zeros(()))
    elif op == "ProbablisticMerge":
        r = torch.where(uniform01() < a[0], x.float(), y.float())
    elif op == "AlphaDropOut":
        r = a[2] * torch.where(uniform01() < a[0], x.float(), torch.full((), a[1])) + a[3]
    elif op == "Linspace":
        r = torch.linspace(a[0], a[1], n).reshape(z.shape)
    elif op in ("BinomialDistribution", "BinomialDistributionEx"):
        r = torch.binomial(torch.full(z.shape, float(trials)), torch.full(z.shape, a[0]), generator=g)
    elif op == "ExponentialDistribution":
        r = -torch.log(uniform01()) / a[0]
    elif op == "ExponentialDistributionInv":
        r = 1 - torch.exp(-a[0] * x.float())
    elif op == "PoissonDistribution":
        lam = x.float() if x is not None else torch.full(z.shape, a[0])
        r = torch.poisson(lam, generator=g)
    elif op == "GammaDistribution":
        alpha = x.double() if x is not None else torch.full(z.shape, a[0], dtype=torch.float64)
        r = torch._standard_gamma(alpha, generator=g).float() / (a[1] if a[1] > 0 else 1.0)
    elif op == "Choice":
        cdf = torch.cumsum(y.float().reshape(-1), 0)
        idx = torch.searchsorted(cdf, uniform01().reshape(-1) * cdf[-1]).clamp_max(cdf.numel() - 1)
        r = x.reshape(-1)[idx].reshape(z.shape)
    else:
        raise KeyError(op)
    z.copy_(r.reshape(z.shape).to(z.dtype))
    return z


# ---------------------------------------------------------------------------------- distributions (linalg/api/rng/distribution)
class Distribution:
    def __init__(self, rng=None):
        self.rng = rng

    def sample(self, shape, device=None, dtype=torch.float32):
        raise NotImplementedError


class UniformDistribution(Distribution):
    def __init__(self, lower=0.0, upper=1.0, rng=None):
        super().__init__(rng)
        self.lower, self.upper = lower, upper

    def sample(self, shape, device=None, dtype=torch.float32):
        return exec_random("UniformDistribution", self.rng, torch.empty(shape, device=device, dtype=dtype), None, None,
                           [self.lower, self.upper])


class NormalDistribution(Distribution):
    def __init__(self, mean=0.0, std=1.0, rng=None):
        super().__init__(rng)
        self.mean, self.std = mean, std

    def sample(self, shape, device=None, dtype=torch.float32):
        return exec_random("GaussianDistribution", self.rng, torch.empty(shape, device=device, dtype=dtype), None, None,
                           [self.mean, self.std])


class TruncatedNormalDistribution(NormalDistribution):
    def sample(self, shape, device=None, dtype=torch.float32):
        return exec_random("TruncatedNormalDistribution", self.rng, torch.empty(shape, device=device, dtype=dtype), None, None,
                           [self.mean, self.std])


class LogNormalDistribution(NormalDistribution):
    def sample(self, shape, device=None, dtype=torch.float32):
        return exec_random("LogNormalDistribution", self.rng, torch.empty(shape, device=device, dtype=dtype), None, None,
                           [self.mean, self.std])


class BinomialDistribution(Distribution):
    def __init__(self, trials, p, rng=None):
        super().__init__(rng)
        self.trials, self.p = trials, p

    def sample(self, shape, device=None, dtype=torch.float32):
        return exec_random("BinomialDistribution", self.rng, torch.empty(shape, device=device, dtype=dtype), None, None,
                           [self.trials, self.p])


class BernoulliDistribution(Distribution):
    def __init__(self, p, rng=None):
        super().__init__(rng)
        self.p = p

    def sample(self, shape, device=None, dtype=torch.float32):
        return exec_random("BernoulliDistribution", self.rng, torch.empty(shape, device=device, dtype=dtype), None, None, [self.p])
