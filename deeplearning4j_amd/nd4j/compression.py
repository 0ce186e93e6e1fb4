"""Gradient compression: threshold + bitmap codecs and the compressor SPI.

Reference: nd4j-api linalg/compression/{NDArrayCompressor,BasicNDArrayCompressor,ThresholdCompression,
CompressedDataBuffer}.java; backend compressors nd4j-cuda compression/CudaThreshold.java:85-110 and
nd4j-native CpuThreshold; native codecs libnd4j loops/cuda/type_conversions.cu:215-499.
Message layout (int32): [count, length, floatBits(tau), flag, payload...] — identical on host
(rt_codec.cpp) and device (codec.hip, wave64 ballots).

On xGMI dense RCCL all-reduce beats sparse exchange for DL4J-sized models, so these are kept for API
compatibility (ParallelWrapper thresholdAlgorithm / residualPostProcessor) and for slow links.
"""
from __future__ import annotations

import ctypes
import struct
from typing import Optional

import torch

from ..native import lib as L

THRESHOLD, BITMAP = "THRESHOLD", "BITMAP"


def _f32_flat(t):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise TypeError("codecs operate in place on contiguous float32 arrays (the flat gradient/residual vector)")
    return t.reshape(-1)


class ThresholdCompression:
    FLEXIBLE_ENCODING, THRESHOLD_ENCODING, BITMAP_ENCODING = 0, 1, 2

    @staticmethod
    def count(x: torch.Tensor, threshold: float) -> int:
        x = _f32_flat(x)
        if x.is_cuda:
            return int((x.abs() >= threshold).sum())
        return int(L.rt().nd4j_threshold_count_f32(x.data_ptr(), x.numel(), threshold))

    @staticmethod
    def thresholdEncode(x: torch.Tensor, threshold: float, boundary: Optional[int] = None) -> Optional[torch.Tensor]:
        """Encode |x| >= threshold in place (x keeps the residual). Returns the int32 message or None if nothing passes."""
        x = _f32_flat(x)
        n = x.numel()
        cap = int(boundary) if boundary is not None else n
        cap = max(1, min(cap, n))
        if x.is_cuda:
            hip = L.hip()
            cnt = int((x.abs() >= threshold).sum())      # sizing pass (CudaThreshold.java: MatchCondition reduce)
            if cnt == 0:
                return None
            cap = min(cap, cnt)
            out = torch.empty(4 + cap, dtype=torch.int32, device=x.device)
            ws = torch.empty(int(hip.dl4j_threshold_ws_ints(L.c_int64(n))), dtype=torch.int32, device=x.device)
            L.check(hip.dl4j_threshold_encode(L.ptr(x), L.c_int64(n), L.c_float(threshold), L.ptr(out), L.c_int(cap), L.ptr(ws),
                                              L.stream()), "threshold_encode")
            return out
        rt = L.rt()
        cnt = int(rt.nd4j_threshold_count_f32(x.data_ptr(), n, threshold))
        if cnt == 0:
            return None
        cap = min(cap, cnt)
        out = torch.empty(4 + cap, dtype=torch.int32)
        got = rt.nd4j_threshold_encode_f32(x.data_ptr(), n, threshold, out.data_ptr(), cap)
        if got < 0:
            L.rt_check(20, "threshold_encode")
        return out

    @staticmethod
    def thresholdDecode(encoded: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        z = _f32_flat(target)
        if encoded.is_cuda != z.is_cuda:
            encoded = encoded.to(z.device)
        if z.is_cuda:
            if int(encoded[1]) > z.numel():
                raise ValueError("threshold decode: target shorter than encoded length")
            L.check(L.hip().dl4j_threshold_decode(L.ptr(encoded), L.c_int(encoded.numel() - 4), L.ptr(z), L.stream()),
                    "threshold_decode")
        else:
            L.rt_check(L.rt().nd4j_threshold_decode_f32(encoded.data_ptr(), z.data_ptr(), z.numel()), "threshold_decode")
        return target

    @staticmethod
    def bitmapEncode(x: torch.Tensor, threshold: float) -> torch.Tensor:
        x = _f32_flat(x)
        n = x.numel()
        out = torch.empty(4 + (n + 15) // 16, dtype=torch.int32, device=x.device)
        if x.is_cuda:
            ws = torch.empty(2048, dtype=torch.int32, device=x.device)
            L.check(L.hip().dl4j_bitmap_encode(L.ptr(x), L.c_int64(n), L.c_float(threshold), L.ptr(out), L.ptr(ws), L.stream()),
                    "bitmap_encode")
        else:
            L.rt().nd4j_bitmap_encode_f32(x.data_ptr(), n, threshold, out.data_ptr())
        return out

    @staticmethod
    def bitmapDecode(encoded: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        z = _f32_flat(target)
        if encoded.is_cuda != z.is_cuda:
            encoded = encoded.to(z.device)
        if z.is_cuda:
            n = int(encoded[1])
            if n > z.numel():
                raise ValueError("bitmap decode: target shorter than encoded length")
            L.check(L.hip().dl4j_bitmap_decode(L.ptr(encoded), L.c_int64(n), L.ptr(z), L.stream()), "bitmap_decode")
        else:
            L.rt_check(L.rt().nd4j_bitmap_decode_f32(encoded.data_ptr(), z.data_ptr(), z.numel()), "bitmap_decode")
        return target

    @staticmethod
    def estimateThreshold(x: torch.Tensor, targetCount: int, seed=0) -> float:
        x = _f32_flat(x)
        if x.is_cuda:
            k = max(1, min(int(targetCount), x.numel()))
            sample = x if x.numel() <= (1 << 22) else x[torch.randint(0, x.numel(), (1 << 22,), device=x.device)]
            kk = max(1, int(round(k * sample.numel() / x.numel())))
            return float(torch.topk(sample.abs(), kk, sorted=True).values[-1])
        return float(L.rt().nd4j_estimate_threshold_f32(x.data_ptr(), x.numel(), int(targetCount), seed))

    @staticmethod
    def header(encoded: torch.Tensor):
        h = encoded[:4].cpu().tolist()
        return {"count": h[0], "length": h[1], "threshold": struct.unpack("<f", struct.pack("<i", h[2]))[0], "flag": h[3]}


class CompressedArray:
    """Analogue of an INDArray backed by CompressedDataBuffer: payload + descriptor."""

    def __init__(self, algo, payload, shape, dtype):
        self.algorithm, self.payload, self.shape, self.dtype = algo, payload, list(shape), dtype

    def isCompressed(self):
        return True


class NDArrayCompressor:
    def getDescriptor(self):
        raise NotImplementedError

    def compress(self, arr):
        raise NotImplementedError

    def decompress(self, c):
        raise NotImplementedError


class _CastCompressor(NDArrayCompressor):
    def __init__(self, name, dtype):
        self.name, self.dtype = name, dtype

    def getDescriptor(self):
        return self.name

    def compress(self, arr):
        t = arr.tensor() if hasattr(arr, "tensor") else arr
        if self.dtype is None:
            return CompressedArray(self.name, t.clone(), t.shape, t.dtype)
        if self.dtype == torch.int8:
            scale = t.abs().amax().clamp_min(1e-30) / 127.0
            return CompressedArray(self.name, torch.cat([scale.reshape(1).float().view(torch.int8).to(t.device),
                                                         torch.round(t / scale).to(torch.int8).reshape(-1)]), t.shape, t.dtype)
        return CompressedArray(self.name, t.to(self.dtype), t.shape, t.dtype)

    def decompress(self, c):
        from .ndarray import INDArray
        if self.dtype == torch.int8:
            scale = c.payload[:4].view(torch.float32)
            return INDArray((c.payload[4:].to(c.dtype) * scale.to(c.dtype)).reshape(c.shape))
        return INDArray(c.payload.to(c.dtype).reshape(c.shape))


class _ThresholdCompressor(NDArrayCompressor):
    def __init__(self, threshold=1e-3):
        self.threshold = threshold

    def getDescriptor(self):
        return THRESHOLD

    def configure(self, threshold):
        self.threshold = float(threshold)

    def compress(self, arr):
        t = arr.tensor() if hasattr(arr, "tensor") else arr
        enc = ThresholdCompression.thresholdEncode(t, self.threshold)
        return None if enc is None else CompressedArray(THRESHOLD, enc, t.shape, t.dtype)

    def decompress(self, c, target=None):
        from .ndarray import INDArray
        z = target if target is not None else torch.zeros(c.shape, dtype=torch.float32, device=c.payload.device)
        ThresholdCompression.thresholdDecode(c.payload, z.tensor() if hasattr(z, "tensor") else z)
        return z if hasattr(z, "tensor") else INDArray(z)


class _BitmapCompressor(_ThresholdCompressor):
    def getDescriptor(self):
        return BITMAP

    def compress(self, arr):
        t = arr.tensor() if hasattr(arr, "tensor") else arr
        return CompressedArray(BITMAP, ThresholdCompression.bitmapEncode(t, self.threshold), t.shape, t.dtype)

    def decompress(self, c, target=None):
        from .ndarray import INDArray
        z = target if target is not None else torch.zeros(c.shape, dtype=torch.float32, device=c.payload.device)
        ThresholdCompression.bitmapDecode(c.payload, z.tensor() if hasattr(z, "tensor") else z)
        return z if hasattr(z, "tensor") else INDArray(z)


class BasicNDArrayCompressor:
    _inst = None

    def __init__(self):
        self.codecs = {"FLOAT16": _CastCompressor("FLOAT16", torch.float16), "BFLOAT16": _CastCompressor("BFLOAT16", torch.bfloat16),
                       "INT8": _CastCompressor("INT8", torch.int8), "NOOP": _CastCompressor("NOOP", None),
                       THRESHOLD: _ThresholdCompressor(), BITMAP: _BitmapCompressor()}
        self.default = "FLOAT16"

    @classmethod
    def getInstance(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def getAvailableCompressors(self):
        return set(self.codecs)

    def setDefaultCompression(self, algo):
        self.default = algo.upper()
        return self

    def getDefaultCompression(self):
        return self.default

    def getCompressor(self, algo=None):
        return self.codecs[(algo or self.default).upper()]

    def compress(self, arr, algo=None):
        return self.getCompressor(algo).compress(arr)

    def decompress(self, c):
        return self.codecs[c.algorithm].decompress(c)

    def isCompressed(self, x):
        return isinstance(x, CompressedArray)


# ---- upstream ParallelWrapper / SharedTrainingMaster knobs -------------------------
class ThresholdAlgorithm:
    def calculateThreshold(self, iteration, epoch, lastThreshold, lastWasDense, lastSparsityRatio, updates):
        raise NotImplementedError


class FixedThresholdAlgorithm(ThresholdAlgorithm):
    def __init__(self, threshold=1e-3):
        self.threshold = threshold

    def calculateThreshold(self, *a):
        return self.threshold


class AdaptiveThresholdAlgorithm(ThresholdAlgorithm):
    """Keeps the encoded sparsity ratio inside [minTargetSparsity, maxTargetSparsity] by scaling tau."""

    def __init__(self, initialThreshold=1e-3, minTargetSparsity=1e-4, maxTargetSparsity=1e-2, decayRate=0.965936):
        self.initial, self.minS, self.maxS, self.decay = initialThreshold, minTargetSparsity, maxTargetSparsity, decayRate

    def calculateThreshold(self, iteration, epoch, lastThreshold, lastWasDense, lastSparsityRatio, updates):
        if lastThreshold is None:
            return self.initial
        if lastWasDense or (lastSparsityRatio is not None and lastSparsityRatio > self.maxS):
            return lastThreshold / self.decay
        if lastSparsityRatio is not None and lastSparsityRatio < self.minS:
            return lastThreshold * self.decay
        return lastThreshold


class TargetSparsityThresholdAlgorithm(AdaptiveThresholdAlgorithm):
    def __init__(self, initialThreshold=1e-3, sparsityTarget=1e-3, decayRate=0.965936):
        super().__init__(initialThreshold, sparsityTarget, sparsityTarget, decayRate)


class ResidualPostProcessor:
    def processResidual(self, iteration, epoch, lastThreshold, residual):
        raise NotImplementedError


class ResidualClippingPostProcessor(ResidualPostProcessor):
    def __init__(self, thresholdMultipleClipValue=5.0, frequency=5):
        self.mult, self.freq = thresholdMultipleClipValue, frequency

    def processResidual(self, iteration, epoch, lastThreshold, residual):
        if self.freq > 0 and iteration > 0 and iteration % self.freq == 0:
            c = self.mult * lastThreshold
            (residual.tensor() if hasattr(residual, "tensor") else residual).clamp_(-c, c)


class EncodedGradientsAccumulator:
    """Threshold-encoded gradient sharing with residuals (upstream EncodedGradientsAccumulator), expressed on
    torch.distributed: each rank encodes its update, messages are all-gathered, every rank decodes all of them."""

    def __init__(self, thresholdAlgorithm=None, residualPostProcessor=None, boundary=None):
        self.algo = thresholdAlgorithm or AdaptiveThresholdAlgorithm()
        self.post = residualPostProcessor
        self.boundary = boundary
        self.residual = None
        self.lastThreshold = None
        self.lastRatio = None
        self.iteration = 0

    def storeUpdate(self, grad: torch.Tensor):
        """Adds grad to the residual, encodes; returns the list of messages from all ranks."""
        import torch.distributed as dist
        g = grad.reshape(-1)
        if self.residual is None:
            self.residual = torch.zeros_like(g)
        self.residual += g
        tau = self.algo.calculateThreshold(self.iteration, 0, self.lastThreshold, False, self.lastRatio, self.residual)
        enc = ThresholdCompression.thresholdEncode(self.residual, tau, self.boundary)
        self.lastThreshold = tau
        self.lastRatio = 0.0 if enc is None else (enc.numel() - 4) / g.numel()
        if self.post is not None:
            self.post.processResidual(self.iteration, 0, tau, self.residual)
        self.iteration += 1
        if enc is None:
            enc = torch.tensor([0, g.numel(), struct.unpack("<i", struct.pack("<f", tau))[0], 0], dtype=torch.int32, device=g.device)
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return [enc]
        w = dist.get_world_size()
        sizes = [torch.zeros(1, dtype=torch.int64, device=g.device) for _ in range(w)]
        dist.all_gather(sizes, torch.tensor([enc.numel()], dtype=torch.int64, device=g.device))
        m = int(max(int(s) for s in sizes))
        pad = torch.zeros(m, dtype=torch.int32, device=g.device)
        pad[:enc.numel()] = enc
        outs = [torch.empty(m, dtype=torch.int32, device=g.device) for _ in range(w)]
        dist.all_gather(outs, pad)
        return [o[:int(s)] for o, s in zip(outs, sizes)]

    def applyUpdate(self, messages, target: torch.Tensor, average=True):
        z = target.reshape(-1)
        z.zero_()
        for m in messages:
            if int(m[0]) > 0:
                ThresholdCompression.thresholdDecode(m.contiguous(), z)
        if average and len(messages) > 1:
            z.mul_(1.0 / len(messages))
        return target
