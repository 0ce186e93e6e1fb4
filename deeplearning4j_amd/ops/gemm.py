"""Hand-written MFMA GEMM on the LDS-DMA core (native/csrc/gemm_core.h, conv_mfma.hip):

    y[M, N] = act(alpha * x[M, K] . w[N, K]^T + bias) (+ residual)      bf16 / fp16 / fp32 operands, fp32 accumulation

``w`` is in the transposed-B ("NT") layout, i.e. exactly a dense layer's [nOut, nIn] weight. Fused epilogues (bias, ReLU / GELU / tanh /
sigmoid, residual add, fp32 output) replace the separate elementwise launches of the reference, whose MmulHelper has no fast bf16
path at all (helpers/cuda_off/MmulHelper.cu:255-280 -> naive kernel :41-106) and synchronises the stream after every GEMM (:347).
Batched form ``bgemm_nt`` = helpers/cuda/batched_gemm.cu:145-193 (cublasGemmBatchedEx) for strided batches.
Measured against hipBLASLt: profiles/r2_gemm_microbench.md.
"""
from __future__ import annotations

import ctypes

import torch

from ..native import lib as L

TILES = ((128, 128), (128, 64), (64, 64), (64, 128), (256, 256))
_DT = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}
ACTS = {None: 0, "identity": 0, "relu": 1, "gelu": 2, "gelu_tanh": 2, "gelu_erf": 3, "tanh": 4, "sigmoid": 5}


def chunk(dtype):
    return 4 if dtype == torch.float32 else 8


def supported(x, w):
    """2-D operands, same dtype, K-contiguous rows with 16-byte aligned starts."""
    if not (x.is_cuda and w.is_cuda and x.dtype == w.dtype and x.dtype in _DT and x.dim() == 2 and w.dim() == 2):
        return False
    if x.shape[1] != w.shape[1] or x.shape[0] == 0 or w.shape[0] == 0:
        return False
    ch = chunk(x.dtype)
    return (x.stride(1) == 1 and w.stride(1) == 1 and x.shape[1] % ch == 0 and x.stride(0) % ch == 0 and w.stride(0) % ch == 0
            and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


def pick_tile(M, N, K, dtype):
    """Measured on MI355X with tools/gemm_lab (profiles/r2_gemm_lab_v2.txt, r2_gemm_lab_v4_pingpong.txt)."""
    def blocks(t):
        bm, bn = TILES[t]
        return ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    if dtype == torch.bfloat16 and blocks(4) >= 448 and K >= 2048:
        return 4                                  # 256x256 ping-pong: only when it fills the chip ~2x with a deep K loop
    if blocks(0) >= 1536 and K >= 1024:
        return 0
    if blocks(1) >= 640:
        return 1
    if N >= 128 and M < 128:
        return 3
    return 2


def _launch(x, w, y, bias, residual, alpha, act, M, N, K, lda, ldb, ldc, sA, sB, sC, batch, tile, out_f32, stats=None,
            stat_shift=None, accum=False):
    d = (ctypes.c_int64 * 14)(M, N, K, lda, ldb, ldc, sA, sB, sC, batch, tile, _DT[x.dtype], 1 if out_f32 else 0, ACTS[act])
    L.check(L.hip().dl4j_gemm_nt_ex(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(bias), L.ptr(residual), ctypes.c_float(alpha),
                                    L.ptr(stats), L.ptr(stat_shift), L.c_int(1 if accum else 0), d, L.stream()), "gemm_nt")


def gemm_nt_stats(x, w, y, bias, tile, want_stats, stat_shift=None, accum=False):
    """y[M, N] (preallocated, row stride N) = x . w^T + bias, optionally with the per-M-tile column statistics of the rounded output
    (returns the [mtiles, 2, N] fp32 partials) — the 1x1 / stride-1 convolution path of ops/conv_mfma.py."""
    M, K = x.shape
    N = w.shape[0]
    stats = None
    if want_stats:
        bm = TILES[tile][0]
        stats = torch.empty(((M + bm - 1) // bm) * 2 * N, device=x.device, dtype=torch.float32)
    _launch(x, w, y, bias, None, 1.0, None, M, N, K, x.stride(0), w.stride(0), N, 0, 0, 0, 1, tile, False, stats, stat_shift, accum)
    return stats


def gemm_nt(x, w, bias=None, act=None, residual=None, alpha=1.0, out_dtype=None, tile=None):
    """x [M, K], w [N, K] -> [M, N]."""
    assert supported(x, w), "gemm_nt: same-dtype 2-D operands with K-contiguous, 16-byte aligned rows are required"
    M, K = x.shape
    N = w.shape[0]
    out_dtype = out_dtype or x.dtype
    assert out_dtype in (x.dtype, torch.float32)
    y = torch.empty(M, N, device=x.device, dtype=out_dtype)
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
        bias = bias.float().contiguous()
    if residual is not None:
        assert residual.shape == y.shape and residual.dtype == out_dtype and residual.is_contiguous()
    t = pick_tile(M, N, K, x.dtype) if tile is None else tile
    if t == 4 and (x.dtype != torch.bfloat16 or out_dtype != torch.bfloat16):
        t = 0
    _launch(x, w, y, bias, residual, alpha, act, M, N, K, x.stride(0), w.stride(0), N, 0, 0, 0, 1, t, out_dtype == torch.float32
            and x.dtype != torch.float32)
    return y


def bgemm_nt(a, b, alpha=1.0, out_dtype=None):
    """a [B, M, K], b [B, N, K] (K-contiguous, uniform batch strides) -> [B, M, N]."""
    assert a.dim() == 3 and b.dim() == 3 and a.shape[0] == b.shape[0] and a.shape[2] == b.shape[2] and a.dtype == b.dtype
    B, M, K = a.shape
    N = b.shape[1]
    ch = chunk(a.dtype)
    assert a.stride(2) == 1 and b.stride(2) == 1 and K % ch == 0
    assert all(s % ch == 0 for s in (a.stride(0), a.stride(1), b.stride(0), b.stride(1)))
    out_dtype = out_dtype or a.dtype
    y = torch.empty(B, M, N, device=a.device, dtype=out_dtype)
    t = pick_tile(M, N, K, a.dtype)
    if t == 4:
        t = 0
    _launch(a, b, y, None, None, alpha, None, M, N, K, a.stride(1), b.stride(1), N, a.stride(0), b.stride(0), M * N, B, t,
            out_dtype == torch.float32 and a.dtype != torch.float32)
    return y
