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

TILES = ((128, 128), (128, 64), (64, 64), (64, 128), (256, 256), (128, 128), (128, 64), (64, 64), (64, 128), (64, 64), (256, 256),
         (128, 96), (96, 128))   # 5..9: deeper pipelines of 0..3 / 2; 10: ping-pong with half-tile staging; 11 / 12: 96-wide tiles (4 x 1 / 1 x 4 waves)
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


PP_MIN_BLOCKS = 176


def pick_tile(M, N, K, dtype):
    """Measured on MI355X (tools/gemm_tile_sweep.py -> profiles/r3_gemm_tile_sweep.md): the 64-row tiles win everywhere below the
    ping-pong regime because three of their workgroups fit a CU's LDS (48 KB each with two stages) and hide each other's barriers; the
    128 x 128 tile (two per CU) and the three / four stage variants (5..9) lose on every shape of the sweep."""
    def blocks(t):
        bm, bn = TILES[t]
        return ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    if dtype == torch.bfloat16 and blocks(10) >= PP_MIN_BLOCKS and K >= 256:
        return 10                                 # 256x256 ping-pong with half-tile staging once it covers ~2/3 of the CUs
    if N >= 128 and blocks(3) >= 512:
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
    if t in (4, 10) and (x.dtype != torch.bfloat16 or out_dtype != torch.bfloat16):
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
    if t in (4, 10):
        t = 0
    _launch(a, b, y, None, None, alpha, None, M, N, K, a.stride(1), b.stride(1), N, a.stride(0), b.stride(0), M * N, B, t,
            out_dtype == torch.float32 and a.dtype != torch.float32)
    return y


# ------------------------------------------------------------------------------------------------------------------ all operand layouts
def _layout(t):
    """(kmajor, ld) of a 2-D / batched operand view whose LAST TWO dims are (rows, k): K-contiguous -> (False, row stride);
    row-contiguous (a transposed view) -> (True, k stride); None when neither dim is dense."""
    if t.stride(-1) == 1:
        return False, t.stride(-2)
    if t.stride(-2) == 1:
        return True, t.stride(-1)
    return None


def supported_ex(a, b):
    """a: [.., M, K] view, b: [.., N, K] view (i.e. C = a . b^T), each either K-contiguous or its transpose; same dtype; 16-byte chunks."""
    if not (a.is_cuda and b.is_cuda and a.dtype == b.dtype and a.dtype in _DT and a.dim() == b.dim() and a.dim() in (2, 3)):
        return False
    if a.shape[-1] != b.shape[-1] or a.numel() == 0 or b.numel() == 0 or (a.dim() == 3 and a.shape[0] != b.shape[0]):
        return False
    la, lb = _layout(a), _layout(b)
    if la is None or lb is None:
        return False
    ch = chunk(a.dtype)
    M, K, N = a.shape[-2], a.shape[-1], b.shape[-2]
    if la[1] % ch or lb[1] % ch or a.data_ptr() % 16 or b.data_ptr() % 16:
        return False
    if (not la[0] or not lb[0]) and K % ch:
        return False
    if (la[0] and M % ch) or (lb[0] and N % ch):
        return False
    if a.dim() == 3 and (a.stride(0) % ch or b.stride(0) % ch):
        return False
    return True


def pick_splits(M, N, K, tile, dtype):
    """Split-K factor for skinny outputs (dense weight gradients: K = tokens): enough workgroups for ~2 per CU, K ranges of >= 4 tiles."""
    bm, bn = TILES[tile]
    blocks = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    bk = 32 if dtype == torch.float32 else 64
    ktiles = (K + bk - 1) // bk
    if blocks >= 384 or ktiles < 8:
        return 1
    return max(1, min(ktiles // 4, (512 + blocks - 1) // blocks))


def gemm(a, b, bias=None, act=None, residual=None, alpha=1.0, out=None, out_dtype=None, tile=None, splits=None, accum=False):
    """C = act(alpha * a . b^T + bias) (+ residual) for operand VIEWS in any of the four layouts: pass ``w`` ([N, K]) for the NT form,
    ``w.t()`` of a [K, N] matrix for NN, ``x.t()`` of a [K, M] matrix for TN - no transpose copy is made (native/csrc/gemm_trans.h).
    3-D operands are a strided batch. ``splits`` > 1 (or None = heuristic) cuts K over workgroups (fp32 output, atomics)."""
    assert supported_ex(a, b), "gemm: same-dtype 2-D / 3-D views, each K-contiguous or transposed, with 16-byte aligned chunks"
    batched = a.dim() == 3
    M, K, N = a.shape[-2], a.shape[-1], b.shape[-2]
    B = a.shape[0] if batched else 1
    (ta, lda), (tb, ldb) = _layout(a), _layout(b)
    out_dtype = out_dtype or (out.dtype if out is not None else a.dtype)
    assert out_dtype in (a.dtype, torch.float32)
    f32out = out_dtype == torch.float32
    t = (pick_tile(M, N, K, a.dtype) if tile is None else tile)
    if (t in (4, 10) and not ta and not tb and not batched and not accum and out is None and splits in (None, 1)
            and out_dtype == torch.bfloat16 and a.dtype == torch.bfloat16):
        return gemm_nt(a, b, bias=bias, act=act, residual=residual, alpha=alpha, tile=t)      # plain NT product: the ping-pong kernel
    t = 0 if t in (4, 10) else t
    if splits is None:
        splits = pick_splits(M, N, K, t, a.dtype) if (f32out and act in (None, "identity") and not batched) else 1
    if splits > 1:
        assert f32out and act in (None, "identity"), "split-K needs an fp32 output and a linear epilogue"
    if out is None:
        y = (torch.zeros if splits > 1 else torch.empty)((B, M, N) if batched else (M, N), device=a.device, dtype=out_dtype)
    else:
        y = out
        assert y.is_contiguous() and y.dtype == out_dtype and y.numel() == B * M * N
        if splits > 1 and not accum:
            y.zero_()
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
        bias = bias.float().contiguous()
    if residual is not None:
        assert residual.shape == y.shape and residual.dtype == out_dtype and residual.is_contiguous()
    d = (ctypes.c_int64 * 18)(M, N, K, lda, ldb, N, a.stride(0) if batched else 0, b.stride(0) if batched else 0, M * N, B, t, _DT[a.dtype],
                              1 if (f32out and a.dtype != torch.float32) else 0, ACTS[act], 1 if ta else 0, 1 if tb else 0, int(splits),
                              1 if (accum and splits <= 1) else 0)
    L.check(L.hip().dl4j_gemm_ex2(L.ptr(a), L.ptr(b), L.ptr(y), L.ptr(bias), L.ptr(residual), ctypes.c_float(alpha), d, L.stream()), "gemm_ex2")
    return y


def matmul(x, w, **kw):
    """Row-major x [.., M, K] @ w [.., K, N] without copies: w enters as the K-major operand."""
    return gemm(x, w.transpose(-1, -2), **kw)
