"""Python side of the round-2 implicit-GEMM convolution on the LDS-DMA core (native/csrc/conv_mfma.hip).

Forward and data gradient for 1-D / 2-D / 3-D convolutions (and transposed convolutions, whose forward is a data gradient) in
bf16, fp16 and fp32 — any stride, padding and dilation, channel counts that are multiples of 8 (16-bit types) or 4 (fp32).
Activations are channels-last (N[D]HWC memory), filters O[D]HWI for the forward and I[D]HW O spatially flipped for the data
gradient. Everything is described to the kernel by a tap table (see the header of conv_mfma.hip); the table for a layer
geometry is built once and cached.

Reference behaviour: libnd4j conv2d = im2col + GEMM (helpers/cuda/convolutions_conv2d.cu:78-100), conv2d_bp = GEMMs + col2im
(convolutions_conv2dBP.cu:83-113), deconv2d / conv3dnew / conv1d in ops/declarable/generic/nn/convo/*.cpp.
"""
from __future__ import annotations

import ctypes
import os

import torch

from ..native import lib as L

TILES = ((128, 128), (128, 64), (64, 64), (64, 128))
_DT = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}
_DESC = {}
_FORCE_TILE = None if os.environ.get("DL4J_AMD_CONV_TILE") is None else int(os.environ["DL4J_AMD_CONV_TILE"])


def dtype_ok(t):
    return t in _DT


def chunk(dtype):
    return 4 if dtype == torch.float32 else 8


def pick_tile(M, K, depth):
    """Tile id for an [M, K] output with reduction depth ``depth`` (taps x channels). Measured with tools/gemm_lab on MI355X
    (profiles/r2_gemm_lab_v2.txt): these GEMMs are small, so the winner is the largest tile that still yields ~2 workgroups per CU."""
    if _FORCE_TILE is not None:
        return _FORCE_TILE
    def blocks(t):
        bm, bn = TILES[t]
        return ((M + bm - 1) // bm) * ((K + bn - 1) // bn)
    if K <= 64:
        return 2
    if blocks(0) >= 1536 and depth >= 1024:
        return 0
    if blocks(1) >= 640:
        return 1
    return 2


def mtiles(M, tile):
    bm = TILES[tile][0]
    return (M + bm - 1) // bm


class _Desc:
    __slots__ = ("arr", "M", "tile", "K")


def _make_desc(key, in_dims, C, out_grid, K, strides, pads, taps, wrow, out_dims, ostr, o0, dtype, accum, act, tile=None):
    """taps: list of (dd, dh, dw, weight_offset)."""
    d = _DESC.get(key)
    if d is not None:
        return d
    N, D, H, W = in_dims
    OD, OH, OW = out_grid
    M = N * OD * OH * OW
    if tile is None:
        tile = pick_tile(M, K, len(taps) * C)
    vals = [N, D, H, W, C, OD, OH, OW, K, strides[0], strides[1], strides[2], pads[0], pads[1], pads[2], len(taps), wrow,
            out_dims[0], out_dims[1], out_dims[2], ostr[0], ostr[1], ostr[2], o0[0], o0[1], o0[2], tile, _DT[dtype],
            1 if accum else 0, act]
    for t in taps:
        vals.extend(t)
    d = _Desc()
    d.arr = (ctypes.c_int64 * len(vals))(*vals)
    d.M, d.tile, d.K = M, tile, K
    _DESC[key] = d
    return d


def _run(x, w, y, bias, stats, shift, d):
    L.check(L.hip().dl4j_conv_mfma(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(bias), L.ptr(stats), L.ptr(shift), d.arr, None, L.stream()),
            "conv_mfma")


def _triple(v, fill):
    v = tuple(int(a) for a in v)
    return (fill,) * (3 - len(v)) + v


def _cl(x):
    """Channels-last memory for 3-D/4-D/5-D NC* tensors."""
    if x.dim() == 4:
        return x.is_contiguous(memory_format=torch.channels_last)
    if x.dim() == 5:
        return x.is_contiguous(memory_format=torch.channels_last_3d)
    return x.permute(0, 2, 1).is_contiguous()


def _w_cl(w):
    """Filters [K, C, *k] stored O*k I (channel innermost)."""
    perm = (0,) + tuple(range(2, w.dim())) + (1,)
    return w.permute(*perm).is_contiguous()


def supported(x, w, groups=1):
    if not (x.is_cuda and w.is_cuda and x.dtype == w.dtype and x.dtype in _DT and x.dim() in (3, 4, 5) and groups == 1):
        return False
    C = w.shape[1]
    if x.shape[1] != C or C % chunk(x.dtype):
        return False
    ntaps = 1
    for k in w.shape[2:]:
        ntaps *= k
    if ntaps > 128:
        return False
    return _cl(x) and _w_cl(w) and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0


def dgrad_supported(w):
    """The data gradient gathers dy rows, so the OUTPUT channel count must be a whole number of 16-byte chunks too."""
    return w.shape[0] % chunk(w.dtype) == 0


def out_size(i, k, s, p, d):
    return (i + 2 * p - d * (k - 1) - 1) // s + 1


def _empty_cl(shape, dtype, device):
    if len(shape) == 4:
        return torch.empty(shape, device=device, dtype=dtype, memory_format=torch.channels_last)
    if len(shape) == 5:
        return torch.empty(shape, device=device, dtype=dtype, memory_format=torch.channels_last_3d)
    N, C, Wd = shape
    return torch.empty((N, Wd, C), device=device, dtype=dtype).permute(0, 2, 1)


class StatPartials:
    """Per-M-tile (sum, sum of squares) rows written by the conv epilogue: tensor [gy, 2, K] fp32 (optionally of x - shift)."""
    __slots__ = ("partial", "gy", "K", "shift")

    def __init__(self, partial, gy, K, shift=None):
        self.partial, self.gy, self.K, self.shift = partial, gy, K, shift

    def sums(self):
        out = torch.empty(2 * self.K, device=self.partial.device, dtype=torch.float32)
        L.check(L.hip().dl4j_bn_sum_partials(L.ptr(self.partial), L.c_int(self.gy), L.c_int(self.K), L.ptr(out), None, None, L.stream()),
                "bn_sum_partials")
        return out


def forward(x, w, b, stride, padding, dilation, want_stats=False, stat_shift=None, act=0):
    """x [N, C, *spatial] channels-last, w [K, C, *k] (O*k I memory). Returns (y, StatPartials | None)."""
    nd = x.dim() - 2
    N, C = x.shape[0], x.shape[1]
    K = w.shape[0]
    isz = _triple(x.shape[2:], 1)
    ksz = _triple(w.shape[2:], 1)
    st, pd, dl = _triple(stride, 1), _triple(padding, 0), _triple(dilation, 1)
    osz = tuple(out_size(isz[i], ksz[i], st[i], pd[i], dl[i]) for i in range(3))
    if ksz == (1, 1, 1) and st == (1, 1, 1) and pd == (0, 0, 0) and act == 0:
        # 1x1 / stride 1: channels-last x IS the [N*spatial, C] matrix -> plain NT GEMM kernel (no gather, no tap table)
        from . import gemm as G
        M = N * isz[0] * isz[1] * isz[2]
        y = _empty_cl((N, K) + isz[3 - nd:], x.dtype, x.device)
        x2 = x.permute(0, *range(2, 2 + nd), 1).reshape(M, C)
        w2 = w.reshape(K, C) if w.is_contiguous() else w.permute(0, *range(2, 2 + nd), 1).reshape(K, C)
        y2 = y.permute(0, *range(2, 2 + nd), 1).reshape(M, K)
        assert x2.data_ptr() == x.data_ptr() and y2.data_ptr() == y.data_ptr() and w2.data_ptr() == w.data_ptr()
        tile = pick_tile(M, K, C)
        bias = b.float().contiguous() if b is not None else None
        stats = G.gemm_nt_stats(x2, w2, y2, bias, tile, want_stats, stat_shift)
        return y, (StatPartials(stats, mtiles(M, tile), K, stat_shift) if want_stats else None)
    key = ("f", N, C, K, isz, ksz, st, pd, dl, x.dtype, act)
    d = _DESC.get(key)
    if d is None:
        taps = []
        for a in range(ksz[0]):
            for r in range(ksz[1]):
                for s in range(ksz[2]):
                    taps.append((a * dl[0], r * dl[1], s * dl[2], ((a * ksz[1] + r) * ksz[2] + s) * C))
        d = _make_desc(key, (N,) + isz, C, osz, K, st, pd, taps, ksz[0] * ksz[1] * ksz[2] * C, osz, (1, 1, 1), (0, 0, 0),
                       x.dtype, False, act)
    y = _empty_cl((N, K) + osz[3 - nd:], x.dtype, x.device)
    bias = b.float().contiguous() if b is not None else None
    stats = None
    if want_stats:
        stats = torch.empty(mtiles(d.M, d.tile) * 2 * K, device=x.device, dtype=torch.float32)
    _run(x, w, y, bias, stats, stat_shift, d)
    return y, (StatPartials(stats, mtiles(d.M, d.tile), K, stat_shift) if want_stats else None)


def flip_filters(w):
    """[K, C, *k] -> data-gradient layout [C, *k flipped, K] (contiguous)."""
    nd = w.dim() - 2
    wf = torch.flip(w, dims=list(range(2, 2 + nd)))
    perm = (1,) + tuple(range(2, 2 + nd)) + (0,)
    return wf.permute(*perm).contiguous()


def dgrad(dz, x_shape, w_dgrad, stride, padding, dilation, accum_into=None):
    """dx of y = conv(x, w) for ANY stride. ``w_dgrad`` = flip_filters(w): [C, *k, K]. One launch per residue class of the input
    pixel index modulo the stride; a class without a valid tap writes zeros (or leaves ``accum_into`` untouched)."""
    nd = len(x_shape) - 2
    N, C = x_shape[0], x_shape[1]
    K = dz.shape[1]
    isz = _triple(x_shape[2:], 1)
    osz = _triple(dz.shape[2:], 1)
    ksz = _triple(w_dgrad.shape[1:1 + nd], 1)
    st, pd, dl = _triple(stride, 1), _triple(padding, 0), _triple(dilation, 1)
    if not _cl(dz):
        dz = dz.contiguous(memory_format=torch.channels_last if nd == 2 else torch.channels_last_3d) if nd >= 2 else \
            dz.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    acc = accum_into is not None
    dx = accum_into if acc else _empty_cl(tuple(x_shape), dz.dtype, dz.device)
    if ksz == (1, 1, 1) and st == (1, 1, 1) and pd == (0, 0, 0):
        from . import gemm as G
        M = N * isz[0] * isz[1] * isz[2]
        dz2 = dz.permute(0, *range(2, 2 + nd), 1).reshape(M, K)
        dx2 = dx.permute(0, *range(2, 2 + nd), 1).reshape(M, C)
        assert dz2.data_ptr() == dz.data_ptr() and dx2.data_ptr() == dx.data_ptr()
        G.gemm_nt_stats(dz2, w_dgrad.reshape(C, K), dx2, None, pick_tile(M, C, K), False, None, accum=acc)
        return dx
    wrow = ksz[0] * ksz[1] * ksz[2] * K
    for ca in range(st[0]):
        for cb in range(st[1]):
            for cc in range(st[2]):
                cls = (ca, cb, cc)
                grid = tuple((isz[i] - cls[i] + st[i] - 1) // st[i] if isz[i] > cls[i] else 0 for i in range(3))
                if min(grid) <= 0:
                    continue
                key = ("d", N, C, K, isz, osz, ksz, st, pd, dl, cls, dz.dtype, acc)
                d = _DESC.get(key)
                if d is None:
                    per_dim = []
                    for i in range(3):
                        lst = []
                        for r in range(ksz[i]):
                            num = cls[i] + pd[i] - r * dl[i]
                            if num % st[i] == 0:
                                lst.append((r, num // st[i]))
                        per_dim.append(lst)
                    taps = []
                    for (a, oa) in per_dim[0]:
                        for (r, orr) in per_dim[1]:
                            for (s, os_) in per_dim[2]:
                                fa, fr, fs = ksz[0] - 1 - a, ksz[1] - 1 - r, ksz[2] - 1 - s
                                taps.append((oa, orr, os_, ((fa * ksz[1] + fr) * ksz[2] + fs) * K))
                    if acc and not taps:
                        _DESC[key] = d = False
                    else:
                        d = _make_desc(key, (N,) + osz, K, grid, C, (1, 1, 1), (0, 0, 0), taps, wrow, isz, st, cls, dz.dtype,
                                       acc, 0)
                if d is False:
                    continue
                _run(dz, w_dgrad, dx, None, None, None, d)
    return dx
