"""Python side of the hand-written gfx950 implicit-GEMM convolution (native/csrc/conv_igemm.hip).

forward / dgrad / wgrad on bf16 channels-last activations with OHWI bf16 filters and fp32
accumulation. ``forward`` can return the per-channel (sum, sum of squares) of its output so a
following BatchNorm skips its statistics pass.
"""
from __future__ import annotations

import ctypes

import torch

from ..native import lib as L

_TILES = ((128, 128), (128, 64), (64, 64))
_MAX_ELEMS = (1 << 31) - 1
_WS = {}


def _ws(n_floats, device, tag):
    """Stream-keyed scratch, allocated ZEROED: the wgrad scratch relies on it (its consumer re-zeroes what it read)."""
    key = (tag, device.index, torch.cuda.current_stream().cuda_stream)
    t = _WS.get(key)
    if t is None or t.numel() < n_floats:
        t = torch.zeros(max(int(n_floats), 1 << 18), device=device, dtype=torch.float32)
        _WS[key] = t
    return t


def _ohwi_contig(w):
    return w.dim() == 4 and w.permute(0, 2, 3, 1).is_contiguous()


def supported(x, w, stride, padding, dilation):
    if not (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 4):
        return False
    K, C = w.shape[0], w.shape[1]
    if C % 64 or K % 64 or x.shape[1] != C:
        return False
    if not x.is_contiguous(memory_format=torch.channels_last) or not _ohwi_contig(w):
        return False
    # operands of 2^31 elements and more run on the 64-bit-offset instantiation of the kernel (native/csrc/conv_igemm.hip BIG);
    # the row count N * OH * OW itself stays a 32-bit quantity
    if x.shape[0] * x.shape[2] * x.shape[3] > _MAX_ELEMS:
        return False
    return True


_FORCE_TILE = None       # tools/tile_sweep.py sets this to measure every tile on every layer shape


def pick_tile(M, K):
    if _FORCE_TILE is not None and K % _TILES[_FORCE_TILE][1] == 0:
        return _FORCE_TILE
    # measured per layer shape with tools/tile_sweep.py (profiles/r1_tile_sweep.md): the largest tile wins as soon as it
    # gives 1.5 blocks per CU; with 64 output channels the 64x64 tile beats 128x64 (these layers are HBM-bound and the
    # smaller tile doubles the resident blocks per CU)
    if K == 64:
        return 2
    best, best_blocks = None, -1
    for t, (bm, bn) in enumerate(_TILES):
        if K % bn:
            continue
        blocks = ((M + bm - 1) // bm) * (K // bn)
        if blocks >= 384:
            return t
        if blocks > best_blocks:
            best, best_blocks = t, blocks
    return best


def _out_hw(H, W, R, S, stride, padding, dilation):
    OH = (H + 2 * padding[0] - dilation[0] * (R - 1) - 1) // stride[0] + 1
    OW = (W + 2 * padding[1] - dilation[1] * (S - 1) - 1) // stride[1] + 1
    return OH, OW


class StatPartials:
    """Per-M-tile (sum, sum of squares) rows written by the conv epilogue: tensor [gy, 2, K] fp32. The consuming BatchNorm
    reduces them inside its finalize kernel (ops/batchnorm.batch_stats), so no separate reduction launch is needed."""
    __slots__ = ("partial", "gy", "K", "pivot")

    def __init__(self, partial, gy, K, pivot=None):
        """``pivot``: (running mean, folded conv bias or None) when the rows hold sums of (y - pivot), pivot = mean - bias."""
        self.partial, self.gy, self.K, self.pivot = partial, gy, K, pivot

    def sums(self):
        if self.pivot is not None:
            raise ValueError("shifted partial sums are finalized by ops/batchnorm.batch_stats, not summed raw")
        out = torch.empty(2 * self.K, device=self.partial.device, dtype=torch.float32)
        L.check(L.hip().dl4j_bn_sum_partials(L.ptr(self.partial), L.c_int(self.gy), L.c_int(self.K), L.ptr(out), None, None, L.stream()),
                "bn_sum_partials")
        return out


class MaskedGrad:
    """A gradient that is ``src * mask`` with the mask kept as bits (one byte per 8 channels, channels-last element order):
    dz = d_out * [out > 0] of a residual block, handed to the identity shortcut WITHOUT being written — the data-gradient
    launch that accumulates into the shortcut reads ``src`` and the bits in its epilogue. ``materialize()`` is the fallback for
    any other consumer."""
    __slots__ = ("src", "bits")

    def __init__(self, src, bits):
        self.src, self.bits = src, bits

    @property
    def shape(self):
        return self.src.shape

    @property
    def dtype(self):
        return self.src.dtype

    def materialize(self):
        s = self.src
        N, C, H, W = s.shape
        phys = s.permute(0, 2, 3, 1)                           # [N,H,W,C] contiguous view of the channels-last tensor
        sh = torch.arange(8, device=s.device, dtype=torch.uint8)
        m = ((self.bits.view(-1, 1) >> sh) & 1).to(torch.bool).reshape(N, H, W, C)
        return torch.where(m, phys, torch.zeros((), device=s.device, dtype=s.dtype)).permute(0, 3, 1, 2)

    def __add__(self, other):
        o = other.materialize() if isinstance(other, MaskedGrad) else other
        return self.materialize() + o

    __radd__ = __add__

    @staticmethod
    def ok(src, bits):
        return (src is not None and bits is not None and src.dim() == 4 and src.dtype == torch.bfloat16 and src.shape[1] % 8 == 0
                and src.is_contiguous(memory_format=torch.channels_last) and bits.numel() * 8 == src.numel()
                and src.data_ptr() % 16 == 0)


class BnBackwardSink:
    """Request to a data-gradient launch: "your output is dy of this BatchNorm(+ReLU): emit its backward partial sums".
    ``y`` is the BatchNorm input as [M, C] (the memory of the channels-last tensor), ``ms``/``mt`` recompute a fused ReLU's
    mask from y, ``bits`` is the saved mask of a residual block. After the launch ``partial`` ([gy, 2, C] fp32: per-tile sum dz and
    sum dz * xhat) and ``gy`` are set; ops/batchnorm.backward(pre_partials=...) then skips its column-reduction pass."""
    __slots__ = ("y", "mean", "invstd", "ms", "mt", "bits", "partial", "gy", "dbeta_out", "dgamma_out", "sums")

    def __init__(self, y, mean, invstd, ms=None, mt=None, bits=None, dbeta_out=None, dgamma_out=None):
        self.y, self.mean, self.invstd, self.ms, self.mt, self.bits = y, mean, invstd, ms, mt, bits
        self.partial, self.gy = None, 0
        # where the finished sums go (fp32 contiguous gradient views or None); ``sums`` ([2C]) is set by whoever ran the second stage
        # of the reduction - the weight-gradient launch that follows the data-gradient launch piggy-backs it (wgrad(tail=...))
        self.dbeta_out, self.dgamma_out, self.sums = dbeta_out, dgamma_out, None

    def tail_ok(self):
        return (self.partial is not None and self.sums is None
                and all(t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda) for t in (self.dbeta_out, self.dgamma_out)))

    def ok(self, x_shape):
        N, C, H, W = x_shape
        y = self.y
        if not (y is not None and y.dtype == torch.bfloat16 and y.is_contiguous() and y.numel() == N * C * H * W
                and y.shape[-1] == C and y.data_ptr() % 16 == 0):
            return False
        for t in (self.mean, self.invstd, self.ms, self.mt):
            if t is not None and not (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == C and t.data_ptr() % 16 == 0):
                return False
        if self.mean is None or self.invstd is None or (self.ms is None) != (self.mt is None):
            return False
        if self.bits is not None and not (self.bits.dtype == torch.uint8 and self.bits.numel() * 8 == y.numel()):
            return False
        return True


_EPI_TILE = int(__import__("os").environ.get("DL4J_AMD_EPI_TILE", "-1"))


def _launch_epi(x_t, w_t, y_t, geom, Kout, M, accum, acc, sink):
    """Data-gradient launch with the fused epilogue (masked accumulate and / or BatchNorm-backward partial sums)."""
    tile = pick_tile(M, Kout)
    if _EPI_TILE >= 0 and Kout % _TILES[_EPI_TILE][1] == 0 and tile < _EPI_TILE:
        tile = _EPI_TILE
    mt = int(L.hip().dl4j_igemm_mtiles(L.c_int64(M), L.c_int(tile)))
    g = (ctypes.c_int * 19)(*geom)
    partial = None
    if sink is not None:
        partial = torch.empty(mt * 2 * Kout, device=x_t.device, dtype=torch.float32)
    ts = (acc.src if acc is not None else None, acc.bits if acc is not None else None,
          sink.y if sink is not None else None, sink.mean if sink is not None else None,
          sink.invstd if sink is not None else None, sink.ms if sink is not None else None,
          sink.mt if sink is not None else None, sink.bits if sink is not None else None, partial)
    ptrs = (ctypes.c_void_p * 9)(*[0 if t is None else t.data_ptr() for t in ts])
    L.check(L.hip().dl4j_igemm_dgrad_bn(L.ptr(x_t), L.ptr(w_t), L.ptr(y_t), g, L.c_int(tile), L.c_int(1 if accum else 0), ptrs,
                                        L.stream()), "igemm_dgrad_bn")
    if sink is not None:
        sink.partial, sink.gy = partial, mt


def _launch(x_t, w_t, y_t, bias, geom, want_stats, Kout, M, accum=False, bn_fin=None, pivot=None):
    tile = pick_tile(M, Kout)
    stats = None
    mt = int(L.hip().dl4j_igemm_mtiles(L.c_int64(M), L.c_int(tile)))
    if want_stats:
        # own buffer per call (not the shared scratch): two convs may run before their fused BN node consumes both
        stats = torch.empty(mt * 2 * Kout, device=x_t.device, dtype=torch.float32)
    g = (ctypes.c_int * 19)(*geom)
    if want_stats and bn_fin is not None and bn_fin.ok() and x_t.numel() <= _MAX_ELEMS:
        from . import batchnorm as BN
        ngroups = (mt + 31) // 32
        ntiles = Kout // _TILES[tile][1]
        tk = BN.ticket_ws(x_t.device) if (BN._FUSE_STAGE2 and ntiles * (1 + ngroups) <= BN.TICKET_COUNTERS
                                          and ngroups * 2 * Kout <= BN.TICKET_GPART) else None
        if tk is not None:
            # the consuming BatchNorm's finalize runs in the last workgroup of this launch (native/csrc/ticket.h)
            buf = torch.empty(4 * Kout, device=x_t.device, dtype=torch.float32)
            f = bn_fin
            ptrs = (ctypes.c_void_p * 11)(*[0 if t is None else t.data_ptr() for t in (
                f.gamma, f.beta, f.conv_bias, f.gmean, f.gvar, buf[:Kout], buf[Kout:2 * Kout], buf[2 * Kout:3 * Kout], buf[3 * Kout:],
                f.dgmean, f.dgvar)])
            fp = (ctypes.c_float * 3)(f.eps, f.decay, float(f.stat_mode))
            L.check(L.hip().dl4j_igemm_conv_fin(L.ptr(x_t), L.ptr(w_t), L.ptr(y_t), L.ptr(bias), L.ptr(stats), g, L.c_int(tile),
                                                L.c_int(1 if accum else 0), None, ptrs, fp, L.ptr(tk[1]), L.ptr(tk[0]), L.stream()),
                    "igemm_conv_fin")
            return BN.Finalized(buf, Kout)
    if want_stats and pivot is not None and pivot[0] is not None:
        pm, pb = pivot
        ok = all(t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == Kout and t.device == x_t.device) for t in (pm, pb))
        if ok:
            L.check(L.hip().dl4j_igemm_conv_piv(L.ptr(x_t), L.ptr(w_t), L.ptr(y_t), L.ptr(bias), L.ptr(stats), g, L.c_int(tile),
                                                L.c_int(1 if accum else 0), None, L.ptr(pm), L.ptr(pb), L.stream()), "igemm_conv_piv")
            return StatPartials(stats, mt, Kout, (pm, pb))
    L.check(L.hip().dl4j_igemm_conv(L.ptr(x_t), L.ptr(w_t), L.ptr(y_t), L.ptr(bias), L.ptr(stats), g, L.c_int(tile),
                                    L.c_int(1 if accum else 0), None, L.stream()), "igemm_conv")
    if not want_stats:
        return None
    return StatPartials(stats, mt, Kout)


def forward(x, w, b, stride, padding, dilation, want_stats=False, bn_fin=None, stat_pivot=None):
    """Returns (y, stats) — stats is a StatPartials (per-tile sum / sumsq rows) when want_stats, or a batchnorm.Finalized when
    ``bn_fin`` (batchnorm.FinalizeRequest of the consuming layer) was given and the finalize ran inside the conv launch."""
    N, C, H, W = x.shape
    K, _, R, S = w.shape
    OH, OW = _out_hw(H, W, R, S, stride, padding, dilation)
    y = torch.empty((N, K, OH, OW), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    M = N * OH * OW
    geom = (N, H, W, C, OH, OW, K, R, S, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
            OH, OW, 1, 1)
    bias = b.float().contiguous() if b is not None else None
    sums = _launch(x, w, y, bias, geom, want_stats, K, M, bn_fin=bn_fin, pivot=stat_pivot)
    return y, sums


def wgrad_supported(x, w):
    """Split-K weight gradient kernels: bf16 / fp16 (transposing-read MFMA kernel) or fp32 (exact fp32 MFMA kernel) channels-last
    activations; channel counts multiples of 8 (fp32: 4)."""
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16, torch.float32) and w.dtype == x.dtype and x.dim() == 4):
        return False
    K, C = w.shape[0], w.shape[1]
    ch = 4 if x.dtype == torch.float32 else 8
    return (C % ch == 0 and K % ch == 0 and x.shape[1] == C and x.is_contiguous(memory_format=torch.channels_last)
            and x.numel() <= _MAX_ELEMS and x.data_ptr() % 16 == 0)


def dgrad_supported(w, stride, padding, dz=None):
    """``dz``: the incoming gradient; dgrad reuses the forward kernel with dz as the gathered operand, whose offsets are 32-bit."""
    if dz is not None and dz.shape[0] * dz.shape[2] * dz.shape[3] > _MAX_ELEMS:
        return False
    R, S = w.shape[2], w.shape[3]
    if tuple(stride) == (1, 1):
        return True
    return R == 1 and S == 1 and tuple(padding) == (0, 0)


def accum_ok(t, x_shape):
    """Can ``t`` (an already produced gradient for the same tensor) be accumulated into in place by the dgrad epilogue?"""
    return (t is not None and t.dtype == torch.bfloat16 and tuple(t.shape) == tuple(x_shape) and t.dim() == 4
            and t.is_contiguous(memory_format=torch.channels_last) and t.data_ptr() % 16 == 0)


def dgrad(dz, x_shape, w_dgrad, stride, padding, dilation, accum_into=None, acc_masked=None, bn_sink=None):
    """dx for y = conv(x, w). ``w_dgrad``: [C][R][S][K] spatially flipped filters (bf16, contiguous).
    ``accum_into``: existing gradient tensor for x (see accum_ok); the epilogue adds into it in place and it is returned.
    ``acc_masked``: a MaskedGrad for x, added in the epilogue into a NEW tensor (never materialised on its own).
    ``bn_sink``: BnBackwardSink — dx is the gradient into that BatchNorm; the epilogue emits its backward partial sums."""
    N, C, H, W = x_shape
    _, K, OH, OW = dz.shape
    R, S = w_dgrad.shape[1], w_dgrad.shape[2]
    if not dz.is_contiguous(memory_format=torch.channels_last):
        dz = dz.contiguous(memory_format=torch.channels_last)
    acc = accum_into is not None
    assert not (acc and acc_masked is not None)
    epi = acc_masked is not None or bn_sink is not None
    if tuple(stride) == (1, 1):
        dx = accum_into if acc else torch.empty((N, C, H, W), device=dz.device, dtype=torch.bfloat16,
                                                memory_format=torch.channels_last)
        ph = (R - 1) * dilation[0] - padding[0]
        pw = (S - 1) * dilation[1] - padding[1]
        geom = (N, OH, OW, K, H, W, C, R, S, 1, 1, ph, pw, dilation[0], dilation[1], H, W, 1, 1)
        if epi:
            _launch_epi(dz, w_dgrad, dx, geom, C, N * H * W, acc, acc_masked, bn_sink)
        else:
            _launch(dz, w_dgrad, dx, None, geom, False, C, N * H * W, accum=acc)
        return dx
    # 1x1 strided: rows scatter into a zeroed dx (or add into the positions they hit of an existing gradient). A masked
    # accumulate would have to touch the rows this launch skips as well: the caller materialises it instead.
    assert acc_masked is None
    dx = accum_into if acc else torch.empty((N, C, H, W), device=dz.device, dtype=torch.bfloat16,
                                            memory_format=torch.channels_last).zero_()
    geom = (N, OH, OW, K, OH, OW, C, 1, 1, 1, 1, 0, 0, 1, 1, H, W, stride[0], stride[1])
    if epi:
        _launch_epi(dz, w_dgrad, dx, geom, C, N * OH * OW, acc, None, bn_sink)
    else:
        _launch(dz, w_dgrad, dx, None, geom, False, C, N * OH * OW, accum=acc)
    return dx


TAIL_SUMS = __import__("os").environ.get("DL4J_AMD_WGRAD_TAIL", "1") == "1"


class PendingSums:
    """Second stages of column reductions that nobody reads before the optimizer step (layer-norm gain / bias gradients): instead of a
    5 us launch each they wait here and ride on the next weight-gradient launch as extra workgroups (WGradP::tail_*). ``flush`` runs
    what is left as ordinary launches; ops/helpers.layer_norm_bwd registers it as an end-of-backward callback of the autograd engine."""
    items = []                       # (partial [gy, 2, C] fp32, gy, C, sums [2C] fp32 destination)
    armed = False

    @classmethod
    def push(cls, partial, gy, C, sums):
        cls.items.append((partial, gy, C, sums))
        if not cls.armed:
            cls.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(cls.flush)

    @classmethod
    def pop(cls, device):
        for k, it in enumerate(cls.items):
            if it[0].device == device:
                return cls.items.pop(k)
        return None

    @classmethod
    def flush(cls):
        cls.armed = False
        items, cls.items = cls.items, []
        for partial, gy, C, sums in items:
            with torch.cuda.device(partial.device):
                L.check(L.hip().dl4j_bn_sum_partials(L.ptr(partial), L.c_int(gy), L.c_int(C), L.ptr(sums), None, None, L.stream()),
                        "pending_param_sums")


def _wgrad_launch(dz, x, tgt, g, target_blocks, prezeroed, dt, tail, db_out=None, db_of_x=False):
    """bf16 / fp16 weight-gradient launch; with ``tail`` (a BnBackwardSink holding partial rows) the same launch also finishes that
    BatchNorm's backward sums and the sink is marked done (``sums`` set); without one, a pending deferred reduction (PendingSums) rides
    along. ``db_out`` (1x1 problems): fp32 vector that receives += the column sums of the gradient operand (a dense layer's bias
    gradient) from the same launch - of dz, or of x with ``db_of_x``."""
    part = gy = C = sums = o0 = o1 = None
    if tail is not None and TAIL_SUMS and tail.tail_ok():
        C = tail.y.shape[-1]
        sums = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        part, gy, o0, o1 = tail.partial, tail.gy, tail.dbeta_out, tail.dgamma_out
        tail.sums = sums
    elif PendingSums.items and TAIL_SUMS:
        it = PendingSums.pop(x.device)
        if it is not None:
            part, gy, C, sums = it
    if db_out is not None:
        assert db_out.dtype == torch.float32 and db_out.is_contiguous() and db_out.numel() == (g[3] if db_of_x else g[6])
    if part is None and db_out is None:
        L.check(L.hip().dl4j_igemm_wgrad_dt(L.ptr(dz), L.ptr(x), L.ptr(tgt), g, L.c_int(target_blocks), L.c_int(prezeroed), dt, L.stream()),
                "igemm_wgrad")
        return
    L.check(L.hip().dl4j_igemm_wgrad_riders(L.ptr(dz), L.ptr(x), L.ptr(tgt), g, L.c_int(target_blocks), L.c_int(prezeroed), dt,
                                            L.ptr(db_out), L.c_int(2 if db_of_x else 1), L.ptr(part), L.c_int(gy or 0), L.c_int(C or 0),
                                            L.ptr(sums), L.ptr(o0), L.ptr(o1), L.stream()), "igemm_wgrad_riders")


def wgrad(dz, x, w_shape, stride, padding, dilation, dw_out=None, target_blocks=0, prezeroed=False, tail=None, db_out=None,
          db_of_x=False):
    """dW. If ``dw_out`` (fp32, contiguous, logical [K,C,R,S]) is given the result is written there and
    None is returned; otherwise a new fp32 [K,C,R,S] tensor is returned. ``prezeroed``: dw_out is known to be all
    zeros (the network clears the flat gradient vector once per iteration), so no per-layer memset is issued."""
    N, C, H, W = x.shape
    K, _, R, S = w_shape
    OH, OW = dz.shape[2], dz.shape[3]
    if not dz.is_contiguous(memory_format=torch.channels_last):
        dz = dz.contiguous(memory_format=torch.channels_last)
    ret = None
    if dw_out is None or not (dw_out.dtype == torch.float32 and dw_out.is_contiguous()):
        ret = torch.empty((K, C, R, S), device=x.device, dtype=torch.float32)
        tgt = ret
    else:
        tgt = dw_out
    g = (ctypes.c_int * 15)(N, H, W, C, OH, OW, K, R, S, stride[0], stride[1], padding[0], padding[1], dilation[0],
                            dilation[1])
    dt = L.c_int(1 if x.dtype == torch.float16 else 0)
    if dz.dtype != x.dtype:
        dz = dz.to(x.dtype)
    if x.dtype == torch.float32:
        if R * S == 1:
            L.check(L.hip().dl4j_wgrad_f32(L.ptr(dz), L.ptr(x), L.ptr(tgt), g, L.c_int(1 if (prezeroed and tgt is dw_out) else 0),
                                           L.stream()), "wgrad_f32")
        else:
            scratch = _ws(K * R * S * C, x.device, "igemm_wgrad")
            L.check(L.hip().dl4j_wgrad_f32(L.ptr(dz), L.ptr(x), L.ptr(scratch), g, L.c_int(1), L.stream()), "wgrad_f32")
            # single-split launches overwrite instead of accumulating: either way the transposer re-zeroes what it read
            L.check(L.hip().dl4j_ohwi_to_oihw(L.ptr(scratch), L.ptr(tgt), L.c_int(K), L.c_int(R * S), L.c_int(C), L.c_int(1),
                                              L.stream()), "ohwi_to_oihw")
    elif R * S == 1:
        _wgrad_launch(dz, x, tgt, g, target_blocks, 1 if (prezeroed and tgt is dw_out) else 0, dt, tail, db_out, db_of_x)
        db_out = None
    else:
        # scratch starts zeroed and the layout transposer re-zeroes exactly what it read: no memset per layer
        scratch = _ws(K * R * S * C, x.device, "igemm_wgrad")
        _wgrad_launch(dz, x, scratch, g, target_blocks, 1, dt, tail)
        L.check(L.hip().dl4j_ohwi_to_oihw(L.ptr(scratch), L.ptr(tgt), L.c_int(K), L.c_int(R * S), L.c_int(C), L.c_int(1),
                                          L.stream()), "ohwi_to_oihw")
    assert db_out is None, "db_out is served by the 1x1 bf16 / fp16 launch only (see db_fusable)"
    if ret is not None and dw_out is not None:
        dw_out.copy_(ret)
        return None
    return ret


def db_fusable(x, w_shape):
    """Can wgrad(..., db_out=...) take the bias gradient along? (1x1 problem on the bf16 / fp16 split-K kernel)"""
    return x.dtype in (torch.bfloat16, torch.float16) and w_shape[2] * w_shape[3] == 1
