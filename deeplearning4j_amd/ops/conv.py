"""conv2d forward / backward dispatch.

Paths (selected per call; analogue of libnd4j's PlatformHelper ``isUsable`` dispatch and the zoo
models' ``cudnnAlgoMode(PREFER_FASTEST)`` — reference: ops/declarable/impl/DeclarableOp.cpp:692-739,
platform/cudnn/conv2d.cu:83):

* ``hip``     – hand-written gfx950 implicit-GEMM MFMA kernels (native/csrc/conv_igemm.hip):
                bf16 NHWC activations, OHWI filters, fp32 accumulation; forward optionally fuses the
                per-channel batch statistics needed by a following BatchNorm into the epilogue.
* ``stem``    – hand-written direct MFMA kernels for 3-channel 7x7/2 stems (native/csrc/conv_stem.hip): forward with
                fused BatchNorm statistics, weight gradient.
* ``library`` – MIOpen through ATen (also the CPU / fp64 path used by gradient checks).

Weight layout contract matches DL4J: logical [nOut, nIn, kH, kW] (wFormat 1 of libnd4j conv2d,
generic/nn/convo/conv2d.cpp:46-59).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from ..native import lib as L

_FORCE = os.environ.get("DL4J_AMD_CONV", "auto")  # auto | hip | library
# auto: per shape the faster of the two hand-written families (tools/conv_core_compare.py, profiles/r2_conv_core_compare.md): the
#       round-1 register-staged kernels for bf16 layers with 64-multiple channels (stride 1, or strided 1x1), the round-2 LDS-DMA
#       core for everything round 1 cannot run (fp16 / fp32, other channel counts, strided KxK data gradients, 1-D / 3-D);
#       mfma2 / igemm force one family
_CORE = os.environ.get("DL4J_AMD_CONV_CORE", "auto")


def _mfma_eligible(x, w):
    """Round-2 core: bf16 / fp16 / fp32, any stride / padding / dilation, channels a multiple of 8 (fp32: 4)."""
    if _FORCE == "library" or _CORE == "igemm" or not L.use_hip(x, w):
        return False
    from . import conv_mfma
    return conv_mfma.supported(x, w)


def _prefer_r1(x, w, stride, padding, dilation):
    """auto mode: shapes where the round-1 kernels measured faster."""
    if _CORE != "auto" or not _hip_eligible(x, w, stride, padding, dilation):
        return False
    from . import conv_igemm
    if not conv_igemm.dgrad_supported(w, stride, padding):
        return False
    # measured in the full training step (profiles/r2_conv_core_compare.md): 10.26 ms with the round-1 kernels on every eligible
    # layer, 10.64 ms with the 1x1 stride-1 layers on the GEMM form of the new core, 11.98 ms with every layer on the new core
    return True


def _hip_eligible(x, w, stride, padding, dilation):
    if _FORCE == "library" or not L.use_hip(x, w):
        return False
    if x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16:
        return False
    try:
        from . import conv_igemm
    except Exception:
        return False
    return conv_igemm.supported(x, w, stride, padding, dilation)


def _stem_eligible(x, w, stride, padding, dilation):
    if _FORCE == "library" or not L.use_hip(x, w):
        return False
    from . import conv_stem
    return conv_stem.supported(x, w, stride, padding, dilation)


def conv2d_forward(x, w, b, stride, padding, dilation, want_stats=False, bn_fin=None, stat_pivot=None):
    """x: [N,C,H,W] logical (channels_last memory on GPU). Returns (z, sums): ``sums`` is the [2K] fp32
    per-channel (sum, sum of squares) of z when the HIP path fused it into its epilogue, else None."""
    if _stem_eligible(x, w, stride, padding, dilation) and x.shape[1] < 8:
        from . import conv_stem
        return conv_stem.forward(x, w, b, stride, padding, dilation, want_stats)
    if _mfma_eligible(x, w) and not _prefer_r1(x, w, stride, padding, dilation):
        from . import conv_mfma
        return conv_mfma.forward(x, w, b, stride, padding, dilation, want_stats)
    if _hip_eligible(x, w, stride, padding, dilation):
        from . import conv_igemm
        return conv_igemm.forward(x, w, b, stride, padding, dilation, want_stats, bn_fin=bn_fin, stat_pivot=stat_pivot)
    if _stem_eligible(x, w, stride, padding, dilation):
        from . import conv_stem
        return conv_stem.forward(x, w, b, stride, padding, dilation, want_stats)
    return F.conv2d(x, w, b, tuple(stride), tuple(padding), tuple(dilation)), None


def dgrad_epilogue_ok(dz, x, w, stride, padding, dilation, w_dgrad):
    """True when the data gradient of this call runs on the implicit-GEMM kernel whose epilogue can add a masked gradient and
    emit BatchNorm-backward partial sums (ops/conv_igemm.dgrad acc_masked / bn_sink)."""
    if w_dgrad is None or dz.dtype != torch.bfloat16 or dz.numel() > (1 << 31) - 1:      # fused epilogue: 32-bit operand offsets only
        return False
    if not (_hip_eligible(x, w, stride, padding, dilation) and _prefer_r1(x, w, stride, padding, dilation)):
        return False
    if _stem_eligible(x, w, stride, padding, dilation) and x.shape[1] < 8:
        return False
    from . import conv_igemm as IG
    return IG.dgrad_supported(w, stride, padding, dz)


def stem_wgrad_bnpool(lazy, x, w, stride, padding, dilation, dw_out):
    """Weight gradient of the 3-channel stem from an unevaluated bn_pool.PoolBnGrad; False when this call is not the stem kernels'."""
    if not (_stem_eligible(x, w, stride, padding, dilation) and x.shape[1] < 8 and lazy.dtype == torch.bfloat16
            and dw_out is not None and dw_out.dtype == torch.float32 and dw_out.is_contiguous()):
        return False
    from . import conv_stem
    oh, ow = conv_stem._out_hw(x, padding)
    if (lazy.geom[0], lazy.geom[1], lazy.geom[2]) != (x.shape[0], oh, ow) or not lazy.x2.is_contiguous() or not lazy.dy.is_contiguous():
        return False
    conv_stem.wgrad_bnpool(lazy, x, stride, padding, dilation, dw_out=dw_out)
    return True


def conv2d_backward(dz, x, w, stride, padding, dilation, need_dx=True, need_dw=True, need_db=False, w_dgrad=None,
                    dw_out=None, dx_accum=None, dw_prezeroed=False, dx_acc_masked=None, bn_sink=None):
    """Returns (dx, dW, db); dW has logical shape [nOut, nIn, kH, kW] — or None when it was written
    straight into ``dw_out`` (fp32 gradient view) by the HIP wgrad kernel.
    ``dx_acc_masked`` (conv_igemm.MaskedGrad): another gradient for x, added in the dgrad epilogue (materialised here when the
    kernel cannot take it). ``bn_sink`` (conv_igemm.BnBackwardSink): filled when the dgrad epilogue emitted the partial sums —
    the caller checks ``bn_sink.partial``."""
    if dx_acc_masked is not None or bn_sink is not None:
        fus = need_dx and dgrad_epilogue_ok(dz, x, w, stride, padding, dilation, w_dgrad)
        from . import conv_igemm as IG
        if dx_acc_masked is not None and not (fus and tuple(stride) == (1, 1) and dx_accum is None
                                              and tuple(dx_acc_masked.shape) == tuple(x.shape)
                                              and IG.MaskedGrad.ok(dx_acc_masked.src, dx_acc_masked.bits)):
            m = dx_acc_masked.materialize()
            dx_accum = m if dx_accum is None else dx_accum + m
            dx_acc_masked = None
        if bn_sink is not None and not (fus and bn_sink.ok(x.shape)):
            bn_sink = None
    if _mfma_eligible(x, w) and dz.dtype == x.dtype and not _prefer_r1(x, w, stride, padding, dilation) \
            and not (_stem_eligible(x, w, stride, padding, dilation) and x.shape[1] < 8):
        return _backward_mfma(dz, x, w, stride, padding, dilation, need_dx, need_dw, need_db, w_dgrad, dw_out, dx_accum,
                              dw_prezeroed)
    if _hip_eligible(x, w, stride, padding, dilation) and dz.dtype == torch.bfloat16:
        from . import conv_igemm as IG
        dx = dW = db = None
        if need_db:
            db = dz.float().sum((0, 2, 3))
        def _wgrad(tail=None):
            from . import streams
            if (streams.active() and dw_out is not None and dw_out.dtype == torch.float32 and dw_out.is_contiguous()
                    and dz.is_contiguous(memory_format=torch.channels_last)):
                # dW goes straight into the flat gradient view and is not read before the updater: side stream
                with streams.side_region(streams.fork(dz, x, dw_out), 'convolution weight-gradient'):
                    IG.wgrad(dz, x, w.shape, stride, padding, dilation, dw_out=dw_out, prezeroed=dw_prezeroed)
                return None
            return IG.wgrad(dz, x, w.shape, stride, padding, dilation, dw_out=dw_out, prezeroed=dw_prezeroed, tail=tail)
        # with a BatchNorm sink the data gradient runs FIRST: its epilogue emits the partial rows and the weight-gradient launch
        # (independent of them, next in the stream) sums them in extra workgroups - no separate second-stage launch
        wgrad_late = need_dw and need_dx and bn_sink is not None
        if need_dw and not wgrad_late:
            dW = _wgrad()
        if need_dx:
            if w_dgrad is not None and IG.dgrad_supported(w, stride, padding, dz):
                acc = dx_accum if IG.accum_ok(dx_accum, x.shape) else None
                if dx_accum is not None and acc is None:
                    bn_sink = None               # the sum is finished outside the kernel: its partial sums would be of a part only
                dx = IG.dgrad(dz, x.shape, w_dgrad, stride, padding, dilation, accum_into=acc, acc_masked=dx_acc_masked,
                              bn_sink=bn_sink)
                if dx_accum is not None and acc is None:
                    dx = dx_accum + dx
            else:
                dx = torch.ops.aten.convolution_backward(
                    dz.contiguous(memory_format=torch.channels_last), x, w, None, list(stride), list(padding),
                    list(dilation), False, [0, 0], 1, [True, False, False])[0]
                if dx_accum is not None:
                    dx = dx_accum + dx
        if wgrad_late:
            dW = _wgrad(tail=bn_sink if (bn_sink is not None and bn_sink.partial is not None) else None)
        return dx, dW, db
    if dz.dtype == torch.bfloat16 and _stem_eligible(x, w, stride, padding, dilation):
        from . import conv_stem
        dx = dW = db = None
        if need_db:
            db = dz.float().sum((0, 2, 3))
        if need_dw:
            dW = conv_stem.wgrad(dz, x, stride, padding, dilation, dw_out=dw_out)
        if need_dx:
            dx = torch.ops.aten.convolution_backward(
                dz.contiguous(memory_format=torch.channels_last), x, w, None, list(stride), list(padding), list(dilation), False,
                [0, 0], 1, [True, False, False])[0]
            if dx_accum is not None:
                dx = dx_accum + dx
        return dx, dW, db
    if x.is_cuda and not dz.is_contiguous(memory_format=torch.channels_last) and x.is_contiguous(memory_format=torch.channels_last):
        dz = dz.contiguous(memory_format=torch.channels_last)
    dx, dw, db = torch.ops.aten.convolution_backward(
        dz, x, w, [w.shape[0]] if need_db else None, list(stride), list(padding), list(dilation), False, [0, 0], 1,
        [bool(need_dx), bool(need_dw), bool(need_db)])
    if dx_accum is not None and dx is not None:
        dx = dx_accum + dx
    return dx, dw, db


def _backward_mfma(dz, x, w, stride, padding, dilation, need_dx, need_dw, need_db, w_dgrad, dw_out, dx_accum, dw_prezeroed):
    """Backward on the round-2 core: data gradient for any stride by residue classes (ops/conv_mfma.dgrad); weight gradient on the
    split-K transposing-read kernel (bf16, 64-multiple channels) or the library for the shapes it does not cover yet."""
    from . import conv_mfma as CM
    from . import conv_igemm as IG
    dx = dW = db = None
    if not dz.is_contiguous(memory_format=torch.channels_last):
        dz = dz.contiguous(memory_format=torch.channels_last)
    if need_db:
        db = dz.float().sum((0, 2, 3))
    if need_dw:
        if IG.wgrad_supported(x, w):
            from . import streams
            if (streams.active() and dw_out is not None and dw_out.dtype == torch.float32 and dw_out.is_contiguous()):
                with streams.side_region(streams.fork(dz, x, dw_out), 'convolution weight-gradient'):
                    IG.wgrad(dz, x, w.shape, stride, padding, dilation, dw_out=dw_out, prezeroed=dw_prezeroed)
            else:
                dW = IG.wgrad(dz, x, w.shape, stride, padding, dilation, dw_out=dw_out, prezeroed=dw_prezeroed)
        else:
            dW = torch.ops.aten.convolution_backward(dz, x, w, None, list(stride), list(padding), list(dilation), False, [0, 0], 1,
                                                     [False, True, False])[1]
    if need_dx and not CM.dgrad_supported(w):
        dx = torch.ops.aten.convolution_backward(dz, x, w, None, list(stride), list(padding), list(dilation), False, [0, 0], 1,
                                                 [True, False, False])[0]
        if dx_accum is not None:
            dx = dx_accum + dx
    elif need_dx:
        wd = w_dgrad if w_dgrad is not None else CM.flip_filters(w)
        acc = dx_accum if (dx_accum is not None and dx_accum.dtype == dz.dtype and tuple(dx_accum.shape) == tuple(x.shape)
                           and dx_accum.is_contiguous(memory_format=torch.channels_last)) else None
        dx = CM.dgrad(dz, x.shape, wd, stride, padding, dilation, accum_into=acc)
        if dx_accum is not None and acc is None:
            dx = dx_accum + dx
    return dx, dW, db
