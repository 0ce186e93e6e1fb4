"""Fused BatchNorm forward/backward on channels-last activations viewed as [M, C].

GPU (fp32 / bf16): hand-written HIP kernels in ``native/csrc/batchnorm.hip``.
CPU or fp64 (gradient checks): the same math in plain PyTorch — this is also the numerics
oracle for the GPU tests.

Math = DL4J BatchNormalization (reference
deeplearning4j-nn/.../nn/layers/normalization/BatchNormalization.java:280-353,492-566):
biased batch variance, running statistics updated *through the gradient vector*.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from ..native import lib as L

_LN10 = math.log(10.0)


class BNContext:
    """Per-call saved state for the backward pass."""
    __slots__ = ("x", "out", "mean", "invstd", "scale", "shift", "relu", "M", "C", "res_ctx")


_WS = {}


def _workspace(C, device):
    """Per-device scratch for the two-stage column reductions (stream-ordered reuse is safe: every
    producer/consumer pair of a call runs back-to-back on the same stream)."""
    n = int(L.hip().dl4j_bn_ws_floats(L.c_int(C)))
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 1 << 20), device=device, dtype=torch.float32)
        _WS[key] = ws
    return ws


# In-kernel second reduction stage ("last block done", native/csrc/ticket.h). OFF by default: measured on MI355X
# (profiles/r2_ticket_finalize.md) every device-scope hand-off between workgroups on different XCDs is a ~2-3 us memory round trip,
# and the two-level scheme chains ~7 of them (store ack, ticket, read, store ack, ticket, read, write): +20 us per BatchNorm backward
# and +11 us per conv epilogue, against ~5 us for the dependent launch it replaces (step 11.73 ms vs 10.57 ms). DL4J_AMD_BN_TICKET=1
# enables it (useful where launches are expensive, e.g. eager mode without graph capture); tests cover both forms.
_FUSE_STAGE2 = __import__("os").environ.get("DL4J_AMD_BN_TICKET", "0") == "1"
_TICKET = {}
TICKET_COUNTERS = 65536
TICKET_GPART = 1 << 20


def ticket_ws(device):
    """(counters, group partials) of the in-kernel second reduction stage (native/csrc/ticket.h): zero-initialised, self-resetting
    counters shared by every launch on the stream. None while a graph is being captured and no buffer exists yet (a buffer from
    the capture's private pool must not outlive the graph) — callers then use the two-launch form."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    e = _TICKET.get(key)
    if e is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        e = (torch.zeros(TICKET_COUNTERS, device=device, dtype=torch.int32), torch.empty(TICKET_GPART, device=device, dtype=torch.float32))
        _TICKET[key] = e
    return e


class FinalizeRequest:
    """What the conv kernel needs to run this BatchNorm's finalize in its own epilogue (conv_igemm.forward(bn_fin=...))."""
    __slots__ = ("gamma", "beta", "conv_bias", "gmean", "gvar", "dgmean", "dgvar", "eps", "decay", "stat_mode")

    def __init__(self, gamma, beta, conv_bias, gmean, gvar, dgmean, dgvar, eps, decay, log_std):
        self.gamma, self.beta, self.conv_bias, self.gmean, self.gvar = gamma, beta, conv_bias, gmean, gvar
        self.dgmean, self.dgvar, self.eps, self.decay = dgmean, dgvar, float(eps), float(decay)
        self.stat_mode = 0 if dgmean is None else (2 if log_std else 1)

    def ok(self):
        ts = [t for t in (self.gamma, self.beta, self.conv_bias, self.gmean, self.gvar, self.dgmean, self.dgvar) if t is not None]
        return all(t.dtype == torch.float32 and t.is_contiguous() for t in ts)


class Finalized:
    """mean / invstd / scale / shift already written by the producing conv's last workgroup."""
    __slots__ = ("mean", "invstd", "scale", "shift", "K")

    def __init__(self, buf, K):
        self.mean, self.invstd, self.scale, self.shift, self.K = buf[:K], buf[K:2 * K], buf[2 * K:3 * K], buf[3 * K:], K


def _hip_ok(x):
    return L.use_hip(x) and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and x.is_contiguous()


def running_stat_grads(mean, var, gmean, gvar, decay, log_std, conv_bias=None):
    bm = mean if conv_bias is None else mean + conv_bias
    dmean = (gmean - bm) * (1.0 - decay)
    if log_std:
        vt = torch.pow(10.0, 2.0 * gvar)
        vnew = decay * vt + (1.0 - decay) * var
        dvar = torch.log(vt / vnew) / (2.0 * _LN10)
    else:
        dvar = (gvar - var) * (1.0 - decay)
    return dmean, dvar


def infer_coeffs(gamma, beta, gmean, gvar, eps, log_std, conv_bias=None):
    """scale/shift for inference from the global statistics."""
    C = gmean.numel()
    if L.use_hip(gmean) and gmean.dtype == torch.float32:
        scale = torch.empty(C, device=gmean.device, dtype=torch.float32)
        shift = torch.empty_like(scale)
        L.check(L.hip().dl4j_bn_infer_coeffs(L.ptr(gamma), L.ptr(beta), L.ptr(gmean), L.ptr(gvar), L.ptr(conv_bias),
                                             L.ptr(scale), L.ptr(shift), L.c_int(C), L.c_float(eps),
                                             L.c_int(1 if log_std else 0), L.stream()), "bn_infer_coeffs")
        return scale, shift
    var = torch.pow(10.0, 2.0 * gvar) if log_std else gvar
    g = gamma if gamma is not None else torch.ones_like(gmean)
    b = beta if beta is not None else torch.zeros_like(gmean)
    scale = g / torch.sqrt(var + eps)
    m = gmean if conv_bias is None else gmean - conv_bias
    return scale, b - m * scale


# ReLU mask of a residual block output as BITS (one byte per 8-channel vector) instead of the stored activation: both backward
# passes then read 1/16 of the bytes (the activation tensor itself stays alive as the next block's input, but is not re-read).
# DL4J_AMD_BN_BITMASK=0 restores the stored-output form.
_BITMASK = __import__("os").environ.get("DL4J_AMD_BN_BITMASK", "1") == "1"


def mask_supported(x2d):
    return _BITMASK and _hip_ok(x2d)


def apply(x2d, scale, shift, residual=None, rscale=None, rshift=None, relu=False, want_mask=False):
    """y = act(x*scale + shift [+ residual(*rscale + rshift)]). With ``want_mask`` (HIP path, relu) returns (y, mask_bits)."""
    M, C = x2d.shape
    if _hip_ok(x2d):
        y = torch.empty_like(x2d)
        if want_mask and relu and _BITMASK:
            v = int(L.hip().dl4j_bn_mask_vec(L.c_int(C), L.c_int(L.dtype_code(x2d))))
            bits = torch.empty((M * C) // v, device=x2d.device, dtype=torch.uint8)
            L.check(L.hip().dl4j_bn_apply_mask(L.ptr(x2d), L.ptr(scale), L.ptr(shift), L.ptr(residual), L.ptr(rscale),
                                               L.ptr(rshift), L.ptr(y), L.ptr(bits), L.c_int64(M), L.c_int(C), L.c_int(1),
                                               L.c_int(L.dtype_code(x2d)), L.stream()), "bn_apply_mask")
            return y, bits
        L.check(L.hip().dl4j_bn_apply(L.ptr(x2d), L.ptr(scale), L.ptr(shift), L.ptr(residual), L.ptr(rscale),
                                      L.ptr(rshift), L.ptr(y), L.c_int64(M), L.c_int(C), L.c_int(1 if relu else 0),
                                      L.c_int(L.dtype_code(x2d)), L.stream()), "bn_apply")
        return (y, None) if want_mask else y
    cd = torch.float32 if x2d.dtype in (torch.bfloat16, torch.float16) else x2d.dtype
    y = x2d.to(cd) * scale.to(cd) + shift.to(cd)
    if residual is not None:
        r = residual.to(cd)
        if rscale is not None:
            r = r * rscale.to(cd) + rshift.to(cd)
        y = y + r
    if relu:
        y = torch.relu(y)
    return (y.to(x2d.dtype), None) if want_mask else y.to(x2d.dtype)


def batch_stats(x2d, gamma, beta, eps, decay=0.9, gmean=None, gvar=None, log_std=False, conv_bias=None,
                dgmean_out=None, dgvar_out=None, sums=None):
    """Batch statistics + affine coefficients. Returns (mean, invstd, scale, shift).
    ``sums`` ([2C] fp32: per-channel sum and sum of squares), when given, comes from the producing
    conv's epilogue and the statistics pass over x is skipped."""
    M, C = x2d.shape
    stat_mode = 0 if dgmean_out is None else (2 if log_std else 1)
    if _hip_ok(x2d):
        dev = x2d.device
        buf = torch.empty(6 * C, device=dev, dtype=torch.float32)
        mean, invstd, scale, shift = buf[2 * C:3 * C], buf[3 * C:4 * C], buf[4 * C:5 * C], buf[5 * C:]
        h = L.hip()
        if isinstance(sums, Finalized):
            assert sums.K == C
            return sums.mean, sums.invstd, sums.scale, sums.shift
        if sums is not None and hasattr(sums, "partial"):
            # conv-epilogue partials: reduce + finalize in one launch
            assert sums.K == C and sums.partial.numel() >= sums.gy * 2 * C
            piv = getattr(sums, "pivot", None)
            if piv is not None:
                # rows are sums of (x - pivot): the pivot is recomputed in the kernel from the SAME running mean / folded bias
                assert gmean is not None and piv[0].data_ptr() == gmean.data_ptr() and \
                    (piv[1] is None) == (conv_bias is None) and (piv[1] is None or piv[1].data_ptr() == conv_bias.data_ptr())
                L.check(h.dl4j_bn_finalize_partials_shifted(
                    L.ptr(sums.partial), L.c_int(sums.gy), L.ptr(gamma), L.ptr(beta), L.ptr(conv_bias), L.ptr(gmean), L.ptr(gvar), L.ptr(mean),
                    L.ptr(invstd), L.ptr(scale), L.ptr(shift), L.ptr(dgmean_out), L.ptr(dgvar_out), L.c_int(C), L.c_int64(M), L.c_float(eps),
                    L.c_float(decay), L.c_int(stat_mode), L.stream()), "bn_finalize_partials_shifted")
                return mean, invstd, scale, shift
            L.check(h.dl4j_bn_finalize_partials(L.ptr(sums.partial), L.c_int(sums.gy), L.ptr(gamma), L.ptr(beta), L.ptr(conv_bias),
                                                L.ptr(gmean), L.ptr(gvar), L.ptr(mean), L.ptr(invstd), L.ptr(scale), L.ptr(shift),
                                                L.ptr(dgmean_out), L.ptr(dgvar_out), L.c_int(C), L.c_int64(M), L.c_float(eps),
                                                L.c_float(decay), L.c_int(stat_mode), L.stream()), "bn_finalize_partials")
            return mean, invstd, scale, shift
        if sums is None:
            sums = buf[:2 * C]
            ws = _workspace(C, dev)
            L.check(h.dl4j_bn_stats(L.ptr(x2d), L.ptr(ws), L.ptr(sums), L.c_int64(M), L.c_int(C),
                                    L.c_int(L.dtype_code(x2d)), L.stream()), "bn_stats")
        L.check(h.dl4j_bn_finalize(L.ptr(sums), L.ptr(gamma), L.ptr(beta), L.ptr(conv_bias), L.ptr(gmean),
                                   L.ptr(gvar), L.ptr(mean), L.ptr(invstd), L.ptr(scale), L.ptr(shift),
                                   L.ptr(dgmean_out), L.ptr(dgvar_out), L.c_int(C), L.c_int64(M), L.c_float(eps),
                                   L.c_float(decay), L.c_int(stat_mode), L.stream()), "bn_finalize")
        return mean, invstd, scale, shift
    cd = torch.float32 if x2d.dtype in (torch.bfloat16, torch.float16) else x2d.dtype
    xf = x2d.to(cd)
    mean = xf.mean(0)
    var = xf.var(0, unbiased=False)
    invstd = torch.rsqrt(var + eps)
    g = gamma.to(cd) if gamma is not None else torch.ones_like(mean)
    b = beta.to(cd) if beta is not None else torch.zeros_like(mean)
    scale = g * invstd
    shift = b - mean * scale
    if stat_mode:
        dm, dv = running_stat_grads(mean, var, gmean.to(cd), gvar.to(cd), decay, log_std,
                                    None if conv_bias is None else conv_bias.to(cd))
        dgmean_out.copy_(dm)
        dgvar_out.copy_(dv)
    return mean, invstd, scale, shift


def backward(x2d, dy2d, out2d, mean, invstd, gamma, dgamma_out, dbeta_out, want_dz=False, mask_coeffs=None, mask_bits=None,
             pre_partials=None):
    """BatchNorm backward. ``out2d`` (optional) is the fused ReLU output used as mask; alternatively ``mask_coeffs`` =
    (scale, shift) lets the kernels recompute the mask as [x*scale + shift > 0] (valid when no residual was added), which
    saves re-reading the activation tensor in both backward passes.

    Returns (dx, dz) where dz = dy*[out>0] (only materialised when want_dz).
    dgamma/dbeta are written into the given fp32 gradient views (may be None for lockGammaBeta).
    """
    M, C = x2d.shape
    if mask_bits is not None and not (_hip_ok(x2d) and dy2d.is_contiguous()):
        raise ValueError("mask_bits is a HIP-path artefact: contiguous device tensors required")
    if _hip_ok(x2d) and dy2d.is_contiguous() and (out2d is None or out2d.is_contiguous()):
        h = L.hip()
        sums = torch.empty(2 * C, device=x2d.device, dtype=torch.float32)
        dt = L.c_int(L.dtype_code(x2d))
        ws = _workspace(C, x2d.device)
        direct = (dbeta_out is None or (dbeta_out.is_contiguous() and dbeta_out.dtype == torch.float32)) and \
                 (dgamma_out is None or (dgamma_out.is_contiguous() and dgamma_out.dtype == torch.float32))
        ms, mt = mask_coeffs if (mask_coeffs is not None and out2d is None) else (None, None)
        tk = ticket_ws(x2d.device) if (_FUSE_STAGE2 and 128 * C <= TICKET_GPART) else None
        if pre_partials is not None:
            # (partial [gy, 2, C], gy): per-tile sum dz / sum dz*xhat rows emitted by the epilogue of the data-gradient launch
            # that produced dy (ops/conv_igemm.BnBackwardSink): only the tiny second stage runs here
            part, gy = pre_partials[0], pre_partials[1]
            assert part.numel() >= gy * 2 * C
            done = pre_partials[2] if len(pre_partials) > 2 else None
            if done is not None and direct:
                sums = done                      # second stage already ran inside the weight-gradient launch (and wrote dbeta / dgamma)
            else:
                L.check(h.dl4j_bn_sum_partials(L.ptr(part), L.c_int(gy), L.c_int(C), L.ptr(sums), L.ptr(dbeta_out if direct else None),
                                               L.ptr(dgamma_out if direct else None), L.stream()), "bn_sum_partials")
        elif mask_bits is not None:
            L.check(h.dl4j_bn_bwd_reduce_mask(L.ptr(x2d), L.ptr(dy2d), L.ptr(mask_bits), L.ptr(mean), L.ptr(invstd), L.ptr(ws),
                                              L.ptr(sums), L.ptr(dbeta_out if direct else None), L.ptr(dgamma_out if direct else None),
                                              L.c_int64(M), L.c_int(C), dt, L.stream()), "bn_bwd_reduce_mask")
        elif tk is not None:
            L.check(h.dl4j_bn_bwd_reduce_fused(L.ptr(x2d), L.ptr(dy2d), L.ptr(out2d), L.ptr(mean), L.ptr(invstd), L.ptr(ms), L.ptr(mt),
                                               L.ptr(ws), L.ptr(sums), L.ptr(dbeta_out if direct else None),
                                               L.ptr(dgamma_out if direct else None), L.c_int64(M), L.c_int(C), dt, L.ptr(tk[1]),
                                               L.ptr(tk[0]), L.stream()), "bn_bwd_reduce_fused")
        else:
            L.check(h.dl4j_bn_bwd_reduce(L.ptr(x2d), L.ptr(dy2d), L.ptr(out2d), L.ptr(mean), L.ptr(invstd), L.ptr(ms), L.ptr(mt),
                                         L.ptr(ws), L.ptr(sums), L.ptr(dbeta_out if direct else None),
                                         L.ptr(dgamma_out if direct else None), L.c_int64(M), L.c_int(C), dt, L.stream()),
                    "bn_bwd_reduce")
        dx = torch.empty_like(x2d)
        dz = torch.empty_like(x2d) if want_dz else None
        if mask_bits is not None:
            L.check(h.dl4j_bn_bwd_dx_mask(L.ptr(x2d), L.ptr(dy2d), L.ptr(mask_bits), L.ptr(mean), L.ptr(invstd), L.ptr(gamma),
                                          L.ptr(sums), L.ptr(dx), L.ptr(dz), L.c_int64(M), L.c_int(C), dt, L.stream()),
                    "bn_bwd_dx_mask")
        else:
            L.check(h.dl4j_bn_bwd_dx(L.ptr(x2d), L.ptr(dy2d), L.ptr(out2d), L.ptr(mean), L.ptr(invstd), L.ptr(gamma),
                                     L.ptr(sums), L.ptr(ms), L.ptr(mt), L.ptr(dx), L.ptr(dz), L.c_int64(M), L.c_int(C), dt,
                                     L.stream()),
                    "bn_bwd_dx")
        if not direct:
            if dbeta_out is not None:
                dbeta_out.copy_(sums[:C])
            if dgamma_out is not None:
                dgamma_out.copy_(sums[C:])
        return dx, dz
    cd = torch.float32 if x2d.dtype in (torch.bfloat16, torch.float16) else x2d.dtype
    xf, dz = x2d.to(cd), dy2d.to(cd)
    if out2d is not None:
        dz = dz * (out2d > 0).to(cd)
    elif mask_coeffs is not None:
        dz = dz * ((xf * mask_coeffs[0].to(cd) + mask_coeffs[1].to(cd)) > 0).to(cd)
    xhat = (xf - mean.to(cd)) * invstd.to(cd)
    dbeta = dz.sum(0)
    dgamma = (dz * xhat).sum(0)
    g = gamma.to(cd) if gamma is not None else torch.ones_like(dbeta)
    dx = (g * invstd.to(cd)) * (dz - dbeta / M - xhat * dgamma / M)
    if dbeta_out is not None:
        dbeta_out.copy_(dbeta)
    if dgamma_out is not None:
        dgamma_out.copy_(dgamma)
    return dx.to(x2d.dtype), (dz.to(x2d.dtype) if want_dz else None)
