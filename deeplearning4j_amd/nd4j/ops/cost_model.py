"""Measured cost model of the platform helpers: the second half of ``isUsable``.

A platform helper's predicate says that the hand-written kernel CAN run the arguments; this module says whether it SHOULD. Every
rule is a measurement on MI355X (tools/op_microbench.py, profiles/r2_op_microbench.md / r3_op_microbench.md): where the generic
ATen composition was the faster route the helper is declined, the reason is counted in the helper statistics
(``OpExecutioner.helperStatistics()['fail']``, key ``(op, 'cost model: ...')``) and the generic implementation runs.

Reference: PlatformHelper::isUsable of the cuDNN / oneDNN helpers, which reject shapes their library is known to be slow on
(ops/declarable/platform/cudnn/*.cu ``PLATFORM_CHECK``, e.g. conv2d.cu declining unsupported padding modes) - here the rejection
criterion is measured speed rather than library capability.

``DL4J_AMD_HELPER_COST_MODEL=0`` (or ``cost_model.ENABLED = False``) switches the model off: every usable helper runs, which is how
the kernels' own tests and the "forced helper" column of the micro-benchmark reach them. Calls whose inputs are autograd-tracked are
never declined: a differentiable helper is the training path of the SameDiff graphs (cached weight replicas, fused epilogues), which
the eager single-call measurement does not describe.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

ENABLED = os.environ.get("DL4J_AMD_HELPER_COST_MODEL", "1") != "0"
SHORT_ROW = 128          # elements of the innermost contiguous run below which the index-map kernels lose to ATen's copy kernels


def _inner(x):
    return int(x.shape[-1]) if isinstance(x, torch.Tensor) and x.dim() else 0


def _pad(ins, i, t, b, d):
    if _inner(ins[0]) < SHORT_ROW:
        return "innermost row of %d elements (< %d): ATen's pad measured 2.1x faster at [64,64,56,56]" % (_inner(ins[0]), SHORT_ROW)
    return None


def _reverse(ins, i, t, b, d):
    if _inner(ins[0]) < SHORT_ROW:
        return "innermost row of %d elements (< %d): ATen's flip measured 1.7x faster at [64,64,56,56]" % (_inner(ins[0]), SHORT_ROW)
    return None


def _matmul(ins, i, t, b, d):
    """Plain GEMMs (no fused epilogue): the vendor library is ahead on most shapes in every operand layout
    (profiles/r3_gemm_layouts.md: in-tree NT / NN / TN / TT at 0.55-1.08x hipBLASLt; >= 0.92x only around 4096 x 768 x 768), so an
    untracked 2-D matmul goes to it unless the product is small enough that the two are level."""
    if len(ins) >= 2 and ins[0].dim() == 2 and ins[1].dim() == 2:
        m, k = ins[0].shape[-2], ins[0].shape[-1]
        if i and int(i[0]):
            m, k = k, m
        n = ins[1].shape[-2] if (i and len(i) > 1 and int(i[1])) else ins[1].shape[-1]
        if 2.0 * m * n * k > 6e9 or min(m, n, k) < 256:
            return "plain 2-D GEMM %dx%dx%d: hipBLASLt measured 1.2-1.8x faster than the in-tree kernel at this size (in-tree wins / ties only near 4096x768x768)" % (m, n, k)
    return None


def _xw_plus_b(ins, i, t, b, d):
    if not (i and i[0]):
        return "weights [K, N]: the NT kernel transposes them per call; the library GEMM + bias measured 1.4x faster at 8192x1024x1024 bf16"
    return None


def _biasadd(ins, i, t, b, d):
    nchw = bool(b[0]) if b else (bool(i[0]) if i else False)
    if nchw and ins[0].dim() > 2:
        return "NCHW layout: ATen's broadcast add measured 1.07x faster at [64,256,56,56]"
    return None


_ALWAYS = {
    "tile": "ATen's repeat measured 1.13x faster at 2x2 of [4096,1024]",
    "concat": "ATen's cat measured 1.32x faster at 2 x [8192,2048] along axis 1",
    "space_to_depth": "ATen's reshape-permute measured 3x faster at [32,112,112,3] block 2",
    "depth_to_space": "mirror of space_to_depth (same index map), measured slower than ATen's reshape-permute",
    "im2col": "ATen's unfold measured 1.06x faster at 3x3 [64,64,56,56]",
    "depthwise_conv2d": "MIOpen's grouped convolution measured 1.06x faster at 3x3 [64,64,56,56]",
}
_RULES = {"pad": _pad, "mirror_pad": _pad, "reverse": _reverse, "matmul": _matmul, "xw_plus_b": _xw_plus_b, "biasadd": _biasadd}


def prefer_generic(name, ins, i, t, b, d) -> Optional[str]:
    """None: run the helper. Otherwise the measured reason why the generic implementation is the faster route for these arguments."""
    if not ENABLED:
        return None
    if name in _ALWAYS:
        return _ALWAYS[name]
    rule = _RULES.get(name)
    if rule is None:
        return None
    try:
        return rule(ins, i or [], t or [], b or [], d or [])
    except Exception:      # noqa: BLE001 - a cost estimate must never break an op call
        return None


def rules():
    """{op: description} of everything the model can decline (documentation / tests)."""
    out = dict(_ALWAYS)
    out.update({k: (v.__doc__ or "shape-dependent rule (see source)") for k, v in _RULES.items()})
    return out
