#!/usr/bin/env python
"""SDPA forward: hand-written flash kernel (native/csrc/attention.hip) vs ATen scaled_dot_product_attention vs materialised math."""
import math
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from deeplearning4j_amd.ops import attention as A


def t(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print("| B | H | T | D | causal | hip us | TF/s | aten us | math us |")
print("|---:|---:|---:|---:|---|---:|---:|---:|---:|")
for B, H, T, D, causal in [(32, 12, 128, 64, False), (64, 12, 128, 64, False), (16, 12, 512, 64, False), (8, 16, 2048, 128, False),
                           (8, 16, 2048, 128, True), (4, 32, 4096, 128, True)]:
    q, k, v = (torch.randn(B, H, T, D, device="cuda").bfloat16() for _ in range(3))
    sc = 1 / math.sqrt(D)
    hip = t(lambda: A.sdpa(q, k, v, None, sc, causal))
    aten = t(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=sc))
    try:
        mth = t(lambda: A.sdpa(q, k, v, None, sc, causal, return_scores=True)[0], 10) if B * H * T * T <= (1 << 31) else float("nan")
    except Exception:
        mth = float("nan")
    fl = 4.0 * B * H * T * T * D * (0.5 if causal else 1.0)
    print("| %d | %d | %d | %d | %s | %.1f | %.0f | %.1f | %.1f |" % (B, H, T, D, causal, hip, fl / hip / 1e6, aten, mth))

# ---- backward: fused kernels (delta + dQ pass + dK/dV pass) vs ATen's SDPA backward vs the GEMM-based recompute
print()
print("| B | H | T | D | causal | hip bwd us | TF/s (5 GEMMs) | aten bwd us | gemm-recompute bwd us |")
print("|---:|---:|---:|---:|---|---:|---:|---:|---:|")
for B, H, T, D, causal in [(32, 12, 128, 64, False), (16, 12, 512, 64, False), (8, 16, 2048, 128, False), (8, 16, 2048, 128, True),
                           (4, 32, 4096, 128, True)]:
    q, k, v = (torch.randn(B, H, T, D, device="cuda").bfloat16().requires_grad_(True) for _ in range(3))
    sc = 1 / math.sqrt(D)
    o = A.sdpa(q, k, v, None, sc, causal)
    g = torch.randn_like(o)
    hip = t(lambda: torch.autograd.grad(o, (q, k, v), g, retain_graph=True), 20)
    oa = F.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=sc)
    aten = t(lambda: torch.autograd.grad(oa, (q, k, v), g, retain_graph=True), 20)
    fn = o.grad_fn
    qs, ks, vs, os_, lse, mk = fn.saved_tensors
    try:
        ref = (t(lambda: A.sdpa_backward_reference(qs, ks, vs, os_, lse, None, g, sc, causal), 5)
               if B * H * T * T <= (1 << 30) else float("nan"))
    except Exception:
        ref = float("nan")
    fl = 10.0 * B * H * T * T * D * (0.5 if causal else 1.0)
    print("| %d | %d | %d | %d | %s | %.1f | %.0f | %.1f | %.1f |" % (B, H, T, D, causal, hip, fl / hip / 1e6, aten, ref))
