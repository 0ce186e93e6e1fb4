#!/usr/bin/env python
"""Per-layer timing of the zoo ResNet-50 conv shapes: round-1 igemm kernels vs the round-2 LDS-DMA core (forward, dgrad), every tile."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import conv_igemm as IG, conv_mfma as CM  # noqa: E402
from conv_microbench import graph_time  # noqa: E402

B = 256
# (C, H, K, R, stride)  distinct conv shapes of the DL4J zoo ResNet-50 (28/14/7/4 feature maps)
SHAPES = [(64, 56, 64, 1, 2), (64, 28, 64, 3, 1), (64, 28, 256, 1, 1), (64, 56, 256, 1, 2), (256, 28, 64, 1, 1),
          (256, 28, 128, 1, 2), (128, 14, 128, 3, 1), (128, 14, 512, 1, 1), (256, 28, 512, 1, 2), (512, 14, 128, 1, 1),
          (512, 14, 256, 1, 2), (256, 7, 256, 3, 1), (256, 7, 1024, 1, 1), (512, 14, 1024, 1, 2), (1024, 7, 256, 1, 1),
          (1024, 7, 512, 1, 2), (512, 4, 512, 3, 1), (512, 4, 2048, 1, 1), (1024, 7, 2048, 1, 2), (2048, 4, 512, 1, 1)]
print("| C | H | K | R | s | fwd r1 us | fwd r2 auto | r2 t0 | r2 t1 | r2 t2 | r2 t3 | dgrad r1 | dgrad r2 |")
print("|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
tot = [0.0, 0.0, 0.0, 0.0]
for C, H, K, R, s in SHAPES:
    p = R // 2
    x = torch.randn(B, C, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, R, R, device="cuda") / (C * R * R) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    st, pd, dl = (s, s), (p, p), (1, 1)
    a = graph_time(lambda: IG.forward(x, w, None, st, pd, dl, want_stats=True))
    b = graph_time(lambda: CM.forward(x, w, None, st, pd, dl, want_stats=True))
    per = []
    for t in range(4):
        CM._FORCE_TILE = t
        CM._DESC.clear()
        per.append(graph_time(lambda: CM.forward(x, w, None, st, pd, dl, want_stats=True)))
    CM._FORCE_TILE = None
    CM._DESC.clear()
    y, _ = CM.forward(x, w, None, st, pd, dl)
    dz = torch.randn_like(y)
    wd = CM.flip_filters(w)
    if IG.dgrad_supported(w, st, pd):
        c = graph_time(lambda: IG.dgrad(dz, x.shape, wd, st, pd, dl))
    else:
        c = float("nan")
    d = graph_time(lambda: CM.dgrad(dz, x.shape, wd, st, pd, dl))
    tot[0] += a; tot[1] += b; tot[2] += c; tot[3] += d
    print("| %d | %d | %d | %d | %d | %.1f | %.1f | %s | %.1f | %.1f |" % (C, H, K, R, s, a, b, " | ".join("%.1f" % v for v in per), c, d))
print("totals: fwd r1 %.0f us, fwd r2 %.0f us, dgrad r1 %.0f us, dgrad r2 %.0f us" % tuple(tot))
