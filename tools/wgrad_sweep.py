#!/usr/bin/env python
"""Split-K sweep of the implicit-GEMM wgrad kernel over the ResNet-50 conv shapes (target_blocks = 128 .. 4096)."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import conv_igemm as IG  # noqa: E402
from conv_microbench import SHAPES, N, graph_time  # noqa: E402

TB = (0, 128, 256, 384, 512, 1024)      # 0 = the kernel's own choice
print("| C | HxW | K | RxR/s | " + " | ".join("tb%d" % t for t in TB) + " |")
print("|---:|---:|---:|---|" + "---:|" * len(TB))
tot = [0.0] * len(TB)
best = 0.0
for C, H, K, R, s, cnt in SHAPES:
    p = R // 2
    x = torch.randn(N, C, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(K, C, R, R, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    y, _ = IG.forward(x, w, None, (s, s), (p, p), (1, 1), want_stats=False)
    dz = torch.randn_like(y)
    dw = torch.zeros(K, C, R, R, device="cuda")
    row = [graph_time(lambda: IG.wgrad(dz, x, w.shape, (s, s), (p, p), (1, 1), dw_out=dw, target_blocks=tb)) for tb in TB]
    for k, v in enumerate(row):
        tot[k] += v * cnt
    best += min(row) * cnt
    print("| %d | %d | %d | %dx%d/%d | " % (C, H, K, R, R, s) + " | ".join("%.1f" % v for v in row) + " |")
print("\nnetwork wgrad totals (ms): " + ", ".join("tb%d %.2f" % (t, v / 1e3) for t, v in zip(TB, tot)) + "; best per layer %.2f" % (best / 1e3))
