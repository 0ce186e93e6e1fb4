#!/usr/bin/env python
"""For every ResNet-50 conv shape: time forward and dgrad with each implicit-GEMM tile (128x128, 128x64, 64x64) to tune pick_tile."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import conv_igemm as IG  # noqa: E402
from conv_microbench import SHAPES, N, graph_time  # noqa: E402

print("| C | HxW | K | RxR/s | default | fwd t0 | t1 | t2 | dgrad default | t0 | t1 | t2 |")
print("|---:|---:|---:|---|---:|---:|---:|---:|---:|---:|---:|---:|")
tot = {"d": 0.0, "best": 0.0}
for C, H, K, R, s, cnt in SHAPES:
    p = R // 2
    x = torch.randn(N, C, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, R, R, device="cuda") / (C * R * R) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    wd = torch.flip(w, dims=[2, 3]).permute(1, 2, 3, 0).contiguous()
    y, _ = IG.forward(x, w, None, (s, s), (p, p), (1, 1), want_stats=True)
    dz = torch.randn_like(y)
    row_f, row_d = [], []
    for t in (None, 0, 1, 2):
        IG._FORCE_TILE = t
        row_f.append(graph_time(lambda: IG.forward(x, w, None, (s, s), (p, p), (1, 1), want_stats=True)))
        row_d.append(graph_time(lambda: IG.dgrad(dz, x.shape, wd, (s, s), (p, p), (1, 1))) if IG.dgrad_supported(w, (s, s), (p, p)) else float("nan"))
    IG._FORCE_TILE = None
    print("| %d | %d | %d | %dx%d/%d | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f |" % ((C, H, K, R, R, s) + tuple(row_f) + tuple(row_d)))
    tot["d"] += (row_f[0] + (row_d[0] if row_d[0] == row_d[0] else 0)) * cnt
    tot["best"] += (min(row_f) + (min(v for v in row_d if v == v) if row_d[0] == row_d[0] else 0)) * cnt
print("\nfwd+dgrad network total: default %.2f ms, best-per-layer %.2f ms" % (tot["d"] / 1e3, tot["best"] / 1e3))
