#!/usr/bin/env python
"""Probe one conv shape: round-2 conv kernel with / without stats vs gemm_nt on the same [M, K] x [N, K] problem vs round-1 kernel."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import conv_igemm as IG, conv_mfma as CM, gemm as G  # noqa: E402
from conv_microbench import graph_time  # noqa: E402

B = 256
shapes = [(64, 28, 256, 1, 1), (256, 28, 64, 1, 1), (128, 14, 128, 3, 1), (1024, 7, 256, 1, 1)]
quick = len(sys.argv) > 1
for C, H, K, R, s in shapes[:1] if quick else shapes:
    p = R // 2
    x = torch.randn(B, C, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, R, R, device="cuda") / (C * R * R) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    st, pd, dl = (s, s), (p, p), (1, 1)
    if quick:
        for _ in range(3):
            CM.forward(x, w, None, st, pd, dl, want_stats=True)
            CM.forward(x, w, None, st, pd, dl, want_stats=False)
            IG.forward(x, w, None, st, pd, dl, want_stats=True)
            if R == 1:
                G.gemm_nt(x.permute(0, 2, 3, 1).reshape(-1, C), w.reshape(K, C), tile=2)
        torch.cuda.synchronize()
        continue
    r = {}
    r["r1+stats"] = graph_time(lambda: IG.forward(x, w, None, st, pd, dl, want_stats=True))
    r["r1"] = graph_time(lambda: IG.forward(x, w, None, st, pd, dl, want_stats=False))
    for t in (1, 2):
        CM._FORCE_TILE = t; CM._DESC.clear()
        r["r2 t%d+stats" % t] = graph_time(lambda: CM.forward(x, w, None, st, pd, dl, want_stats=True))
        CM._DESC.clear()
        r["r2 t%d" % t] = graph_time(lambda: CM.forward(x, w, None, st, pd, dl, want_stats=False))
    CM._FORCE_TILE = None; CM._DESC.clear()
    if R == 1:
        x2, w2 = x.permute(0, 2, 3, 1).reshape(-1, C), w.reshape(K, C)
        for t in (1, 2):
            r["gemm t%d" % t] = graph_time(lambda: G.gemm_nt(x2, w2, tile=t))
        r["hipblaslt"] = graph_time(lambda: torch.matmul(x2, w2.t()))
    print((C, H, K, R, s), "  ".join("%s=%.1f" % kv for kv in r.items()), flush=True)
