#!/usr/bin/env python
"""bf16 NT GEMM: every tile of ops/gemm.py (0: 128x128, 1: 128x64, 2: 64x64, 3: 64x128, 4: 256x256 ping-pong) per shape, against pick_tile's
choice and hipBLASLt - the table pick_tile is tuned from."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import gemm as G  # noqa: E402
from conv_microbench import graph_time  # noqa: E402

SHAPES = [(4096, 768, 768), (3200, 2048, 512), (4096, 2304, 768), (4096, 3072, 768), (4096, 768, 3072), (8192, 4096, 4096), (16384, 1024, 1024), (65536, 256, 64),
          (200704, 256, 64), (1024, 1024, 1024), (2048, 2048, 2048), (4096, 4096, 1024), (8192, 3072, 768), (8192, 768, 3072), (512, 512, 512)]
print("| M | N | K | pick | t0 128x128 | t1 128x64 | t2 64x64 | t3 64x128 | t4 256x256pp | t5 =t0 s3 | t6 =t1 s3 | t7 =t2 s3 | t8 =t3 s3 | t9 =t2 s4 | t10 256x256 deep | t11 128x96 | t12 96x128 | hipBLASLt | pick/lib | best/lib |")
print("|---:|---:|---:|---:|" + "---:|" * 16)
for M, N, K in SHAPES:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    row = []
    for t in range(13):
        try:
            row.append(graph_time(lambda: G.gemm_nt(x, w, tile=t)))
        except Exception:
            row.append(float("nan"))
    lib = graph_time(lambda: torch.matmul(x, w.t()))
    best = min(v for v in row if v == v)
    pk = G.pick_tile(M, N, K, torch.bfloat16)
    print("| %d | %d | %d | %d | " % (M, N, K, pk) + " | ".join("%.1f" % v for v in row) + " | %.1f | %.2f | %.2f |" % (lib, lib / row[pk], lib / best), flush=True)
