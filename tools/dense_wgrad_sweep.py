#!/usr/bin/env python
"""dW = x^T . dy for dense / transformer shapes: split-K sweep of the wgrad kernel (target_blocks) vs hipBLASLt and vs
explicit transposes + the NT GEMM kernel."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import conv_igemm as IG  # noqa: E402
from deeplearning4j_amd.ops import gemm as G  # noqa: E402
from conv_microbench import graph_time  # noqa: E402

TB = (0, 36, 72, 144, 288, 432, 576, 1024)
print("| M | K | N | " + " | ".join("tb%d" % t for t in TB) + " | hipBLASLt | transposes+gemm_nt |")
print("|---:|---:|---:|" + "---:|" * (len(TB) + 2))
for M, K, N in [(4096, 768, 768), (4096, 768, 3072), (4096, 3072, 768), (16384, 1024, 1024), (8192, 4096, 4096)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    dy = torch.randn(M, N, device="cuda").bfloat16()
    out = torch.zeros(K, N, device="cuda")
    row = []
    for tb in TB:
        row.append(graph_time(lambda: IG.wgrad(x.view(M, K, 1, 1), dy.view(M, N, 1, 1), (K, N, 1, 1), (1, 1), (0, 0), (1, 1),
                                               dw_out=out.view(K, N, 1, 1), target_blocks=tb, prezeroed=True)))
    lib = graph_time(lambda: torch.matmul(x.t(), dy))
    alt = graph_time(lambda: G.gemm_nt(x.t().contiguous(), dy.t().contiguous(), out_dtype=torch.float32))
    print("| %d | %d | %d | " % (M, K, N) + " | ".join("%.1f" % v for v in row) + " | %.1f | %.1f |" % (lib, alt))
