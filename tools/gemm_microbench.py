#!/usr/bin/env python
"""bf16 GEMM y = x . w^T: the hand-written MFMA kernel (ops/gemm.py -> conv_igemm.hip as a 1x1 convolution) vs hipBLASLt through ATen."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import gemm as G  # noqa: E402
from conv_microbench import graph_time  # noqa: E402

print("| M | N | K | hip us | TF/s | hipBLASLt us | TF/s |")
print("|---:|---:|---:|---:|---:|---:|---:|")
for M, N, K in [(4096, 768, 768), (4096, 3072, 768), (4096, 768, 3072), (8192, 4096, 4096), (16384, 1024, 1024), (65536, 256, 64),
                (200704, 256, 64), (1024, 1024, 1024)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    ref = (x.float() @ w.float().t())
    got = G.gemm_nt(x, w).float()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 2e-2, err
    a = graph_time(lambda: G.gemm_nt(x, w))
    b = graph_time(lambda: torch.matmul(x, w.t()))
    fl = 2.0 * M * N * K
    print("| %d | %d | %d | %.1f | %.0f | %.1f | %.0f |" % (M, N, K, a, fl / a / 1e6, b, fl / b / 1e6))
