#!/usr/bin/env python
"""A few launches of the implicit-GEMM conv kernels on representative ResNet-50 shapes, for rocprofv3 --pmc runs."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import conv_igemm as IG  # noqa: E402

N = 256
for C, H, K, R, s in [(256, 7, 256, 3, 1), (128, 14, 128, 3, 1), (64, 28, 256, 1, 1), (1024, 7, 256, 1, 1)]:
    p = R // 2
    x = torch.randn(N, C, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(K, C, R, R, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        y, _ = IG.forward(x, w, None, (s, s), (p, p), (1, 1), want_stats=True)
    dz = torch.randn_like(y)
    dw = torch.zeros(K, C, R, R, device="cuda")
    for _ in range(3):
        IG.wgrad(dz, x, w.shape, (s, s), (p, p), (1, 1), dw_out=dw)
torch.cuda.synchronize()
