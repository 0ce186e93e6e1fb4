#!/usr/bin/env python
"""Stem conv (3->64, 7x7/2, batch 256, 224x224): hand-written kernels (conv_stem.hip) vs MIOpen through ATen."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import conv_stem as CS  # noqa: E402
from conv_microbench import graph_time  # noqa: E402

N = 256
x = torch.randn(N, 3, 224, 224, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 3, 7, 7, device="cuda") * 0.1).bfloat16().contiguous(memory_format=torch.channels_last)
y = F.conv2d(x, w, None, 2, 3)
dz = torch.randn_like(y)
dw = torch.zeros(64 * 147, device="cuda")
print("| op | rows/block | us |")
print("|---|---:|---:|")
print("| MIOpen fwd | - | %.1f |" % graph_time(lambda: F.conv2d(x, w, None, 2, 3)))
print("| MIOpen wgrad | - | %.1f |" % graph_time(lambda: torch.ops.aten.convolution_backward(
    dz, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])))
for rpb in (8, 14, 16, 28, 56):
    CS.FWD_ROWS_PER_BLOCK = rpb
    print("| hip fwd + BN stats | %d | %.1f |" % (rpb, graph_time(lambda: CS.forward(x, w, None, (2, 2), (3, 3), (1, 1), True))))
CS.FWD_ROWS_PER_BLOCK = 4
print("| hip fwd (no stats) | 4 | %.1f |" % graph_time(lambda: CS.forward(x, w, None, (2, 2), (3, 3), (1, 1), False)))
for rpb in (28, 38, 56):
    CS.WGRAD_ROWS_PER_BLOCK = rpb
    print("| hip wgrad | %d | %.1f |" % (rpb, graph_time(lambda: CS.wgrad(dz, x, (2, 2), (3, 3), (1, 1), dw_out=dw))))
