#!/usr/bin/env python
"""Isolated bandwidth check of the BatchNorm streaming kernels on the ResNet-50 layer shapes (run on the GPU box):
prints per-shape time and achieved GB/s for apply / bwd-reduce+dx, so in-graph per-call times can be sanity-checked."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import batchnorm as BN  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3          # us


def main():
    dev = torch.device("cuda")
    N = 256
    shapes = [(64, 112), (64, 28), (256, 28), (128, 14), (512, 14), (256, 7), (1024, 7), (512, 4), (2048, 4)]
    print("| C | HxW | MB/tensor | apply us | GB/s | bwd(reduce+dx) us | GB/s | bwd+res(dz) us | GB/s |")
    print("|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for C, hw in shapes:
        M = N * hw * hw
        x = torch.randn(M, C, device=dev).bfloat16()
        dy = torch.randn(M, C, device=dev).bfloat16()
        res = torch.randn(M, C, device=dev).bfloat16()
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        mean, invstd, scale, shift = BN.batch_stats(x, gamma, beta, 1e-5)
        y = BN.apply(x, scale, shift, relu=True)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        mb = M * C * 2 / 1e6
        ta = timeit(lambda: BN.apply(x, scale, shift, relu=True))
        tb = timeit(lambda: BN.backward(x, dy, y, mean, invstd, gamma, dg, db))
        tr = timeit(lambda: BN.backward(x, dy, y, mean, invstd, gamma, dg, db, want_dz=True))
        print("| %d | %d | %.1f | %.1f | %.0f | %.1f | %.0f | %.1f | %.0f |" % (C, hw, mb, ta, 2 * mb / ta * 1e3, tb, 7 * mb / tb * 1e3, tr,
                                                                             8 * mb / tr * 1e3))


if __name__ == "__main__":
    main()
