#!/usr/bin/env python
"""bf16 GEMM in every operand layout (ops/gemm.py gemm on views, native/csrc/gemm_trans.h) vs hipBLASLt through ATen on the same views,
plus the dense weight-gradient form dW = dz^T . x (TN, split-K sweep) vs the round-1 wgrad kernel and the library."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import conv_igemm as IG  # noqa: E402
from deeplearning4j_amd.ops import gemm as G  # noqa: E402
from conv_microbench import graph_time  # noqa: E402

SHAPES = [(4096, 768, 768), (4096, 3072, 768), (4096, 768, 3072), (8192, 4096, 4096), (16384, 1024, 1024), (65536, 256, 64),
          (200704, 256, 64), (1024, 1024, 1024)]


def main():
    print("## C[M, N] = op(A) . op(B), bf16, us per call (in-tree / hipBLASLt)\n")
    print("| M | N | K | NT | lib | x | NN | lib | x | TN | lib | x | TT | lib | x |")
    print("|---:|---:|---:|" + "---:|" * 12)
    for M, N, K in SHAPES:
        row = []
        for ta, tb in ((False, False), (False, True), (True, False), (True, True)):
            a = torch.randn(K, M, device="cuda").bfloat16().t() if ta else torch.randn(M, K, device="cuda").bfloat16()
            b = torch.randn(K, N, device="cuda").bfloat16().t() if tb else torch.randn(N, K, device="cuda").bfloat16()
            ref = a.float() @ b.float().t()
            got = G.gemm(a, b, splits=1).float()
            assert float((got - ref).abs().max() / ref.abs().max()) < 2e-2
            if not ta and not tb:
                mine = graph_time(lambda: G.gemm_nt(a, b))
            else:
                mine = graph_time(lambda: G.gemm(a, b, splits=1))
            lib = graph_time(lambda: torch.matmul(a, b.t()))
            row += ["%.1f" % mine, "%.1f" % lib, "%.2f" % (lib / mine)]
        print("| %d | %d | %d | " % (M, N, K) + " | ".join(row) + " |")
    print("\n## dense weight gradient dW[nOut, nIn] = dz^T . x (fp32 output, pre-zeroed), us per call\n")
    print("| tokens | nIn | nOut | TN split-K heuristic (splits) | best of splits 1/2/4/8/16/32 | round-1 wgrad kernel | hipBLASLt (bf16 out) | heuristic / lib |")
    print("|---:|---:|---:|---:|---:|---:|---:|---:|")
    for M, K, N in [(4096, 768, 768), (4096, 768, 3072), (4096, 3072, 768), (16384, 1024, 1024), (8192, 4096, 4096)]:
        x = torch.randn(M, K, device="cuda").bfloat16()
        dz = torch.randn(M, N, device="cuda").bfloat16()
        out = torch.zeros(N, K, device="cuda")
        t = G.pick_tile(N, K, M, torch.bfloat16)
        t = 0 if t in (4, 10) else t                     # as ops.gemm.gemm maps the ping-pong tiles for the layout-aware kernels
        sp = G.pick_splits(N, K, M, t, torch.bfloat16)
        heur = graph_time(lambda: G.gemm(dz.t(), x.t(), out=out, splits=sp, accum=True))
        best = min((graph_time(lambda: G.gemm(dz.t(), x.t(), out=out, splits=s, accum=True)), s) for s in (1, 2, 4, 8, 16, 32))
        out2 = torch.zeros(K, N, device="cuda")
        old = graph_time(lambda: IG.wgrad(x.view(M, K, 1, 1), dz.view(M, N, 1, 1), (K, N, 1, 1), (1, 1), (0, 0), (1, 1),
                                          dw_out=out2.view(K, N, 1, 1), prezeroed=True))
        lib = graph_time(lambda: torch.matmul(dz.t(), x))
        print("| %d | %d | %d | %.1f (%d) | %.1f (%d) | %.1f | %.1f | %.2f |" % (M, K, N, heur, sp, best[0], best[1], old, lib, heur / lib))


if __name__ == "__main__":
    main()
