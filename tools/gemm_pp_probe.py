#!/usr/bin/env python
"""256x256 ping-pong GEMM: drain-per-tile loop (tile 4) vs half-tile staging with counted waits (tile 10) vs hipBLASLt.
Correctness first (fp32 reference, bitwise agreement of the two loops over repeated runs = race screen), then timings."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import gemm as G  # noqa: E402
from conv_microbench import graph_time  # noqa: E402

torch.manual_seed(0)
bad = 0
for M, N, K in [(256, 256, 64), (256, 256, 128), (512, 768, 200), (1000, 1300, 72), (2048, 2048, 2048), (4096, 3072, 768), (300, 260, 8)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    ref = x.float() @ w.float().t()
    y4 = G.gemm_nt(x, w, tile=4)
    err = ((y4.float() - ref).abs().max() / ref.abs().max()).item()
    worst = 0.0
    nbit = 0
    for rep in range(10):
        y10 = G.gemm_nt(x, w, tile=10)
        worst = max(worst, ((y10.float() - ref).abs().max() / ref.abs().max()).item())
        nbit += int(not torch.equal(y10, y4))
    ok = worst < 1e-2 and nbit == 0
    bad += int(not ok)
    print("check %6d %6d %6d  t4 err %.2e  t10 worst err %.2e  runs differing from t4 bitwise: %d  %s" % (M, N, K, err, worst, nbit, "ok" if ok else "FAIL"), flush=True)
if bad:
    print("FAILED correctness, no timings")
    sys.exit(1)

print("\n| M | N | K | t4 us | t10 us | best small tile us | hipBLASLt us | t10 TF/s | lib/t10 |")
print("|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for M, N, K in [(8192, 4096, 4096), (8192, 8192, 8192), (4096, 4096, 4096), (4096, 4096, 1024), (16384, 1024, 1024), (2048, 2048, 2048), (4096, 3072, 768),
                (4096, 2304, 768), (4096, 768, 3072), (8192, 3072, 768), (16384, 4096, 1024)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    t4 = graph_time(lambda: G.gemm_nt(x, w, tile=4))
    t10 = graph_time(lambda: G.gemm_nt(x, w, tile=10))
    small = min(graph_time(lambda t=t: G.gemm_nt(x, w, tile=t)) for t in (1, 2, 3))
    lib = graph_time(lambda: torch.matmul(x, w.t()))
    print("| %d | %d | %d | %.1f | %.1f | %.1f | %.1f | %.0f | %.2f |" % (M, N, K, t4, t10, small, lib, 2.0 * M * N * K / t10 / 1e6, lib / t10), flush=True)
