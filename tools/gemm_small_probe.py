import torch, sys
sys.path.insert(0, "/root/repo")
from deeplearning4j_amd.ops import gemm as G
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for dt in (torch.bfloat16, torch.float32):
    for (M, N, K) in ((576, 2048, 512), (576, 512, 2048), (320, 2048, 512), (320, 512, 2048)):
        x = torch.randn(M, K, device="cuda", dtype=dt); w = torch.randn(N, K, device="cuda", dtype=dt)
        row = ["%s %dx%dx%d" % (str(dt)[6:], M, N, K)]
        for tile in (0, 1, 2, 3):
            row.append("t%d %.1f" % (tile, t(lambda: G.gemm_nt(x, w, tile=tile))))
        wt = w.t().contiguous()
        row.append("lib %.1f" % t(lambda: torch.mm(x, wt)))
        print("  ".join(row), "pick", G.pick_tile(M, N, K, dt))
