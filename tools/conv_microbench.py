#!/usr/bin/env python
"""Per-layer efficiency of the implicit-GEMM conv kernels on the zoo ResNet-50 shapes (batch 256), timed through HIP-graph
replay so host launch overhead is excluded.  Prints time, TFLOP/s and the min-traffic GB/s for fwd / dgrad / wgrad."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import conv_igemm as IG  # noqa: E402

N = 256
# (C, H, K, R, stride, count-in-network)
SHAPES = [(64, 28, 64, 1, 1, 1), (64, 28, 64, 3, 1, 3), (64, 28, 256, 1, 1, 3), (256, 28, 64, 1, 1, 2),
          (64, 56, 64, 1, 2, 1), (64, 56, 256, 1, 2, 1),
          (256, 28, 128, 1, 2, 1), (256, 28, 512, 1, 2, 1), (128, 14, 128, 3, 1, 4), (128, 14, 512, 1, 1, 4), (512, 14, 128, 1, 1, 3),
          (512, 14, 256, 1, 2, 1), (512, 14, 1024, 1, 2, 1), (256, 7, 256, 3, 1, 6), (256, 7, 1024, 1, 1, 6), (1024, 7, 256, 1, 1, 5),
          (1024, 7, 512, 1, 2, 1), (1024, 7, 2048, 1, 2, 1), (512, 4, 512, 3, 1, 3), (512, 4, 2048, 1, 1, 3), (2048, 4, 512, 1, 1, 2)]


def graph_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def main():
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print("| C | HxW | K | RxR/s | n | fwd us | TF/s | GB/s | dgrad us | TF/s | wgrad us | TF/s | GB/s |")
    print("|---:|---:|---:|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for C, H, K, R, s, cnt in SHAPES:
        p = R // 2
        x = torch.randn(N, C, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, C, R, R, device="cuda") / (C * R * R) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
        wd = torch.flip(w, dims=[2, 3]).permute(1, 2, 3, 0).contiguous()
        y, _ = IG.forward(x, w, None, (s, s), (p, p), (1, 1), want_stats=True)
        dz = torch.randn_like(y)
        dw = torch.empty(K, C, R, R, device="cuda")
        OH = y.shape[2]
        flops = 2.0 * N * OH * OH * K * C * R * R
        io = (x.numel() + y.numel()) * 2
        tf = graph_time(lambda: IG.forward(x, w, None, (s, s), (p, p), (1, 1), want_stats=True))
        td = graph_time(lambda: IG.dgrad(dz, x.shape, wd, (s, s), (p, p), (1, 1))) if IG.dgrad_supported(w, (s, s), (p, p)) else float("nan")
        tw = graph_time(lambda: IG.wgrad(dz, x, w.shape, (s, s), (p, p), (1, 1), dw_out=dw))
        print("| %d | %d | %d | %dx%d/%d | %d | %.1f | %.0f | %.0f | %.1f | %.0f | %.1f | %.0f | %.0f |" % (
            C, H, K, R, R, s, cnt, tf, flops / tf / 1e6, io / tf / 1e3, td, flops / td / 1e6, tw, flops / tw / 1e6, io / tw / 1e3))
        tot["fwd"] += tf * cnt
        tot["dgrad"] += (td if td == td else 0) * cnt
        tot["wgrad"] += tw * cnt
    print("\nnetwork totals (ms/step): fwd %.2f  dgrad %.2f  wgrad %.2f" % (tot["fwd"] / 1e3, tot["dgrad"] / 1e3, tot["wgrad"] / 1e3))


if __name__ == "__main__":
    main()
