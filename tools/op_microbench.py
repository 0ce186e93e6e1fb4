#!/usr/bin/env python
"""Custom-op micro-benchmarks (the role of the reference's contrib/benchmarking_nd4j JMH harness): every op below is executed through the
op executioner twice, on its HIP platform-helper route (native/csrc kernels) and on the generic implementation (ATen composition), at a
size where the kernel rather than the launch is measured. Prints a Markdown table (us per call, speed-up).
usage: python tools/op_microbench.py [out.md]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.nd4j.ops.custom import OpExecutioner, HELPER_HIT  # noqa: E402

dev = "cuda" if torch.cuda.is_available() else "cpu"
g = torch.Generator(device=dev).manual_seed(0)


def R(*s, dt=torch.float32):
    return torch.randn(*s, device=dev, generator=g).to(dt)


def I(*v, dt=torch.int64):
    return torch.tensor(v, device=dev, dtype=dt)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def cases():
    bf = torch.bfloat16
    img = R(64, 64, 56, 56)                      # NCHW
    nhwc = R(32, 112, 112, 3).abs()
    idx = torch.randint(0, 50000, (65536,), device=dev, generator=g)
    seg = torch.randint(0, 1024, (1 << 20,), device=dev, generator=g)
    boxes = torch.rand(4096, 4, device=dev, generator=g)
    boxes = torch.cat([boxes[:, :2] * 0.8, boxes[:, :2] * 0.8 + boxes[:, 2:] * 0.2 + 0.01], 1)
    return [
        ("softmax [8192, 4096] bf16", "softmax", [R(8192, 4096, dt=bf)], [1], [], []),
        ("softmax [8192, 4096] fp32", "softmax", [R(8192, 4096)], [1], [], []),
        ("log_softmax [8192, 4096] fp32", "log_softmax", [R(8192, 4096)], [1], [], []),
        ("layer_norm [16384, 768] bf16", "layer_norm", [R(16384, 768, dt=bf), R(768, dt=bf), R(768, dt=bf)], [1], [], []),
        ("layer_norm [16384, 768] fp32", "layer_norm", [R(16384, 768), R(768), R(768)], [1], [], []),
        ("biasadd [64,256,56,56] NCHW", "biasadd", [R(64, 256, 56, 56), R(256)], [], [], [True]),
        ("lrn [64,64,56,56] depth 2", "lrn", [img], [2], [1.0, 1e-4, 0.75], []),
        ("cumsum [4096, 4096] dim 1", "cumsum", [R(4096, 4096)], [0, 0, 1], [], []),
        ("top_k k=8 [8192, 4096]", "top_k", [R(8192, 4096)], [8], [], []),
        ("nth_element n=100 [8192, 4096]", "nth_element", [R(8192, 4096), I(100)], [0], [], []),
        ("embedding_lookup 65536 x [50000, 256]", "embedding_lookup", [R(50000, 256), idx], [], [], []),
        ("gather axis 0: 65536 of [50000, 256]", "gather", [R(50000, 256), idx], [0], [], []),
        ("onehot 65536 -> depth 1024", "onehot", [idx % 1024], [-1, 1024], [1.0, 0.0], []),
        ("unsorted_segment_sum [1M, 16] -> 1024", "unsorted_segment_sum", [R(1 << 20, 16), seg], [1024], [], []),
        ("im2col 3x3 s1 p1 [64,64,56,56]", "im2col", [img], [3, 3, 1, 1, 1, 1, 1, 1, 0], [0.0], []),
        ("upsampling2d x2 [64,64,56,56]", "upsampling2d", [img], [2, 2, 1], [], []),
        ("pad [64,64,56,56] +1 spatial", "pad", [img, torch.tensor([[0, 0], [0, 0], [1, 1], [1, 1]], device=dev)], [0], [0.0], []),
        ("reverse dims 3 [64,64,56,56]", "reverse", [img], [3], [], []),
        ("tile 2x2 [4096, 1024]", "tile", [R(4096, 1024)], [2, 2], [], []),
        ("roll 17 axis 1 [8192, 4096]", "roll", [R(8192, 4096)], [17, 1], [], []),
        ("concat axis 1 2 x [8192, 2048]", "concat", [R(8192, 2048), R(8192, 2048)], [1], [], []),
        ("space_to_depth bs 2 NHWC [32,112,112,3]", "space_to_depth", [nhwc], [2, 1], [], []),
        ("depthwise_conv2d 3x3 [64,64,56,56]", "depthwise_conv2d", [img, R(3, 3, 64, 1)], [3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0], [], []),
        ("adjust_contrast [32,112,112,3]", "adjust_contrast", [nhwc], [], [1.7], []),
        ("histogram_fixed_width 16M -> 256 bins", "histogram_fixed_width", [R(1 << 24), torch.tensor([-3.0, 3.0], device=dev), I(256)], [], [], []),
        ("non_max_suppression 4096 boxes -> 100", "non_max_suppression", [boxes, torch.rand(4096, device=dev, generator=g)], [100], [0.5], []),
        ("matmul [4096,1024] x [1024,4096] bf16", "matmul", [R(4096, 1024, dt=bf), R(1024, 4096, dt=bf)], [], [], []),
        ("xw_plus_b [8192,1024] x [1024,1024] bf16", "xw_plus_b", [R(8192, 1024, dt=bf), R(1024, 1024, dt=bf), R(1024, dt=bf)], [], [], []),
    ]


def main():
    ex = OpExecutioner.getInstance()
    from deeplearning4j_amd.nd4j.ops import cost_model
    lines = ["# Custom-op micro-benchmarks on MI355X: the route the executioner takes by default (platform helper unless the measured cost "
             "model declines it) vs the generic (ATen) implementation, us per call (tools/op_microbench.py)", "",
             "| case | default route us | generic us | default vs generic | route taken | forced helper us |", "|---|---:|---:|---:|---|---:|"]
    for label, name, ins, i, t, b in cases():
        row = [label]
        try:
            cost_model.ENABLED = True
            hit0 = HELPER_HIT[name]
            ex._run(name, ins, i, t, b, [], True)
            ran = HELPER_HIT[name] > hit0
            why = cost_model.prefer_generic(name, ins, i, t, b, [])
            td = timed(lambda: ex._run(name, ins, i, t, b, [], True), 30)
            tg = timed(lambda: ex._run(name, ins, i, t, b, [], False), 30)
            if not ran:
                td = tg = min(td, tg)          # the same code ran both times: one number, not two samples of its noise
            cost_model.ENABLED = False
            tf = timed(lambda: ex._run(name, ins, i, t, b, [], True), 30)
            cost_model.ENABLED = True
            row += ["%.1f" % td, "%.1f" % tg, "%.2fx" % (tg / td), "helper" if ran else "generic (cost model: %s)" % (why or "?")[:90], "%.1f" % tf]
        except Exception as e:                     # keep the table going; the failure is part of the report
            row += ["failed: %s" % (str(e)[:80].replace("|", "/"),), "", "", ""]
        lines.append("| " + " | ".join(row) + " |")
        print(lines[-1], flush=True)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
