#!/usr/bin/env python
"""`gru` op (fused step kernels, native/csrc/gru_seq.hip) against torch.nn.GRU (MIOpen) on MI355X: forward + backward of one layer,
ms per call. usage: python tools/bench_gru.py [out.md]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.nd4j.ops.custom import exec_op  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def case(T, B, n, nIn=None):
    nIn = nIn or n
    g = torch.Generator(device="cuda").manual_seed(0)
    bf = torch.bfloat16
    x = torch.randn(T, B, nIn, device="cuda", generator=g)
    h0 = torch.zeros(B, n, device="cuda")
    Wx = torch.randn(nIn, 3 * n, device="cuda", generator=g) / nIn ** 0.5
    Wh = torch.randn(n, 3 * n, device="cuda", generator=g) / n ** 0.5
    b = torch.zeros(3 * n, device="cuda")
    dH = torch.randn(T, B, n, device="cuda", generator=g)
    rows = {}

    def ours(dt):
        ins = [v.to(dt).requires_grad_(True) for v in (x, h0, Wx, Wh, b)]
        d = dH.to(dt)

        def f():
            out = exec_op("gru", ins)[0]
            torch.autograd.grad(out, ins, d)
        return f

    def fwd_only(dt):
        ins = [v.to(dt) for v in (x, h0, Wx, Wh, b)]

        def f():
            with torch.no_grad():
                exec_op("gru", ins)
        return f

    def torch_gru(dt):
        m = torch.nn.GRU(nIn, n).cuda().to(dt)
        xi, hi, d = x.to(dt).requires_grad_(True), h0.to(dt).unsqueeze(0).requires_grad_(True), dH.to(dt)

        def f():
            out = m(xi, hi)[0]
            torch.autograd.grad(out, [xi, hi] + list(m.parameters()), d)
        return f

    def torch_fwd(dt):
        m = torch.nn.GRU(nIn, n).cuda().to(dt)
        xi, hi = x.to(dt), h0.to(dt).unsqueeze(0)

        def f():
            with torch.no_grad():
                m(xi, hi)
        return f
    def safe(mk):
        try:
            return timed(mk())
        except Exception as e:          # noqa: BLE001 - e.g. a library without a bf16 RNN: the row says so
            print("variant failed:", str(e)[:120], flush=True)
            return float("inf")
    rows["ours bf16 fwd+bwd"] = safe(lambda: ours(bf))
    rows["ours fp32 fwd+bwd"] = safe(lambda: ours(torch.float32))
    rows["torch.nn.GRU bf16 fwd+bwd"] = safe(lambda: torch_gru(bf))
    rows["torch.nn.GRU fp32 fwd+bwd"] = safe(lambda: torch_gru(torch.float32))
    rows["ours bf16 fwd"] = safe(lambda: fwd_only(bf))
    rows["torch.nn.GRU bf16 fwd"] = safe(lambda: torch_fwd(bf))
    rows["torch.nn.GRU fp32 fwd"] = safe(lambda: torch_fwd(torch.float32))
    return rows


def main():
    lines = ["# gru op vs torch.nn.GRU on MI355X (tools/bench_gru.py), ms per call", ""]
    for T, B, n in ((64, 128, 512), (128, 64, 512), (64, 128, 256)):
        r = case(T, B, n)
        best_torch = min(r["torch.nn.GRU bf16 fwd+bwd"], r["torch.nn.GRU fp32 fwd+bwd"])
        lines += ["## T=%d, batch=%d, n=nIn=%d" % (T, B, n), "", "| variant | ms |", "|---|---:|"]
        lines += ["| %s | %.3f |" % kv for kv in r.items()]
        lines += ["", "training step speed-up over the faster torch.nn.GRU variant: **%.2fx**; inference: %.2fx" % (
            best_torch / r["ours bf16 fwd+bwd"], min(r["torch.nn.GRU bf16 fwd"], r["torch.nn.GRU fp32 fwd"]) / r["ours bf16 fwd"]), ""]
        print("\n".join(lines[-14:]), flush=True)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
