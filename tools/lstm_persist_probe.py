#!/usr/bin/env python
"""Time the persistent LSTM forward / backward kernels alone (char-RNN shape) and the forward without the hand-off (bit 1 of `reverse`)."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.native import lib as L  # noqa: E402
from deeplearning4j_amd.ops import lstm as R  # noqa: E402

T, mb, n = 50, 64, 512
g = torch.Generator().manual_seed(0)
xw = (torch.randn(T, mb, 4 * n, generator=g) * 0.5).bfloat16().cuda()
RW4 = (torch.randn(4 * n, n, generator=g) / n ** 0.5).bfloat16().cuda().t()
h0 = torch.zeros(mb, n).bfloat16().cuda()
c0 = torch.zeros(mb, n).bfloat16().cuda()
E = (torch.randn(T, mb, n, generator=g) * 0.5).bfloat16().cuda()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


H, st = R.seq_forward(xw, RW4, h0, c0, None, None, True, False)
R._PERSIST_MODE = "grid"
print("fwd persist      %.1f us (%d steps)" % (timeit(lambda: R.seq_forward(xw, RW4, h0, c0, None, None, True, False)), T))
print("bwd persist      %.1f us" % timeit(lambda: R.seq_backward(E, RW4, st, False)))
Hh, Cs, SV = torch.empty_like(H), torch.empty_like(H), torch.empty(T, mb, 6 * n, device="cuda", dtype=torch.bfloat16)
bar = torch.zeros(2, device="cuda", dtype=torch.int32)


def raw(flag):
    L.check(L.hip().dl4j_lstm_persist_fwd(L.ptr(xw), L.ptr(RW4), L.c_int64(n), L.ptr(h0), L.ptr(c0), None, None, None, None, L.ptr(Hh), L.ptr(Cs),
                                          L.ptr(SV), L.ptr(bar), L.c_int(T), L.c_int(mb), L.c_int(n), L.c_int(flag), L.stream()), "probe")


print("fwd raw          %.1f us" % timeit(lambda: raw(0)))
print("fwd no hand-off  %.1f us" % timeit(lambda: raw(2)))
for bits, what in ((2 | 4, "no SV/Cs stores"), (2 | 8, "no h copy"), (2 | 16, "no MFMA"), (2 | 4 | 8 | 16, "none of them"), (4, "hand-off, no SV/Cs stores")):
    print("fwd %-26s %.1f us" % (what, timeit(lambda: raw(bits))))
def steps():
    L.check(L.hip().dl4j_lstm_steps_fwd(L.ptr(xw), L.ptr(RW4), L.c_int64(n), L.ptr(h0), L.ptr(c0), None, None, None, None, L.ptr(Hh), L.ptr(Cs),
                                        L.ptr(SV), L.ptr(bar), L.c_int(T), L.c_int(mb), L.c_int(n), L.c_int(0), L.stream()), "steps")


print("fwd fused step launches  %.1f us" % timeit(steps))
R._PERSIST_MODE = "steps"
print("bwd fused step launches  %.1f us" % timeit(lambda: R.seq_backward(E, RW4, st, False)))
steps()
torch.cuda.synchronize()
print("   max |H steps - H persist| = %.4g" % float((Hh.float() - H.float()).abs().max()))
R._PERSIST = False
print("fwd per-step     %.1f us" % timeit(lambda: R.seq_forward(xw, RW4, h0, c0, None, None, True, False)))
print("bwd per-step     %.1f us" % timeit(lambda: R.seq_backward(E, RW4, st, False)))
