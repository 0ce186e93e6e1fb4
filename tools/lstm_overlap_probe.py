#!/usr/bin/env python
"""Do the per-step LSTM kernels of independent sequences overlap on the chip? K independent (T = 50, mb = 64, n = 512, bf16) recurrences are
issued on K streams (no cross-stream dependencies) and timed against the same work on one stream. Separates "kernels cannot share the chip"
from "the pipeline's event hand-offs cost time" for nn/layers/lstm_stack.py."""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.native import lib as L  # noqa: E402

T, mb, n = 50, 64, 512
dev = "cuda"


def mk():
    bf = torch.bfloat16
    d = dict(xw=torch.randn(T, mb, 4 * n, device=dev).to(bf) * 0.1, rw=(torch.randn(4 * n, n, device=dev) * 0.03).to(bf),
             h0=torch.zeros(mb, n, device=dev, dtype=bf), c0=torch.zeros(mb, n, device=dev, dtype=bf),
             H=torch.empty(T, mb, n, device=dev, dtype=bf), Cs=torch.empty(T, mb, n, device=dev, dtype=bf),
             SV=torch.empty(T, mb, 6 * n, device=dev, dtype=bf), bar=torch.empty(2, device=dev, dtype=torch.int32),
             E=(torch.randn(T, mb, n, device=dev) * 0.1).to(bf), rwt=(torch.randn(n, 4 * n, device=dev) * 0.03).to(bf),
             dG=torch.empty(T, mb, 4 * n, device=dev, dtype=bf), dc=torch.empty(2, mb, n, device=dev, dtype=torch.float32))
    return d


def fwd(d, st, s0=0, s1=T):
    L.check(L.hip().dl4j_lstm_steps_fwd_range(L.ptr(d["xw"]), L.ptr(d["rw"].t()), L.c_int64(n), L.ptr(d["h0"]), L.ptr(d["c0"]), L.ptr(None), L.ptr(None),
                                              L.ptr(None), L.ptr(None), L.ptr(d["H"]), L.ptr(d["Cs"]), L.ptr(d["SV"]), L.ptr(d["bar"]), L.c_int(T),
                                              L.c_int(mb), L.c_int(n), L.c_int(0), L.c_int(s0), L.c_int(s1), L.c_void_p(st.cuda_stream)), "fwd")


def bwd(d, st, s0=0, s1=T):
    L.check(L.hip().dl4j_lstm_steps_bwd_range(L.ptr(d["E"]), L.ptr(d["rwt"]), L.c_int64(4 * n), L.ptr(d["SV"]), L.ptr(d["Cs"]), L.ptr(d["c0"]),
                                              L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(d["dG"]), L.ptr(None), L.ptr(d["bar"]),
                                              L.ptr(d["dc"]), L.ptr(None), L.c_int(T), L.c_int(mb), L.c_int(n), L.c_int(0), L.c_int(s0), L.c_int(s1),
                                              L.c_void_p(st.cuda_stream)), "bwd")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    sets = [mk() for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    cur = torch.cuda.current_stream()
    for d in sets:
        fwd(d, cur)
    torch.cuda.synchronize()
    res = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for k in (1, 2, 3, 4):
            def one_stream():
                for d in sets[:k]:
                    fn(d, streams[0])

            def k_streams():
                for i in range(k):
                    streams[i].wait_stream(cur)
                # interleave the host-side issue in chunks of 10 steps so that no stream starves while another is being fed
                for s0 in range(0, T, 10):
                    for i in range(k):
                        if name == "fwd":
                            fn(sets[i], streams[i], s0, s0 + 10)
                        else:
                            fn(sets[i], streams[i], T - s0 - 10, T - s0)
                for i in range(k):
                    cur.wait_stream(streams[i])
            res["%s_x%d_one_stream_us" % (name, k)] = round(timed(one_stream), 1)
            res["%s_x%d_k_streams_us" % (name, k)] = round(timed(k_streams), 1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
