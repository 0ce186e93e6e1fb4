#!/usr/bin/env python
"""Bidirectional LSTM training step with the two directions on one stream vs on two HIP streams (nn/layers/recurrent.py _side_stream):
2 x GravesBidirectionalLSTM(512) + RnnOutputLayer on [batch 64, 77 features, 50 steps], bf16 and fp32, ms per fit()."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.nd4j.learning import Adam  # noqa: E402
from deeplearning4j_amd.nn.conf.base import InputType  # noqa: E402
from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration  # noqa: E402
from deeplearning4j_amd.nn.conf.layers import GravesBidirectionalLSTM, RnnOutputLayer  # noqa: E402
from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork  # noqa: E402

V, Hn, B, T = 77, 512, 64, 50


def build(dtype):
    b = NeuralNetConfiguration.Builder().seed(1).updater(Adam(1e-3)).weightInit("XAVIER")
    if dtype:
        b = b.computeDataType(dtype)
    conf = (b.list().layer(GravesBidirectionalLSTM(nIn=V, nOut=Hn, activation="TANH"))
            .layer(GravesBidirectionalLSTM(nIn=Hn, nOut=Hn, activation="TANH"))
            .layer(RnnOutputLayer(nIn=Hn, nOut=V, activation="SOFTMAX", lossFunction="MCXENT"))
            .setInputType(InputType.recurrent(V, T)).build())
    return MultiLayerNetwork(conf, device="cuda").init()


def timed(fn, steps=20, warm=6):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    import faulthandler
    if os.environ.get("BIDIR_TRACE"):
        faulthandler.dump_traceback_later(50, file=open(os.environ["BIDIR_TRACE"], "w"), exit=False)
    idx = torch.randint(0, V, (B, T))
    x = torch.nn.functional.one_hot(idx, V).permute(0, 2, 1).float().cuda()
    y = torch.nn.functional.one_hot(torch.roll(idx, -1, 1), V).permute(0, 2, 1).float().cuda()
    res = {}
    for dtype in ("BFLOAT16", None):
        for flag in ("0", "1"):
            os.environ["DL4J_AMD_BIDIR_STREAMS"] = flag
            net = build(dtype)
            print("running", dtype, flag, file=sys.stderr, flush=True)
            res["%s_%s" % ("bf16" if dtype else "fp32", "two_streams" if flag == "1" else "one_stream")] = round(timed(lambda: net.fit(x, y)), 3)
    res.update(batch=B, seq_len=T, hidden=Hn, layers=2, unit="ms per fit()")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
