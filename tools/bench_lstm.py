#!/usr/bin/env python
"""Character-RNN benchmark on BASELINE.json config 3: GravesLSTM / LSTM char-RNN, 3 x 512 units, truncated BPTT 50 (the DL4J
char-modelling example shape: one-hot characters in, softmax over the vocabulary out, MultiLayerNetwork.fit). One "step" = one fit()
of a [batch 64, 77 chars, 200 time steps] minibatch = 4 tBPTT segments of 50 steps, each with its own forward, backward and update.
Reports ms/step for fp32 eager, bf16 eager, bf16 HIP-graph, and the same model in plain PyTorch (nn.LSTM -> MIOpen RNN kernels,
same segmenting with detached state) for orientation."""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.nd4j.learning import Adam  # noqa: E402
from deeplearning4j_amd.nn.conf.base import InputType, BackpropType  # noqa: E402
from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration  # noqa: E402
from deeplearning4j_amd.nn.conf.layers import LSTM, GravesLSTM, RnnOutputLayer  # noqa: E402
from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork  # noqa: E402

V, Hn, B, T, TB, NL = 77, 512, 64, 200, 50, 3


def build(dtype, graph, graves=False):
    b = NeuralNetConfiguration.Builder().seed(1).updater(Adam(1e-3)).weightInit("XAVIER")
    if dtype:
        b = b.computeDataType(dtype)
    Lc = GravesLSTM if graves else LSTM
    lb = b.list()
    for i in range(NL):
        lb = lb.layer(Lc(nIn=V if i == 0 else Hn, nOut=Hn, activation="TANH"))
    conf = (lb.layer(RnnOutputLayer(nIn=Hn, nOut=V, activation="SOFTMAX", lossFunction="MCXENT"))
            .backpropType(BackpropType.TruncatedBPTT).tBPTTForwardLength(TB).tBPTTBackwardLength(TB)
            .setInputType(InputType.recurrent(V, T)).build())
    net = MultiLayerNetwork(conf, device="cuda").init()
    if graph:
        net.enableGraphCapture(True)
    return net


def timed(fn, steps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    idx = torch.randint(0, V, (B, T), generator=g)
    x = torch.nn.functional.one_hot(idx, V).permute(0, 2, 1).float().cuda()
    y = torch.nn.functional.one_hot(torch.roll(idx, -1, 1), V).permute(0, 2, 1).float().cuda()
    import os
    only = [v for v in os.environ.get("BENCH_LSTM_ONLY", "").split(",") if v]      # e.g. bf16_eager,fp32_eager: skip the rest
    res = {}
    for name, dtype, graph, graves in (("fp32_eager", None, False, False), ("bf16_eager", "BFLOAT16", False, False),
                                       ("bf16_graves_eager", "BFLOAT16", False, True), ("bf16_hip_graph", "BFLOAT16", True, False)):
        if only and name not in only:
            continue
        try:
            net = build(dtype, graph, graves)
            res[name] = round(timed(lambda: net.fit(x, y)), 2)
            res[name + "_score"] = round(float(net.score()), 4)
        except Exception as e:                      # keep the other variants
            res[name] = "failed: %s" % (str(e)[:160],)

    if only:
        print(json.dumps(res))
        return

    class Ref(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.rnn = torch.nn.LSTM(V, Hn, NL, batch_first=True)
            self.out = torch.nn.Linear(Hn, V)

        def forward(self, z, st):
            o, st = self.rnn(z, st)
            return self.out(o), st
    ref = Ref().cuda()
    opt = torch.optim.Adam(ref.parameters(), 1e-3)
    xr, yr = x.permute(0, 2, 1).contiguous(), torch.roll(idx, -1, 1).cuda()

    def ref_step():
        st = None
        for s in range(0, T, TB):
            opt.zero_grad(set_to_none=True)
            o, st = ref(xr[:, s:s + TB], st)
            st = tuple(v.detach() for v in st)
            loss = torch.nn.functional.cross_entropy(o.reshape(-1, V), yr[:, s:s + TB].reshape(-1))
            loss.backward()
            opt.step()
    res["pytorch_nn_lstm_fp32_eager"] = round(timed(ref_step), 2)

    # the same PyTorch model under bf16 autocast (MIOpen's bf16 RNN path where it has one), for a like-for-like bf16 comparison
    def ref_step_bf16():
        st = None
        for s in range(0, T, TB):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                o, st = ref(xr[:, s:s + TB], st)
            st = tuple(v.detach() for v in st)
            loss = torch.nn.functional.cross_entropy(o.float().reshape(-1, V), yr[:, s:s + TB].reshape(-1))
            loss.backward()
            opt.step()
    try:
        res["pytorch_nn_lstm_bf16_autocast_eager"] = round(timed(ref_step_bf16), 2)
    except Exception as e:
        res["pytorch_nn_lstm_bf16_autocast_eager"] = "failed: %s" % (str(e)[:120],)
    best = min(v for k, v in res.items() if isinstance(v, float) and not k.endswith("_score") and not k.startswith("pytorch"))
    res.update(batch=B, seq_len=T, tbptt=TB, hidden=Hn, layers=NL, unit="ms per minibatch fit (4 tBPTT segments)",
               chars_per_sec_best=round(B * T / best * 1e3))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
