#!/usr/bin/env python
"""Character-RNN style benchmark (DL4J TextGenerationLSTM / char-modelling example shape): 2 x LSTM(512) + RnnOutputLayer over 77
characters, batch 64, sequence length 128, MultiLayerNetwork.fit with bf16 compute. Reports ms/step eager, HIP-graph captured, and
the same model in plain PyTorch (nn.LSTM -> MIOpen RNN kernels) for orientation."""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.nd4j.learning import Adam  # noqa: E402
from deeplearning4j_amd.nn.conf.base import InputType  # noqa: E402
from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration  # noqa: E402
from deeplearning4j_amd.nn.conf.layers import LSTM, RnnOutputLayer  # noqa: E402
from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork  # noqa: E402

V, Hn, B, T = 77, 512, 64, 128


def build(dtype, graph):
    b = NeuralNetConfiguration.Builder().seed(1).updater(Adam(1e-3)).weightInit("XAVIER")
    if dtype:
        b = b.computeDataType(dtype)
    conf = (b.list().layer(LSTM(nIn=V, nOut=Hn, activation="TANH")).layer(LSTM(nIn=Hn, nOut=Hn, activation="TANH"))
            .layer(RnnOutputLayer(nIn=Hn, nOut=V, activation="SOFTMAX", lossFunction="MCXENT"))
            .setInputType(InputType.recurrent(V, T)).build())
    net = MultiLayerNetwork(conf, device="cuda").init()
    if graph:
        net.enableGraphCapture(True)
    return net


def timed(fn, steps=15, warm=5):
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
    res = {}
    for name, dtype, graph in (("fp32_eager", None, False), ("bf16_eager", "BFLOAT16", False), ("bf16_hip_graph", "BFLOAT16", True)):
        try:
            net = build(dtype, graph)
            res[name] = round(timed(lambda: net.fit(x, y)), 2)
            res[name + "_score"] = round(float(net.score()), 4)
        except Exception as e:                      # keep the other variants
            res[name] = "failed: %s" % (str(e)[:120],)

    class Ref(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.rnn = torch.nn.LSTM(V, Hn, 2, batch_first=True)
            self.out = torch.nn.Linear(Hn, V)

        def forward(self, z):
            return self.out(self.rnn(z)[0])
    ref = Ref().cuda()
    opt = torch.optim.Adam(ref.parameters(), 1e-3)
    xr, yr = x.permute(0, 2, 1).contiguous(), torch.roll(idx, -1, 1).cuda()

    def ref_step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(ref(xr).reshape(-1, V), yr.reshape(-1))
        loss.backward()
        opt.step()
    res["pytorch_nn_lstm_fp32_eager"] = round(timed(ref_step), 2)
    res.update(batch=B, seq_len=T, hidden=Hn, layers=2, unit="ms/step")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
