#!/usr/bin/env python
"""One LSTM layer's recurrence (ops/lstm.py seq_forward + seq_backward) at the char-RNN shape: time per step of the step kernels.
`DL4J_AMD_LSTM_DIAG1=0` selects the 8-unit x all-rows step kernels, the default is the one-tile-per-workgroup form."""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.ops import lstm as LS  # noqa: E402


def main():
    T, mb, n = 50, 64, 512
    res = {}
    for dt in (torch.bfloat16, torch.float32):
        g = torch.Generator(device="cuda").manual_seed(0)
        xw = (torch.randn(T, mb, 4 * n, device="cuda", generator=g) * 0.5).to(dt)
        RW4 = (torch.randn(4 * n, n, device="cuda", generator=g) / n ** 0.5).to(dt).t()      # 'f' order [n, 4n]
        h0 = torch.zeros(mb, n, device="cuda", dtype=dt)
        E = torch.randn(T, mb, n, device="cuda", generator=g).to(dt)

        def fwd():
            return LS.seq_forward(xw, RW4, h0, h0, None, None, True, False)

        def both():
            _, st = fwd()
            LS.seq_backward(E, RW4, st, False)
        for name, fn in (("fwd", fwd), ("fwd_bwd", both)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            res["%s_%s_us_per_step" % (str(dt).split(".")[-1], name)] = round((time.perf_counter() - t0) / 20 / T * 1e6, 2)
    res["diag1"] = LS._DIAG1
    print(json.dumps(res))


if __name__ == "__main__":
    main()
