#!/usr/bin/env python
"""Run a few bf16 char-RNN fits (BASELINE config 3 shape) for a rocprofv3 kernel trace: `rocprofv3 --kernel-trace --stats -d out -- python3 tools/prof_lstm.py`."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tools.bench_lstm import B, T, V, build  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    g = torch.Generator().manual_seed(0)
    idx = torch.randint(0, V, (B, T), generator=g)
    x = torch.nn.functional.one_hot(idx, V).permute(0, 2, 1).float().cuda()
    y = torch.nn.functional.one_hot(torch.roll(idx, -1, 1), V).permute(0, 2, 1).float().cuda()
    net = build("BFLOAT16", False)
    for _ in range(3):
        net.fit(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        net.fit(x, y)
    torch.cuda.synchronize()
    print("ms/fit %.2f" % ((time.perf_counter() - t0) / steps * 1e3))


if __name__ == "__main__":
    main()
