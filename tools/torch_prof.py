#!/usr/bin/env python
"""Attribute the ATen (non hand-written) kernels of an eager ResNet-50 training step to ops + shapes + python call sites
(run on the GPU box): ``python tools/torch_prof.py [batch]``."""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.datasets.iterators import BenchmarkDataSetIterator  # noqa: E402
from deeplearning4j_amd.models.resnet50 import ResNet50  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda", 0)
    net = ResNet50(numClasses=1000, computeDataType="BFLOAT16").init(device=dev)
    ds = BenchmarkDataSetIterator([batch, 3, 224, 224], 1000, 1, device=dev, seed=1).next()
    x, y = ds.getFeatures(), ds.getLabels()
    for _ in range(3):
        net.fit(x, y)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for _ in range(2):
            net.fit(x, y)
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=40,
                                                              max_shapes_column_width=70))
    print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=40))


if __name__ == "__main__":
    main()
