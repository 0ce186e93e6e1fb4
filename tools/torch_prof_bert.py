#!/usr/bin/env python
"""Attribute the launches of one eager BERT-base SameDiff training step to ATen ops and python call sites (run on the GPU box)."""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.autodiff.samediff import SameDiff, TrainingConfig  # noqa: E402
from deeplearning4j_amd.models.bert import BertConfig, build_bert_classifier  # noqa: E402
from deeplearning4j_amd.nd4j.learning import Adam  # noqa: E402


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    sd = SameDiff.create("cuda")
    cfg = BertConfig.base()
    cfg.numLayers = layers
    build_bert_classifier(sd, cfg, 128, 2, computeDataType="BFLOAT16")
    from deeplearning4j_amd.autodiff.optimize import GraphOptimizer
    GraphOptimizer.optimize(sd)
    sd.setTrainingConfig(TrainingConfig.builder().updater(Adam(2e-5)).build())
    g = torch.Generator().manual_seed(0)
    ph = {"tokenIdx": torch.randint(0, cfg.vocabSize, (32, 128), generator=g).cuda(), "mask": torch.ones(32, 128).cuda(),
          "label": torch.eye(2)[torch.randint(0, 2, (32,), generator=g)].cuda()}
    for _ in range(3):
        sd.fit(ph)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        sd.fit(ph)
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=36,
                                                              max_shapes_column_width=60))
    print(prof.key_averages(group_by_stack_n=8).table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=36,
                                                       max_src_column_width=110))


if __name__ == "__main__":
    main()
