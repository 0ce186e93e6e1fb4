#!/usr/bin/env python
"""Secondary benchmark (SURVEY config 4): BERT-base sequence classification as a SameDiff graph, bf16 compute with fp32 master
variables, batch 32 x sequence 128, synthetic tokens, Adam. Prints one JSON line (tokens/sec, ms/step)."""
import argparse
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.autodiff.samediff import SameDiff, TrainingConfig  # noqa: E402
from deeplearning4j_amd.models.bert import BertConfig, build_bert_classifier  # noqa: E402
from deeplearning4j_amd.nd4j.learning import Adam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="BFLOAT16")
    ap.add_argument("--imported", action="store_true", help="run the TF-GraphDef round trip: export -> TFGraphMapper import -> optimize")
    ap.add_argument("--no-fuse", action="store_true", help="skip the graph optimizer (LinearFusion: fused_dense ops)")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the training step into a HIP graph")
    args = ap.parse_args()
    cfg = BertConfig.base()
    if args.imported:
        # SURVEY config 4 as the reference runs it: a frozen TF graph (written here by the in-tree GraphDef writer: no network to
        # fetch one) -> TFGraphMapper -> constants to variables -> graph optimizer (layer_norm / gelu / attention / fused_dense
        # rediscovered from TF primitives) -> bf16 mixed-precision pass -> fine-tuning
        from deeplearning4j_amd.autodiff.export_tf import bert_graphdef
        from deeplearning4j_amd.autodiff.import_tf import TFGraphMapper
        from deeplearning4j_amd.autodiff.optimize import GraphOptimizer, MixedPrecision
        src = SameDiff.create("cpu")
        build_bert_classifier(src, cfg, args.seq, 2)
        weights = {n: src.getArrForVarName(n).tensor() for n in src.trainableVariables()}
        sd = TFGraphMapper.importGraph(bert_graphdef(cfg, args.seq, 2, weights), device="cuda")
        del src
        sd.convertToVariables(list(weights))
        label = sd.placeHolder("label", "FLOAT", -1, 2)
        sd.loss.softmaxCrossEntropy(label, sd.getVariable("logits"), None, 3, name="loss").markAsLoss()
        GraphOptimizer.optimize(sd, ["logits", "loss"])
        if args.dtype.upper() in ("BFLOAT16", "BF16"):
            MixedPrecision.apply(sd, "BFLOAT16")
    else:
        sd = SameDiff.create("cuda")
        build_bert_classifier(sd, cfg, args.seq, 2, computeDataType=args.dtype)
        if not args.no_fuse:
            from deeplearning4j_amd.autodiff.optimize import GraphOptimizer
            GraphOptimizer.optimize(sd)
    sd.setTrainingConfig(TrainingConfig.builder().updater(Adam(2e-5)).build())
    if not args.no_graph:
        sd.enableGraphCapture(True)
    g = torch.Generator().manual_seed(0)
    ph = {"tokenIdx": torch.randint(0, cfg.vocabSize, (args.batch, args.seq), generator=g).to(torch.int32 if args.imported else torch.int64).cuda(),
          "mask": torch.ones(args.batch, args.seq).cuda(),
          "label": torch.eye(2)[torch.randint(0, 2, (args.batch,), generator=g)].cuda()}
    for _ in range(args.warmup):
        sd.fit(ph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sd.fit(ph)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    nparams = sum(int(v.getArr().tensor().numel()) for v in sd.variables.values() if v.variableType == "VARIABLE")
    print(json.dumps({"metric": "tokens/sec BERT-base SameDiff fine-tuning step", "value": round(args.batch * args.seq / dt, 1),
                      "ms_per_step": round(dt * 1e3, 2), "batch": args.batch, "seq_len": args.seq, "dtype": args.dtype,
                      "params": nparams, "graph_source": "tf_graphdef_import" if args.imported else "samediff_builder", "hip_graph": bool(getattr(sd, "_cap", None) and sd._cap.get("graph") is not None), "data": "synthetic", "final_loss": round(float(sd.output(ph, ["loss"])["loss"]), 4)}))


if __name__ == "__main__":
    main()
