"""Per-step score of the bench network (same model, data and updater as bench.py): finds the first non-finite step and names the
parameters that went non-finite.  `python tools/nan_probe.py [--steps N] [--no-graph]`"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch


def trace_conv_backward():
    """Wrap ops.conv.conv2d_backward: report the first call whose outputs are non-finite although its inputs are finite."""
    from deeplearning4j_amd.ops import conv as CONV
    inner = CONV.conv2d_backward
    state = {"n": 0, "done": False}

    def fin(t):
        return t is None or bool(torch.isfinite(t.float()).all())

    def wrapped(dz, x, w, stride, padding, dilation, need_dx=True, need_dw=True, need_db=False, w_dgrad=None, dw_out=None,
                dx_accum=None, dw_prezeroed=False, dx_acc_masked=None, bn_sink=None):
        state["n"] += 1
        ins = {"dz": fin(dz), "x": fin(x), "w": fin(w), "dx_accum": fin(dx_accum), "dw_out_in": fin(dw_out),
               "masked_src": fin(getattr(dx_acc_masked, "src", None))}
        dx, dW, db = inner(dz, x, w, stride, padding, dilation, need_dx, need_dw, need_db, w_dgrad, dw_out, dx_accum, dw_prezeroed,
                           dx_acc_masked, bn_sink)
        torch.cuda.synchronize()
        outs = {"dx": fin(dx), "dW": fin(dW), "dw_out": fin(dw_out), "db": fin(db),
                "sink_partial": fin(getattr(bn_sink, "partial", None)), "sink_sums": fin(getattr(bn_sink, "sums", None)),
                "sink_dbeta": fin(getattr(bn_sink, "dbeta_out", None)), "sink_dgamma": fin(getattr(bn_sink, "dgamma_out", None))}
        if not state["done"] and not all(outs.values()):
            state["done"] = True
            print("conv backward call", state["n"], "dz", tuple(dz.shape), "x", tuple(x.shape), "w", tuple(w.shape), "stride", stride,
                  "pad", padding, "ins", ins, "outs", outs, "accum", dx_accum is not None, "masked", dx_acc_masked is not None,
                  "sink", bn_sink is not None, "sink_gy", getattr(bn_sink, "gy", None), flush=True)
            part = getattr(bn_sink, "partial", None)
            if part is not None:
                bad = (~torch.isfinite(part.float())).nonzero()
                print("partial shape", tuple(part.shape), "non-finite entries", bad.shape[0], "first", bad[:8].tolist(), flush=True)
            if dx is not None and not outs["dx"]:
                bad = (~torch.isfinite(dx.float())).nonzero()
                print("dx non-finite entries", bad.shape[0], "first", bad[:8].tolist(), flush=True)
        return dx, dW, db

    CONV.conv2d_backward = wrapped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    from deeplearning4j_amd.models.resnet50 import ResNet50
    from deeplearning4j_amd.datasets.iterators import BenchmarkDataSetIterator
    dev = torch.device("cuda", 0)
    if args.trace:
        trace_conv_backward()
    net = ResNet50(numClasses=1000, inputShape=(3, 224, 224), computeDataType="BFLOAT16").init(device=dev)
    if not args.no_graph:
        net.enableGraphCapture(True)
    ds = BenchmarkDataSetIterator([args.batch, 3, 224, 224], 1000, 1, device=dev, seed=1234).next()
    x, y = ds.getFeatures(), ds.getLabels()
    scores = []
    for i in range(args.steps):
        net.fit(x, y)
        s = float(net.score())
        scores.append(round(s, 3))
        if not math.isfinite(s):
            bad, good = [], []
            for name, p in net.paramTable().items():
                t = p.data if hasattr(p, "data") else p
                t = t.tensor() if hasattr(t, "tensor") else t
                if not bool(torch.isfinite(t.float()).all()):
                    bad.append(name)
                else:
                    good.append(name)
            print("first non-finite step", i, "non-finite params:", len(bad), bad[:6], "finite weights:", [g for g in good if g.endswith("_W")], flush=True)
            break
    print("scores", scores, flush=True)


if __name__ == "__main__":
    main()
