#!/usr/bin/env python
"""Inference / serving throughput of the zoo ResNet-50 on one MI355X (bf16): direct ComputationGraph.output at batch 256 and 32, and
ParallelInference in BATCHED mode fed by concurrent single-image requests (reference: ParallelInference.java + BatchedInferenceObservable)."""
import json
import sys
import threading
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.models.resnet50 import ResNet50  # noqa: E402
from deeplearning4j_amd.parallel.parallel_inference import ParallelInference, InferenceMode  # noqa: E402


def main():
    dev = torch.device("cuda")
    net = ResNet50(numClasses=1000, inputShape=(3, 224, 224), computeDataType="BFLOAT16").init(device=dev)
    res = {}
    for mode in ("eager", "hip_graph"):
      if mode == "hip_graph":
        net.enableInferenceCapture(True)
      for bsz in (256, 32, 1):
        x = torch.randn(bsz, 3, 224, 224, device=dev)
        for _ in range(4):
            net.output(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            net.output(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res["%s_batch%d_img_per_s" % (mode, bsz)] = round(bsz / dt, 1)
        res["%s_batch%d_ms" % (mode, bsz)] = round(dt * 1e3, 2)
    net.enableInferenceCapture(False)
    pi = ParallelInference(net, workers=1, inferenceMode=InferenceMode.BATCHED, batchLimit=64, queueLimit=256)
    xs = [torch.randn(1, 3, 224, 224, device=dev) for _ in range(16)]
    for k in range(8):
        pi.output(xs[k])
    nthreads, per = 16, 40
    def client(i):
        for k in range(per):
            pi.output(xs[(i + k) % len(xs)])
    th = [threading.Thread(target=client, args=(i,)) for i in range(nthreads)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["parallel_inference_batched_req_per_s"] = round(nthreads * per / dt, 1)
    res["parallel_inference_clients"] = nthreads
    pi.shutdown() if hasattr(pi, "shutdown") else None
    res.update(model="ResNet-50 zoo", dtype="bf16", data="synthetic")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
