#!/usr/bin/env python
"""Achieved HBM bandwidth of the legacy op kernels (native/csrc/legacy_ops.hip) vs the ATen kernels for the same math."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from deeplearning4j_amd.nd4j.ops import legacy as LG  # noqa: E402
from conv_microbench import graph_time  # noqa: E402

ex = LG.executioner()
N = 1 << 26
print("| op | dtype | bytes moved | legacy us | GB/s | ATen us | GB/s |")
print("|---|---|---:|---:|---:|---:|---:|")
for dt in (torch.float32, torch.bfloat16):
    x = torch.rand(N, device="cuda").to(dt) + 0.5
    y = torch.rand(N, device="cuda").to(dt) + 0.5
    z = torch.empty_like(x)
    es = x.element_size()
    m = x.view(8192, -1)
    cases = [
        ("pairwise Add", 3 * N * es, lambda: ex.execPairwiseTransform(LG.op_num("pairwise", "Add"), x, y, z), lambda: torch.add(x, y, out=z)),
        ("pairwise Multiply", 3 * N * es, lambda: ex.execPairwiseTransform(LG.op_num("pairwise", "Multiply"), x, y, z), lambda: torch.mul(x, y, out=z)),
        ("scalar Multiply", 2 * N * es, lambda: ex.execScalar(LG.op_num("scalar", "Multiply"), x, 1.5, z), lambda: torch.mul(x, 1.5, out=z)),
        ("transform Exp", 2 * N * es, lambda: ex.execTransformStrict(LG.op_num("transform_strict", "Exp"), x, z), lambda: torch.exp(x, out=z)),
        ("transform Tanh", 2 * N * es, lambda: ex.execTransformStrict(LG.op_num("transform_strict", "Tanh"), x, z), lambda: torch.tanh(x, out=z)),
        ("transform Abs", 2 * N * es, lambda: ex.execTransformSame(LG.op_num("transform_same", "Abs"), x, z), lambda: torch.abs(x, out=z)),
        ("broadcast Add [8192,8192]+[8192]", 2 * N * es, lambda: ex.execBroadcast(LG.op_num("broadcast", "Add"), m, m[0], None, [1]), lambda: m + m[0]),
        ("reduce Sum (all)", N * es, lambda: ex.execReduceSame(LG.op_num("reduce_same", "Sum"), x), lambda: x.sum()),
        ("reduce Sum dim 1", N * es, lambda: ex.execReduceSame(LG.op_num("reduce_same", "Sum"), m, [1]), lambda: m.sum(1)),
        ("reduce Sum dim 0", N * es, lambda: ex.execReduceSame(LG.op_num("reduce_same", "Sum"), m, [0]), lambda: m.sum(0)),
        ("reduce Norm2 (all)", N * es, lambda: ex.execReduceFloat(LG.op_num("reduce_float", "Norm2"), x), lambda: x.float().norm()),
        ("indexreduce IndexMax dim 1", N * es, lambda: ex.execIndexReduce(LG.op_num("index_reduce", "IndexMax"), m, [1]), lambda: m.argmax(1)),
        ("reduce3 CosineSimilarity (all)", 2 * N * es, lambda: ex.execReduce3(LG.op_num("reduce3", "CosineSimilarity"), x, y),
         lambda: torch.nn.functional.cosine_similarity(x.float(), y.float(), 0)),
    ]
    for name, nbytes, a, b in cases:
        try:
            ta = graph_time(a, reps=5)
        except Exception as e:           # noqa: BLE001
            ta = float("nan")
            print("# %s %s: %s" % (name, dt, str(e)[:100]))
        tb = graph_time(b, reps=5)
        print("| %s | %s | %.0f MB | %.1f | %.0f | %.1f | %.0f |" % (name, str(dt).replace("torch.", ""), nbytes / 1e6, ta, nbytes / ta / 1e3, tb, nbytes / tb / 1e3))
