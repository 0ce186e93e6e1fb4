"""RESOLVED (round 1) — bisection script kept as evidence. Root cause: hipMemsetAsync memset nodes inside captured HIP graphs
(loss accumulator, split-K wgrad buffers) misbehave on replay once a second graph exists in the process; all in-capture zero fills
are kernels now (softmax_xent.hip zero_f32_kernel, conv_igemm.hip zero_f32x4_kernel).

Original notes: two HIP-graph-captured networks in ONE process.

Net `a` captured, then net `b` captured: b's second and later replays read garbage (first bad tensor: output of the first
stride-2 1x1 igemm conv), even if `a` is never replayed again. One captured net per process (the bench / DP layout: one
process per GPU, one model) is unaffected, as is an eager net next to a captured one. Ruled out so far: shared workspace
caches (CLEARWS=1), capture stream sharing, torch.cuda.empty_cache (NOEMPTY=1), replay concurrency (MIDSYNC=1), pool id clash,
the fusion planner (NOFUSE=1), the HIP conv kernels (DL4J_AMD_CONV=library). Observation: with lr 0 the second net's data loss
is off by the SAME constant (+394.54) for every seed while its parameters stay bit-identical to the first net's; a 6-layer
conv/BN graph captured twice does not reproduce it.
Env knobs: NOSEG=1 single-graph capture for b, AEAGER=1 keep a eager, ASTOP=1 stop replaying a after capture.
"""
import torch, sys
sys.path.insert(0, ".")
from deeplearning4j_amd.models.resnet50 import ResNet50
from deeplearning4j_amd.parallel.comm import GradientBuckets
from deeplearning4j_amd.nd4j.learning import Sgd
dev = torch.device("cuda")
import os
if os.environ.get('NOEMPTY') == '1':
    torch.cuda.empty_cache = lambda: None
    import torch.cuda.graphs as _g
    _g.torch.cuda.empty_cache = torch.cuda.empty_cache
m = ResNet50(numClasses=5, inputShape=(3, 160, 160), computeDataType="BFLOAT16", updater=Sgd(0.0))
a = m.init(device=dev); b = m.init(device=dev)
if os.environ.get('NOFUSE') == '1':
    for n_ in (a, b):
        n_.fusion_enabled = False
        n_._plan = n_._build_plan()
import os
if os.environ.get('AEAGER') != '1':
    a.enableGraphCapture(True)
b.enableGraphCapture(True)
bk = GradientBuckets(b._grads, int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 8 << 20)
import os
if os.environ.get('NOSEG') != '1':
    b._dp_buckets = bk; b._grad_sync = bk.reduce_all
x = torch.rand(8, 3, 160, 160, device=dev)
y = torch.zeros(8, 5, device=dev); y[torch.arange(8), torch.arange(8) % 5] = 1
for it in range(6):
    if not (os.environ.get('ASTOP') == '1' and it >= 4):
        a.fit(x, y)
    if os.environ.get('CLEARWS') == '1':
        from deeplearning4j_amd.ops import batchnorm as _B, conv_igemm as _C
        _B._WS.clear(); _C._WS.clear()
    if os.environ.get('MIDSYNC') == '1':
        torch.cuda.synchronize()
    b.fit(x, y)
    torch.cuda.synchronize()
    if it == 3 and a._graph_step is not None:
        print("pools", a._graph_step.graph_a.pool(), a._graph_step.graph_b.pool(), b._graph_step.graph_a.pool(), b._graph_step.graph_b.pool())
    print("   cparams nan a/b", int(torch.isnan(a._cparams_flat.float()).sum()), int(torch.isnan(b._cparams_flat.float()).sum()),
          "cparams a==b", bool(torch.equal(a._cparams_flat, b._cparams_flat)), "params a==b", float((a.params()-b.params()).abs().max()))
    print("   raw _score a/b", float(a._score), float(b._score), "reg a/b", float(a.calcRegularizationScore()), float(b.calcRegularizationScore()),
          "score ptr a/b", a._score.data_ptr(), b._score.data_ptr())
    ga, gb = a._grads, b._grads
    print(it, "score", a.score(), b.score(), "nan a/b", int(torch.isnan(a.params()).sum()), int(torch.isnan(b.params()).sum()),
          "segs", None if b._graph_step.segments is None else [nb for _, nb in b._graph_step.segments])
    if b._graph_step.segments is not None:
        d = (ga - gb).abs()
        bad = torch.nonzero(torch.isnan(gb).reshape(-1) | (d.reshape(-1) > 1e-1 * ga.abs().max()))
        for (s0, e0) in bk.ranges:
            print("   bucket", s0, "rel", float((ga[s0:e0] - gb[s0:e0]).norm() / ga[s0:e0].norm().clamp_min(1e-20)), "nan", int(torch.isnan(gb[s0:e0]).sum()))
        print("  grads rel", float((ga - gb).norm() / ga.norm()), "first bad", bad[:3].reshape(-1).tolist(), "count", bad.numel(), "ranges", bk.ranges[:3])
print("---- first NaN tensors in b (plan order)")
cnt = 0
for k, nd in enumerate(b._plan):
    impls = [nd.impl] + ([nd.bn2] if getattr(nd, "bn2", None) is not None else [])
    for impl in impls:
        for an, av in vars(impl).items():
            if isinstance(av, torch.Tensor) and av.is_floating_point() and av.numel() > 0 and av.is_cuda:
                n = int(torch.isnan(av.float()).sum())
                if n:
                    print(k, nd.out, type(impl).__name__, an, tuple(av.shape), "nan", n)
                    cnt += 1
            elif isinstance(av, tuple) and av and all(isinstance(t, torch.Tensor) for t in av):
                for j, t in enumerate(av):
                    n = int(torch.isnan(t.float()).sum())
                    if n:
                        print(k, nd.out, type(impl).__name__, an, j, tuple(t.shape), "nan", n); cnt += 1
    if cnt > 12:
        break
print("---- first differing saved tensors a vs b (plan order)")
cnt = 0
for k, (na, nb_) in enumerate(zip(a._plan, b._plan)):
    ia = [na.impl] + ([na.bn2] if getattr(na, "bn2", None) is not None else [])
    ib = [nb_.impl] + ([nb_.bn2] if getattr(nb_, "bn2", None) is not None else [])
    for xa, xb in zip(ia, ib):
        for an, av in vars(xa).items():
            bv = getattr(xb, an, None)
            if isinstance(av, torch.Tensor) and isinstance(bv, torch.Tensor) and av.shape == bv.shape and av.is_cuda and av.is_floating_point() and av.numel():
                d = float((av.float() - bv.float()).abs().max())
                ref = float(av.float().abs().max())
                if d > 1e-2 * max(ref, 1e-6) or d != d:
                    print(k, na.out, type(xa).__name__, an, tuple(av.shape), "maxdiff", d, "ref", ref)
                    cnt += 1
    if cnt > 8:
        break
print("---- zero-input hypothesis")
gs = b._graph_step
print("staging _x equals x:", bool(torch.equal(gs._x.float().reshape(-1)[:1000], b._prep_features(x).float().reshape(-1)[:1000])), gs._x.dtype, tuple(gs._x.shape))
b._graph_step = None
for name, xin in (("x", x), ("zeros", torch.zeros_like(x)), ("ones", torch.ones_like(x))):
    b.computeGradientAndScore([xin], [y])
    print("eager b score on", name, float(b._score))
