"""GPU numerics tests: every hand-written HIP kernel vs a plain PyTorch fp32 reference of the same op."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def test_native_lib_loaded():
    from deeplearning4j_amd.native import lib
    assert lib.hip() is not None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C", [(4096, 64), (1000, 256), (777, 24), (50, 2048), (3000, 7)])
def test_batchnorm_forward_backward(dtype, M, C):
    from deeplearning4j_amd.ops import batchnorm as BN
    torch.manual_seed(0)
    x = (torch.randn(M, C, device=dev()) * 2 + 0.5).to(dtype)
    res = torch.randn(M, C, device=dev()).to(dtype)
    dy = torch.randn(M, C, device=dev()).to(dtype)
    gamma = torch.rand(C, device=dev()) + 0.5
    beta = torch.randn(C, device=dev())
    gmean = torch.randn(C, device=dev()) * 0.1
    gvar = torch.rand(C, device=dev()) * 0.2
    for log_std in (False, True):
        dgm, dgv = torch.empty(C, device=dev()), torch.empty(C, device=dev())
        mean, invstd, scale, shift = BN.batch_stats(x, gamma, beta, 1e-5, 0.9, gmean, gvar, log_std, None, dgm, dgv)
        xc = x.float().cpu()
        rm, rv = xc.mean(0), xc.var(0, unbiased=False)
        tol = 1e-4 if dtype == torch.float32 else 2e-3
        assert torch.allclose(mean.cpu(), rm, atol=tol, rtol=tol)
        assert torch.allclose(invstd.cpu(), torch.rsqrt(rv + 1e-5), atol=tol, rtol=1e-3)
        rdm, rdv = BN.running_stat_grads(rm, rv, gmean.cpu(), gvar.cpu(), 0.9, log_std)
        assert torch.allclose(dgm.cpu(), rdm, atol=1e-3, rtol=1e-3)
        assert torch.allclose(dgv.cpu(), rdv, atol=1e-3, rtol=1e-3)
    y = BN.apply(x, scale, shift, res, None, None, True)
    yr = torch.relu(x.float() * scale + shift + res.float())
    atol = 1e-5 if dtype == torch.float32 else 3e-2
    assert torch.allclose(y.float(), yr, atol=atol, rtol=1e-2)
    # dual-BN residual
    y2 = BN.apply(x, scale, shift, res, scale * 0.5, shift + 1.0, False)
    y2r = x.float() * scale + shift + (res.float() * scale * 0.5 + shift + 1.0)
    assert torch.allclose(y2.float(), y2r, atol=atol * 2, rtol=1e-2)
    # backward vs CPU reference implementation of the same math
    dg, db = torch.empty(C, device=dev()), torch.empty(C, device=dev())
    dx, dz = BN.backward(x, dy, y, mean, invstd, gamma, dg, db, want_dz=True)
    dgr, dbr = torch.empty(C), torch.empty(C)
    dxr, dzr = BN.backward(x.float().cpu(), dy.float().cpu(), y.float().cpu(), mean.cpu(), invstd.cpu(), gamma.cpu(),
                           dgr, dbr, want_dz=True)
    t = 1e-3 if dtype == torch.float32 else 5e-2
    assert torch.allclose(db.cpu(), dbr, atol=t * (M ** 0.5), rtol=2e-2)
    assert torch.allclose(dg.cpu(), dgr, atol=t * (M ** 0.5), rtol=2e-2)
    assert torch.allclose(dz.float().cpu(), dzr, atol=1e-6, rtol=1e-6)
    assert torch.allclose(dx.float().cpu(), dxr, atol=t, rtol=5e-2)
    # ReLU mask recomputed from x (no residual): must equal the result obtained with the stored output as mask
    y1 = BN.apply(x, scale, shift, None, None, None, True)
    dg1, db1, dg2, db2 = (torch.empty(C, device=dev()) for _ in range(4))
    dxa, _ = BN.backward(x, dy, y1, mean, invstd, gamma, dg1, db1)
    dxb, dzb = BN.backward(x, dy, None, mean, invstd, gamma, dg2, db2, want_dz=True, mask_coeffs=(scale, shift))
    assert torch.equal(dxa, dxb) and torch.equal(dg1, dg2) and torch.equal(db1, db2)
    assert torch.equal(dzb, torch.where(y1 > 0, dy, torch.zeros_like(dy)))
    # ReLU mask of a residual block kept as bits (one byte per channel vector): identical results to the stored-output mask
    ym, bits = BN.apply(x, scale, shift, res, None, None, True, want_mask=True)
    assert bits is not None and bits.dtype == torch.uint8 and torch.equal(ym, y)
    dg3, db3 = torch.empty(C, device=dev()), torch.empty(C, device=dev())
    dxm, dzm = BN.backward(x, dy, None, mean, invstd, gamma, dg3, db3, want_dz=True, mask_bits=bits)
    assert torch.equal(dxm, dx) and torch.equal(dzm, dz) and torch.equal(dg3, dg) and torch.equal(db3, db)


@pytest.mark.parametrize("kind", ["NoOp", "Sgd", "Nesterovs", "Adam", "RmsProp", "AdaGrad", "AdaDelta", "AdaMax", "Nadam",
                                  "AMSGrad", "AdaBelief"])
def test_segmented_updater_matches_reference(kind):
    from deeplearning4j_amd.ops import updater as U
    torch.manual_seed(1)
    n1, n2 = 5000, 3333
    n = n1 + n2
    k = U.KIND[kind]
    mult = U.STATE_MULT[k]
    segs = [U.Segment(0, n1, 0, k, 0.01, 0.9, 0.999, 1e-8, l1=1e-4, l2=1e-3, wd=0.0, divide_by_batch=True),
            U.Segment(n1, n2, mult * n1, k, 0.02, 0.8, 0.99, 1e-6, l1=0.0, l2=0.0, wd=1e-2, divide_by_batch=False)]
    p = torch.randn(n, device=dev())
    st = torch.rand(max(mult * n, 1), device=dev()) * 0.1
    pr, sr = p.cpu().clone(), st.cpu().clone()
    su = U.SegmentedUpdater(segs, dev())
    for it in range(3):
        g = torch.randn(n, device=dev())
        gr = g.cpu().clone()
        su.step(p, g, st, 16)
        for s in segs:
            U.apply_segment_reference(s, pr, gr, sr, 1.0 / 16, float(it + 1))
    assert torch.allclose(p.cpu(), pr, atol=2e-5, rtol=1e-4)
    if mult:
        assert torch.allclose(st.cpu()[:mult * n], sr[:mult * n], atol=2e-5, rtol=1e-4)
    assert int(su.iter_dev.item()) == 3


def test_cast_table_permute():
    from deeplearning4j_amd.ops import updater as U
    O, I, kh, kw = 6, 5, 3, 3
    src = torch.randn(100 + O * I * kh * kw + 40, device=dev())
    ct = U.CastTable([(0, 0, 100, 1, 1), (100, 104, O * I * kh * kw, I, kh * kw)], dev())
    dst = torch.zeros(104 + O * I * kh * kw, dtype=torch.bfloat16, device=dev())
    ct.run(src, dst)
    assert torch.equal(dst[:100], src[:100].bfloat16())
    w = src[100:100 + O * I * kh * kw].view(O, I, kh, kw)
    ohwi = dst[104:].view(O, kh, kw, I).permute(0, 3, 1, 2)
    assert torch.equal(ohwi, w.bfloat16())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N", [(256, 1000), (33, 10), (7, 4099)])
def test_softmax_xent(dtype, M, N):
    from deeplearning4j_amd.ops import softmax as S
    torch.manual_seed(2)
    z = (torch.randn(M, N, device=dev()) * 3).to(dtype)
    idx = torch.randint(0, N, (M,), device=dev())
    y = torch.zeros(M, N, device=dev(), dtype=dtype)
    y[torch.arange(M), idx] = 1
    score, grad, probs = S.softmax_xent(z, y, want_probs=True)
    p = torch.softmax(z.float(), 1)
    ref_score = -(y.float() * torch.log(p.clamp(1e-10, 1))).sum()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert torch.allclose(probs.float(), p, atol=tol)
    assert torch.allclose(grad.float(), p - y.float(), atol=tol)
    assert abs(float(score) - float(ref_score)) < 1e-3 * max(1.0, abs(float(ref_score)))
    s2, g2, _ = S.softmax_xent(z, None, idx)
    assert abs(float(s2) - float(ref_score)) < 1e-3 * max(1.0, abs(float(ref_score)))
    assert torch.allclose(g2.float(), p - y.float(), atol=tol)
    assert torch.allclose(S.softmax(z).float(), p, atol=tol)
    assert torch.allclose(S.softmax(z, log=True).float(), torch.log_softmax(z.float(), 1), atol=tol * 10, rtol=1e-2)
    dy = torch.randn(M, N, device=dev()).to(dtype)
    ref = p * (dy.float() - (dy.float() * p).sum(1, keepdim=True))
    assert torch.allclose(S.softmax_bp(probs, dy).float(), ref, atol=tol * 5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["max", "avg"])
@pytest.mark.parametrize("cfg", [((2, 16, 13, 13), (3, 3), (2, 2), (0, 0)), ((3, 8, 9, 10), (2, 2), (2, 2), (1, 1)),
                                 ((2, 5, 8, 8), (3, 3), (1, 1), (1, 1)), ((2, 64, 28, 28), (3, 3), (2, 2), (0, 0))])
def test_pool2d(dtype, mode, cfg):
    import torch.nn.functional as F
    from deeplearning4j_amd.ops import pooling as P
    shape, k, s, p = cfg
    torch.manual_seed(3)
    x = torch.randn(shape, device=dev()).to(dtype).contiguous(memory_format=torch.channels_last)
    y, ctx = P.pool2d_forward(x, mode, k, s, p, train=True)
    assert ctx.geom is not None, "HIP pooling path not taken"
    # reference on the CPU (plain PyTorch fp32, NCHW-contiguous)
    xf = x.float().cpu().contiguous().requires_grad_(True)
    yr = F.max_pool2d(xf, k, s, p) if mode == "max" else F.avg_pool2d(xf, k, s, p, count_include_pad=True)
    assert torch.allclose(y.float().cpu(), yr, atol=1e-6 if dtype == torch.float32 else 2e-2)
    dy = torch.randn_like(yr).to(dtype)
    dx = P.pool2d_backward(dy.to(dev()), ctx)
    (dxr,) = torch.autograd.grad(yr, xf, dy.float())
    assert torch.allclose(dx.float().cpu(), dxr, atol=1e-5 if dtype == torch.float32 else 3e-2)


def test_resnet_block_gpu_matches_cpu_fp32():
    """Fused BN/add/ReLU graph on the GPU (HIP kernels) vs the same network on CPU (PyTorch math)."""
    from tests.test_networks_cpu import _onehot
    from deeplearning4j_amd.models.resnet50 import ResNet50
    m = ResNet50(numClasses=5, inputShape=(3, 160, 160))
    a = m.init(device="cpu")
    b = m.init(device=dev())
    assert torch.equal(a.params(), b.params().cpu())
    x = torch.rand(2, 3, 160, 160)
    y = _onehot(2, 5).float()
    a.computeGradientAndScore([x], [y])
    b.computeGradientAndScore([x.to(dev())], [y.to(dev())])
    assert abs(float(a._score) - float(b._score)) < 2e-2 * max(1.0, abs(float(a._score)))
    ga, gb = a._grads, b._grads.cpu()
    rel = (ga - gb).norm() / ga.norm().clamp_min(1e-12)
    assert rel < 5e-2, "relative gradient error %g" % float(rel)


def test_lenet_bf16_graph_capture_trains():
    from deeplearning4j_amd.models.lenet import LeNet
    from deeplearning4j_amd.nd4j.learning import Adam
    net = LeNet(inputShape=(1, 28, 28), numClasses=10, updater=Adam(1e-3), computeDataType="BFLOAT16").init(device=dev())
    net.enableGraphCapture(True)
    torch.manual_seed(0)
    x = torch.rand(64, 784, device=dev())
    lab = torch.randint(0, 10, (64,), device=dev())
    y = torch.zeros(64, 10, device=dev())
    y[torch.arange(64), lab] = 1
    net.fit(x, y)
    s0 = net.score()
    for _ in range(60):
        net.fit(x, y)
    assert net._graph_step.graph_a is not None
    assert net.score() < 0.5 * s0
    assert net.getIterationCount() == 61


def test_segmented_graph_capture_matches_eager(monkeypatch):
    """DP overlap path: forward+backward captured as a chain of graphs cut at gradient-bucket boundaries
    (world size 1, so the bucket launches are no-ops) must produce the gradients of the eager step, replay after replay."""
    from tests.test_networks_cpu import _onehot
    # one rank normally gets no cuts (nothing to overlap with): rehearse the cut path itself, two groups of buckets
    monkeypatch.setenv("DL4J_AMD_DP_CUTS_AT_ONE_RANK", "1")
    monkeypatch.setenv("DL4J_AMD_DP_MAX_CUTS", "2")
    from deeplearning4j_amd.models.resnet50 import ResNet50
    from deeplearning4j_amd.parallel.comm import GradientBuckets
    from deeplearning4j_amd.nd4j.learning import Sgd
    # lr 0: weights stay put, so every iteration must reproduce the same gradient (BN running-stat slots aside)
    m = ResNet50(numClasses=5, inputShape=(3, 160, 160), computeDataType="BFLOAT16", updater=Sgd(0.0), weightInit="RELU")
    a = m.init(device=dev())            # eager reference
    b = m.init(device=dev())
    _well_conditioned(a, b)      # "same gradient" only means something at a well-conditioned start (see _well_conditioned)
    b.enableGraphCapture(True)
    buckets = GradientBuckets(b._grads, 8 << 20)
    launched = []
    orig = buckets._launch
    buckets._launch = lambda k: (launched.append(k), orig(k))[1]
    b._dp_buckets = buckets
    b._grad_sync = buckets.reduce_all
    x = torch.rand(8, 3, 160, 160, device=dev())
    y = _onehot(8, 5).float().to(dev())
    wmask = torch.zeros_like(a._grads, dtype=torch.bool)       # weight / bias / gamma / beta slots (running stats drift with decay)
    for e in a._entries:
        if e.spec.role != "stat":
            wmask[e.off:e.off + e.spec.numel] = True
    for it in range(7):
        # the eager reference takes the same (lr = 0) steps: its running statistics — the pivots of the conv statistics epilogues —
        # then match the captured network's, so both round the batch statistics the same way
        a.fit(x, y)
        g_ref, s_ref = a._grads.clone(), float(a._score)
        b.fit(x, y)
        if it >= 3:
            assert b._graph_step.segments is not None
            rel = ((b._grads - g_ref)[wmask]).norm() / g_ref[wmask].norm()
            assert rel < 2e-2, (it, float(rel))
            assert abs(float(b._score) - s_ref) < 1e-2 * abs(s_ref)
    segs = b._graph_step.segments
    assert len(segs) == 3, "capture was not cut into two bucket groups + the remainder"
    assert sum(nb for _, nb in segs) == len(buckets.ranges) - 1
    # every bucket launched exactly once per replayed iteration, tail first
    per_iter = launched[-len(buckets.ranges):]
    assert per_iter == list(range(len(buckets.ranges) - 1, -1, -1))


def test_two_captured_networks_in_one_process():
    """Regression: memset NODES in captured HIP graphs misbehaved once a second graph existed (score / split-K wgrad buffers were
    not re-zeroed on replay). All zero-fills inside captured regions are kernels now; two captured nets must stay identical."""
    from tests.test_networks_cpu import _onehot
    from deeplearning4j_amd.models.resnet50 import ResNet50
    from deeplearning4j_amd.nd4j.learning import Sgd
    m = ResNet50(numClasses=5, inputShape=(3, 160, 160), computeDataType="BFLOAT16", updater=Sgd(0.0))
    a = m.init(device=dev())
    b = m.init(device=dev())
    _well_conditioned(a, b)      # the untrained zoo init is chaotic (see _well_conditioned): rounding-level changes move its loss by 20 %
    a.enableGraphCapture(True)
    b.enableGraphCapture(True)
    x = torch.rand(8, 3, 160, 160, device=dev())
    y = _onehot(8, 5).float().to(dev())
    scores = []
    for it in range(7):
        a.fit(x, y)
        b.fit(x, y)
        junk = [torch.randn(1 << 20, device=dev()) for _ in range(4)]      # churn the general pool between replays
        scores.append((float(a._score), float(b._score)))
    assert a._graph_step.graph_a is not None and b._graph_step.graph_a is not None
    s0 = scores[0][0]
    for sa, sb in scores:
        # the two networks go through identical arithmetic every iteration ...
        assert abs(sa - sb) < 1e-4 * abs(s0), scores
        # ... and across iterations only the statistics pivots (running means, lr = 0) move: rounding-level differences
        assert abs(sa - s0) < 2e-2 * abs(s0), scores
    rel = (a._grads - b._grads).norm() / a._grads.norm()
    assert rel < 1e-3, float(rel)


def test_inference_graph_capture_matches_eager_and_parallel_inference_pads_batches():
    """enableInferenceCapture: output() replayed as a HIP graph equals the eager forward for several batch sizes, also after a
    training step changed the weights; ParallelInference (BATCHED) serves odd batch sizes through power-of-two padded graphs."""
    from deeplearning4j_amd.models.resnet50 import ResNet50
    from deeplearning4j_amd.parallel.parallel_inference import ParallelInference, InferenceMode
    torch.manual_seed(0)
    net = ResNet50(numClasses=10, inputShape=(3, 160, 160), computeDataType="BFLOAT16").init(device=dev())
    xs = {b: torch.randn(b, 3, 160, 160, device=dev()) for b in (4, 7)}
    ref = {b: net.output(x)[0].clone() for b, x in xs.items()}
    net.enableInferenceCapture(True)
    for _ in range(4):
        for b, x in xs.items():
            y = net.output(x)[0]
            assert torch.allclose(y, ref[b], atol=2e-3, rtol=2e-2), float((y - ref[b]).abs().max())
    assert sum(e["graph"] is not None for e in net._infer_cap.entries.values()) == 2
    # weights change between calls: the captured graph reads the refreshed parameters (LeNet, MultiLayerNetwork path)
    from deeplearning4j_amd.models.lenet import LeNet
    ln = LeNet(numClasses=10, inputShape=(1, 28, 28)).init(device=dev())
    xl = torch.randn(16, 1, 28, 28, device=dev())
    yl = torch.nn.functional.one_hot(torch.arange(16) % 10, 10).float().to(dev())
    before = ln.output(xl).clone()
    ln.enableInferenceCapture(True)
    for _ in range(4):
        assert torch.allclose(ln.output(xl), before, atol=1e-5)
    for _ in range(3):
        ln.fit(xl, yl)
    got = ln.output(xl).clone()
    ln.enableInferenceCapture(False)
    after = ln.output(xl)
    assert torch.allclose(got, after, atol=1e-5) and not torch.allclose(after, before, atol=1e-4)
    net.enableInferenceCapture(False)
    pi = ParallelInference(net, workers=1, inferenceMode=InferenceMode.BATCHED, batchLimit=8)
    try:
        for _ in range(5):
            out = pi.output(xs[7].cpu())
            out = out[0] if isinstance(out, (list, tuple)) else out
            assert tuple(out.shape) == (7, 10)
            assert torch.allclose(out.to(dev()).float(), net.output(xs[7])[0].float(), atol=2e-3, rtol=2e-2)
    finally:
        pi.shutdown()


def test_wgrad_side_stream_matches_main_stream_eager_and_captured():
    """ops/streams.py (opt-in): conv wgrads on a second HIP stream — forked per layer, joined when backward ends — give the same
    gradients as the single-stream path, eagerly and as a parallel branch inside a captured HIP graph."""
    from tests.test_networks_cpu import _onehot
    from deeplearning4j_amd.models.resnet50 import ResNet50
    from deeplearning4j_amd.nd4j.learning import Sgd
    from deeplearning4j_amd.ops import streams
    m = ResNet50(numClasses=5, inputShape=(3, 160, 160), computeDataType="BFLOAT16", updater=Sgd(0.0))
    x = torch.rand(8, 3, 160, 160, device=dev())
    y = _onehot(8, 5).float().to(dev())
    ref = m.init(device=dev())
    _well_conditioned(ref)
    p0 = ref.params().clone()
    ref.computeGradientAndScore([x], [y])
    torch.cuda.synchronize()
    g_ref = ref._grads.clone()
    wmask = torch.zeros_like(g_ref, dtype=torch.bool)
    for e in ref._entries:
        if e.spec.role != "stat":
            wmask[e.off:e.off + e.spec.numel] = True
    # single-stream reference after the same six (lr = 0) steps the captured network takes: same running statistics, hence the
    # same statistics pivots in the conv epilogues
    for it in range(6):
        ref.fit(x, y)
    torch.cuda.synchronize()
    g_ref6 = ref._grads.clone()
    streams.set_enabled(True)
    try:
        a = m.init(device=dev())
        a.setParams(p0)
        a.computeGradientAndScore([x], [y])
        assert streams._side is not None and not streams._pending and not streams._held      # forked, and joined at scope exit
        torch.cuda.synchronize()
        rel = ((a._grads - g_ref)[wmask]).norm() / g_ref[wmask].norm()
        assert rel < 1e-3, float(rel)
        b = m.init(device=dev())
        b.setParams(p0)
        b.enableGraphCapture(True)
        for it in range(6):
            b.fit(x, y)
        assert b._graph_step.graph_a is not None
        torch.cuda.synchronize()
        rel = ((b._grads - g_ref6)[wmask]).norm() / g_ref6[wmask].norm()
        assert rel < 5e-3, float(rel)
    finally:
        streams.set_enabled(False)


def _well_conditioned(*nets):
    """A start point where gradients mean something numerically. The untrained 53-layer graph (He or the zoo's TruncatedNormal init,
    all gammas 1, logits of magnitude ~10) is chaotic: tools/debug_bf16_grads.py shows that perturbing the fp32 parameters ONCE by
    bf16-sized relative noise already drops the fp32-vs-fp32 gradient cosine to 0.42 (profiles/r2_bf16_grad_conditioning.txt), so no
    reduced-precision kernel could agree with fp32 there. Scaling the last BatchNorm gamma of every residual branch to 0.2 and the
    classifier weights by 0.05 (the usual "small residual branch" start) makes the same perturbation give cosine 0.9997."""
    p = nets[0].params().clone().view(-1)
    for e in nets[0]._entries:
        name = str(nets[0]._layer_key(e.layer_idx))
        if e.spec.key == "gamma" and name.endswith("branch2c"):
            p[e.off:e.off + e.spec.numel] = 0.2
        if name == "output" and e.spec.key == "W":
            p[e.off:e.off + e.spec.numel] *= 0.05
    for n in nets:
        n.setParams(p.to(n.params().device))


def test_resnet50_bf16_gpu_gradients_match_fp32_cpu():
    """Full zoo ResNet-50 graph: bf16 compute on the GPU (own igemm / BN / pooling kernels) against the fp32 CPU network, same
    parameters and batch: score within 1 %, flat gradient cosine > 0.995. Per layer the bound is RELATIVE TO A FLOOR measured in the
    same test: the fp32 CPU network with only its weights and input rounded to bf16 (activations and all arithmetic fp32 — the least
    any bf16 implementation perturbs) already drops the worst per-layer weight-gradient cosine to 0.94 on this 53-layer graph at
    batch 8, so "0.95 everywhere" is not reachable in bf16 at all; the GPU pipeline (activations bf16 as well) must stay within 0.08
    of that floor on every layer and above 0.85 absolutely (measured: 0.88 worst, tools/debug_layer_cosines.py,
    profiles/r3_bf16_layer_cosines.md)."""
    from tests.test_networks_cpu import _onehot
    from deeplearning4j_amd.models.resnet50 import ResNet50
    a = ResNet50(numClasses=10, inputShape=(3, 160, 160), weightInit="RELU").init(device="cpu")
    b = ResNet50(numClasses=10, inputShape=(3, 160, 160), weightInit="RELU", computeDataType="BFLOAT16").init(device=dev())
    _well_conditioned(a, b)
    assert torch.equal(a.params(), b.params().cpu())
    x = torch.rand(8, 3, 160, 160)
    y = _onehot(8, 10).float()
    a.computeGradientAndScore([x], [y])
    b.computeGradientAndScore([x.to(dev())], [y.to(dev())])
    sa, sb = float(a._score), float(b._score)
    assert abs(sa - sb) < 1e-2 * max(1.0, abs(sa)), (sa, sb)
    ga, gb = a._grads.double(), b._grads.cpu().double()
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()).clamp_min(1e-30))
    print("bf16 vs fp32 gradient cosine %.5f, norm ratio %.4f" % (cos, float(gb.norm() / ga.norm())))
    assert cos > 0.995, cos
    assert 0.97 < float(gb.norm() / ga.norm()) < 1.03
    # the floor: same fp32 CPU network, weights and input rounded to bf16 once
    a2 = ResNet50(numClasses=10, inputShape=(3, 160, 160), weightInit="RELU").init(device="cpu")
    a2.setParams(a.params().clone().view(-1).bfloat16().float())
    a2.computeGradientAndScore([x.bfloat16().float()], [y])
    g2 = a2._grads.double()

    def _cos(u, v):
        return float((u * v).sum() / (u.norm() * v.norm()).clamp_min(1e-30))
    worst, worst_gap = 1.0, 0.0
    for e in a._entries:
        if e.spec.key == "W":
            sl = slice(e.off, e.off + e.spec.numel)
            c_gpu, c_floor = _cos(ga[sl], gb[sl]), _cos(ga[sl], g2[sl])
            worst, worst_gap = min(worst, c_gpu), max(worst_gap, c_floor - c_gpu)
            assert c_gpu > c_floor - 0.08, (str(a._layer_key(e.layer_idx)), c_gpu, c_floor)
    print("worst per-layer weight-gradient cosine %.4f, largest gap to the bf16-rounding floor %.4f" % (worst, worst_gap))
    assert worst > 0.85, worst


@pytest.mark.parametrize("cdt", [None, "HALF"])
def test_resnet50_fp32_and_fp16_run_on_own_conv_kernels(cdt, monkeypatch):
    """FLOAT network (DL4J's default dtype) and FLOAT master + HALF compute: every convolution except the 3-channel stem runs on
    the in-tree kernels (the library entry points are poisoned after the stem) and the gradients match the fp32 CPU network."""
    import torch.nn.functional as F
    from tests.test_networks_cpu import _onehot
    from deeplearning4j_amd.models.resnet50 import ResNet50
    from deeplearning4j_amd.ops import conv as CONV
    a = ResNet50(numClasses=5, inputShape=(3, 160, 160), weightInit="RELU").init(device="cpu")
    b = ResNet50(numClasses=5, inputShape=(3, 160, 160), weightInit="RELU", computeDataType=cdt).init(device=dev())
    _well_conditioned(a, b)
    x = torch.rand(8, 3, 160, 160)
    y = _onehot(8, 5).float()
    a.computeGradientAndScore([x], [y])
    real_fwd, real_bwd = F.conv2d, torch.ops.aten.convolution_backward
    calls = {"lib": 0}

    def fwd(inp, w, *r, **k):
        calls["lib"] += 1
        assert w.shape[1] == 3, "library forward convolution on a %s filter" % (tuple(w.shape),)
        return real_fwd(inp, w, *r, **k)

    def bwd(dz, inp, w, *r, **k):
        calls["lib"] += 1
        assert w.shape[1] == 3, "library convolution backward on a %s filter" % (tuple(w.shape),)
        return real_bwd(dz, inp, w, *r, **k)
    monkeypatch.setattr(F, "conv2d", fwd)
    monkeypatch.setattr(CONV.F, "conv2d", fwd)
    monkeypatch.setattr(torch.ops.aten, "convolution_backward", bwd, raising=False)
    b.computeGradientAndScore([x.to(dev())], [y.to(dev())])
    assert calls["lib"] <= 2
    sa, sb = float(a._score), float(b._score)
    assert abs(sa - sb) < (2e-2 if cdt else 1e-3) * max(1.0, abs(sa)), (sa, sb)
    ga, gb = a._grads.double(), b._grads.cpu().double()
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()).clamp_min(1e-30))
    print("%s vs fp32 CPU gradient cosine %.6f" % (cdt or "FLOAT", cos))
    # fp32: the kernels are exact to ~1e-5; what remains are single ReLU-mask flips at |y| ~ 1e-7 (tools/debug_fp32_bn.py found
    # one in 51200 elements of bn4f_branch2a), each worth ~3e-3 relative error downstream
    assert cos > (0.999 if cdt else 0.9999), cos
    # BatchNorm running-statistic deltas travel in the gradient vector (written by the forward pass): they must survive on the GPU
    for e in a._entries:
        if e.spec.key in ("mean", "var", "log10stdev"):
            u, v = ga[e.off:e.off + e.spec.numel], gb[e.off:e.off + e.spec.numel]
            assert float(u.norm()) > 0 and float((u - v).norm() / u.norm()) < (2e-2 if cdt else 1e-3), (e.spec.key, e.layer_idx)


@pytest.mark.parametrize("capture", [False, True])
def test_batchnorm_running_statistics_follow_cpu_training(capture):
    """Two fit() steps: the global mean / variance parameters of every BatchNormalization move exactly as on the CPU (they are updated
    through the gradient vector by the NoOp updater, BatchNormalization.java:330-360), eager and with the step captured in a graph."""
    from tests.test_networks_cpu import _onehot
    from deeplearning4j_amd.models.resnet50 import ResNet50
    from deeplearning4j_amd.nd4j.learning import Sgd
    a = ResNet50(numClasses=5, inputShape=(3, 160, 160), weightInit="RELU", updater=Sgd(1e-3)).init(device="cpu")
    b = ResNet50(numClasses=5, inputShape=(3, 160, 160), weightInit="RELU", updater=Sgd(1e-3)).init(device=dev())
    _well_conditioned(a, b)
    if capture:
        b.enableGraphCapture(True)
    x = torch.rand(4, 3, 160, 160)
    y = _onehot(4, 5).float()
    p0 = a.params().clone().view(-1)
    for _ in range(3):
        a.fit([x], [y])
        b.fit([x.to(dev())], [y.to(dev())])
    pa, pb = a.params().view(-1), b.params().cpu().view(-1)
    moved = 0.0
    for e in a._entries:
        if e.spec.key in ("mean", "var", "log10stdev"):
            sl = slice(e.off, e.off + e.spec.numel)
            moved += float((pa[sl] - p0[sl]).norm())
            assert float((pa[sl] - pb[sl]).norm()) < 2e-3 * max(1e-3, float((pa[sl] - p0[sl]).norm())) + 1e-6, (e.spec.key, e.layer_idx)
    assert moved > 1.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N", [8, 1000, 1024, 2056, 4096, 8192, 16384, 20000, 1001])
def test_softmax_rows_register_resident_forms(dtype, N):
    """softmax / log-softmax / softmax backward with the row held in registers (softmax_xent.hip *_vec_kernel, 1-8 16-byte pieces per
    thread) and the three-pass fallback for rows that do not fit or are not a multiple of the vector width; vs fp32 torch."""
    from deeplearning4j_amd.ops import softmax as S
    g = torch.Generator(device="cuda").manual_seed(N)
    z = (torch.randn(37, N, device="cuda", generator=g) * 3).to(dtype)
    dy = torch.randn(37, N, device="cuda", generator=g).to(dtype)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    p = S.softmax(z)
    ref = torch.softmax(z.float(), -1)
    assert (p.float() - ref).abs().max().item() < tol * ref.max().item() + 1e-7
    lp = S.softmax(z, log=True)
    refl = torch.log_softmax(z.float(), -1)
    assert (lp.float() - refl).abs().max().item() < (0.1 if dtype == torch.bfloat16 else 1e-4)
    dz = S.softmax_bp(p, dy)
    pf = p.float()
    refd = pf * (dy.float() - (dy.float() * pf).sum(-1, keepdim=True))
    assert (dz.float() - refd).abs().max().item() < tol * max(refd.abs().max().item(), 1e-3) + 1e-6


def test_fused_input_staging_matches_cast_then_permute():
    from deeplearning4j_amd.ops import spatial as SP
    g = torch.Generator(device="cuda").manual_seed(0)
    for c, dt in ((3, torch.bfloat16), (3, torch.float32), (1, torch.bfloat16), (4, torch.bfloat16)):
        x = torch.randn(5, c, 17, 13, device="cuda", generator=g)
        y = SP.nchw_f32_to_nhwc(x, dt)
        ref = x.to(dt).contiguous(memory_format=torch.channels_last)
        assert y.shape == ref.shape and y.dtype == dt and torch.equal(y, ref)
        assert y.is_contiguous(memory_format=torch.channels_last) or c == 1
    assert SP.nchw_f32_to_nhwc(torch.randn(2, 8, 4, 4, device="cuda"), torch.bfloat16) is None
