"""Declarable-helper HIP kernels (helpers_nn.hip, helpers_data.hip, lstm.hip) vs plain PyTorch fp32 references.
Every platform helper is run through its isUsable predicate (must accept the case) and compared with the generic op."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _kernels_not_cost_model():
    """These tests check the hand-written kernels themselves: the measured cost model (which would route some shapes to ATen) is off."""
    from deeplearning4j_amd.nd4j.ops import cost_model
    old, cost_model.ENABLED = cost_model.ENABLED, False
    yield
    cost_model.ENABLED = old


def dev():
    return torch.device("cuda")


def _platform_vs_generic(name, ins, i=(), t=(), b=(), d=(), atol=1e-5, rtol=1e-4):
    import deeplearning4j_amd.nd4j.ops.custom as c
    c._load_impls()
    gi = [x.to(dev()).contiguous() if isinstance(x, torch.Tensor) else x for x in ins]
    hit = [fn for usable, fn in c._PLATFORM.get(name, ()) if usable(gi, list(i), list(t), list(b), list(d))]
    assert hit, "no platform helper accepted %s" % name
    got = hit[0](gi, list(i), list(t), list(b), list(d))
    ci = [x.float() if isinstance(x, torch.Tensor) and x.is_floating_point() else x for x in ins]
    ref = c._IMPL[name](ci, list(i), list(t), list(b), list(d))
    got = got if isinstance(got, (tuple, list)) else [got]
    ref = ref if isinstance(ref, (tuple, list)) else [ref]
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert tuple(g.shape) == tuple(r.shape), (name, g.shape, r.shape)
        assert torch.allclose(g.float().cpu(), r.float(), atol=atol, rtol=rtol), (name, float((g.float().cpu() - r.float()).abs().max()))
    return got


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_softmax_family(dtype):
    torch.manual_seed(0)
    tol = dict(atol=1e-5, rtol=1e-4) if dtype == torch.float32 else dict(atol=2e-2, rtol=2e-2)
    x = (torch.randn(3, 7, 50) * 2).to(dtype)
    g = torch.randn(3, 7, 50).to(dtype)
    _platform_vs_generic("softmax", [x], [-1], **tol)
    _platform_vs_generic("log_softmax", [x], [2], atol=tol["atol"] * 4, rtol=tol["rtol"])
    _platform_vs_generic("softmax_bp", [x, g], [-1], **tol)
    _platform_vs_generic("log_softmax_bp", [x, g], [-1], atol=tol["atol"] * 4, rtol=tol["rtol"])
    xl = torch.randn(5, 3000).to(dtype)          # long rows: 4-wave path
    _platform_vs_generic("softmax_bp", [xl, torch.randn(5, 3000).to(dtype)], [1], **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(4, 9, 64), (33, 1536)])
def test_layer_norm_fwd_bwd(dtype, shape):
    torch.manual_seed(1)
    tol = dict(atol=2e-5, rtol=1e-4) if dtype == torch.float32 else dict(atol=6e-2, rtol=3e-2)
    x = torch.randn(*shape).to(dtype)
    gain, bias = torch.rand(shape[-1]) + 0.5, torch.randn(shape[-1])
    g = torch.randn(*shape).to(dtype)
    _platform_vs_generic("layer_norm", [x, gain, bias], [len(shape) - 1], [1e-5], **tol)
    _platform_vs_generic("layer_norm", [x, gain], [], **tol)
    bt = dict(atol=tol["atol"] * 8, rtol=tol["rtol"] * 2)
    _platform_vs_generic("layer_norm_bp", [x, gain, bias, g], [len(shape) - 1], **bt)


def test_bias_prelu_lrn_clip_dropout():
    torch.manual_seed(2)
    x4 = torch.randn(2, 6, 5, 7)
    _platform_vs_generic("biasadd", [x4, torch.randn(6)], [], [], [True])
    _platform_vs_generic("biasadd", [torch.randn(8, 11), torch.randn(11)])
    _platform_vs_generic("biasadd", [x4.bfloat16(), torch.randn(7)], atol=3e-2, rtol=2e-2)
    _platform_vs_generic("prelu", [x4, torch.rand(6, 1, 1)])
    _platform_vs_generic("lrn", [x4, ], [2], [1.0, 1e-2, 0.75], atol=1e-5)
    big = torch.randn(1000, 77)
    _platform_vs_generic("clipbynorm", [big], [], [3.0])
    _platform_vs_generic("clipbynorm", [big * 1e-4], [], [3.0])          # below the threshold: unchanged
    # dropout: statistics, mask/value consistency, determinism for a fixed generator state
    from deeplearning4j_amd.ops import helpers as H
    xd = torch.ones(1 << 20, device=dev())
    y, m = H.dropout(xd, 0.8, 1234, 7)
    assert abs(float(m.float().mean()) - 0.8) < 5e-3
    assert torch.equal(y, m.float() / 0.8)
    y2, m2 = H.dropout(xd, 0.8, 1234, 7)
    assert torch.equal(m, m2) and not torch.equal(m, H.dropout(xd, 0.8, 1234, 8)[1])
    g = torch.randn(1 << 20, device=dev())
    assert torch.allclose(H.dropout_bwd(g, m, 0.8), g * m.float() / 0.8)
    import deeplearning4j_amd.nd4j.ops.custom as c
    out = c.OpExecutioner.getInstance()._run("dropout", [xd], [], [0.5], [], [])
    assert abs(float(out[1].mean()) - 0.5) < 5e-3


def test_indexed_ops():
    torch.manual_seed(3)
    table = torch.randn(50, 24)
    idx = torch.randint(0, 50, (4, 9))
    _platform_vs_generic("embedding_lookup", [table, idx])
    _platform_vs_generic("embedding_lookup", [table.bfloat16(), idx.int()], atol=1e-2)
    _platform_vs_generic("gather", [torch.randn(20, 3, 5), torch.randint(0, 20, (7,))], [0])
    _platform_vs_generic("onehot", [torch.randint(0, 10, (6, 5))], [-1, 10], [2.0, -1.0])
    x = torch.randn(64, 3, 4)
    seg = torch.randint(0, 9, (64,))
    _platform_vs_generic("unsorted_segment_sum", [x, seg], [9], atol=1e-4)
    _platform_vs_generic("segment_sum", [x, torch.sort(seg)[0]], atol=1e-4)
    from deeplearning4j_amd.ops import helpers as H
    tbl = torch.zeros(50, 24, device=dev())
    src = torch.randn(36, 24, device=dev())
    H.scatter_add_rows_(tbl, idx.reshape(-1).to(dev()), src, alpha=0.5)
    ref = torch.zeros(50, 24).index_add_(0, idx.reshape(-1), src.cpu() * 0.5)
    assert torch.allclose(tbl.cpu(), ref, atol=1e-5)


def test_scan_merge():
    torch.manual_seed(4)
    x = torch.randn(7, 1000)
    for excl in (0, 1):
        for rev in (0, 1):
            _platform_vs_generic("cumsum", [x], [excl, rev, 1], atol=2e-3)
    _platform_vs_generic("cumsum", [torch.randn(3, 4, 33)], [0, 0, -1], atol=1e-4)
    arrs = [torch.randn(5, 100) for _ in range(4)]
    for n in ("mergeadd", "mergeavg", "mergemax"):
        _platform_vs_generic(n, arrs)


def test_im2col_col2im_upsampling_resize():
    torch.manual_seed(5)
    x = torch.randn(2, 3, 9, 11)
    col = _platform_vs_generic("im2col", [x], [3, 2, 2, 1, 1, 0, 1, 2])[0]
    _platform_vs_generic("col2im", [col.cpu()], [2, 1, 1, 0, 9, 11, 1, 2], atol=1e-4)
    _platform_vs_generic("upsampling2d", [x], [2, 3, 1])
    g = torch.randn(2, 3, 18, 33)
    _platform_vs_generic("upsampling2d_bp", [x, g], [1], atol=1e-4)
    img = torch.rand(2, 7, 9, 3)
    _platform_vs_generic("resize_bilinear", [img], [14, 5], atol=1e-5)
    _platform_vs_generic("resize_bilinear", [img], [13, 17], [], [True, False], atol=1e-5)
    _platform_vs_generic("resize_bilinear", [img], [13, 17], [], [False, True], atol=1e-5)
    _platform_vs_generic("resize_nearest_neighbor", [img], [14, 18])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("peephole", [False, True])
def test_lstm_cell_fused_matches_reference(dtype, peephole):
    from deeplearning4j_amd.ops import lstm as LS
    from deeplearning4j_amd.nd4j.activations import ActivationSigmoid, ActivationTanH
    torch.manual_seed(6)
    mb, n = 17, 96
    gates = torch.randn(mb, 4 * n)
    c_prev = torch.randn(mb, n)
    peep = tuple(torch.randn(n) * 0.5 for _ in range(3)) if peephole else None
    mask = (torch.rand(mb, 1) > 0.3).float()
    act, gact = ActivationTanH(), ActivationSigmoid()
    h_r, c_r, sv_r = LS.cell_forward(gates, c_prev, peep, act, gact, mask, True)
    dh, dc = torch.randn(mb, n), torch.randn(mb, n)
    dg_r, dcp_r, dp_r = LS.cell_backward(dh, dc, c_prev, sv_r, peep, act, gact, mask)
    D = dev()
    gp = tuple(p.to(D) for p in peep) if peep else None
    h, c, sv = LS.cell_forward(gates.to(D, dtype), c_prev.to(D, dtype), gp, act, gact, mask.to(D, dtype), True)
    assert sv[0] == "fast", "fused HIP LSTM cell was not used"
    tol = 1e-5 if dtype == torch.float32 else 4e-2
    assert torch.allclose(h.float().cpu(), h_r, atol=tol) and torch.allclose(c.float().cpu(), c_r, atol=tol)
    dg, dcp, dp = LS.cell_backward(dh.to(D, dtype), dc.to(D, dtype), c_prev.to(D, dtype), sv, gp, act, gact, mask.to(D, dtype))
    assert torch.allclose(dg.float().cpu(), dg_r, atol=tol * 4) and torch.allclose(dcp.float().cpu(), dcp_r, atol=tol * 4)
    if peephole:
        for a, r in zip(dp, dp_r):
            assert torch.allclose(a.cpu(), r, atol=tol * 30, rtol=5e-2)


def test_sg_pairs_step():
    from deeplearning4j_amd.ops import helpers as H
    torch.manual_seed(7)
    V, D, n = 300, 100, 256
    syn0, syn1 = torch.randn(V, D) * 0.1, torch.randn(V, D) * 0.1
    in_idx = torch.arange(n) % 40
    tgt = torch.randperm(V)[:n]                         # distinct output rows: the hogwild update order cannot matter
    tgt[5] = -1                                         # padded slot
    label = (torch.rand(n) > 0.5).float()
    lr = 0.05
    s1 = syn1.clone().to(dev())
    dh = H.sg_pairs_(syn0.to(dev()), s1, in_idx.to(dev()), tgt.to(dev()), label.to(dev()), lr)
    valid = tgt >= 0
    h = syn0[in_idx]
    w = syn1[tgt.clamp_min(0)]
    g = (label - torch.sigmoid((h * w).sum(1))) * lr * valid.float()
    assert torch.allclose(dh.cpu(), g.unsqueeze(1) * w, atol=1e-5)
    ref1 = syn1.clone().index_add_(0, tgt.clamp_min(0), g.unsqueeze(1) * h)
    assert torch.allclose(s1.cpu(), ref1, atol=1e-5)


@pytest.mark.parametrize("algo,hs,neg", [("SkipGram", False, 5), ("CBOW", True, 0), ("SkipGram", True, 3)])
def test_word2vec_trains_on_gpu_with_fused_kernel(algo, hs, neg):
    """End-to-end: Word2Vec on the GPU goes through sg_pairs_kernel (hogwild atomics) and still separates the two topics."""
    from tests.test_nlp_cpu import _corpus, _topic_margin
    from deeplearning4j_amd.nlp import Word2Vec
    sents, a, b = _corpus()
    m = (Word2Vec.Builder().minWordFrequency(1).layerSize(24).windowSize(4).epochs(25).learningRate(0.05).seed(1).iterate(sents)
         .useHierarchicSoftmax(hs).negativeSample(neg).elementsLearningAlgorithm(algo).batchSize(512).device("cuda").build())
    m.fit()
    assert m.lookupTable.syn0.is_cuda and torch.isfinite(m.lookupTable.syn0).all()
    # hogwild atomics: the update order (and so the exact margin) differs from run to run; 0.15 is far above an untrained model's ~0
    assert _topic_margin(m, a, b) > 0.15


def test_index_map_family():
    torch.manual_seed(8)
    x = torch.randn(3, 5, 7)
    p = torch.tensor([[1, 2], [0, 3], [2, 0]])
    _platform_vs_generic("pad", [x, p], [0], [1.5])
    _platform_vs_generic("pad", [x, torch.tensor([[1, 2], [2, 3], [4, 0]])], [1])
    _platform_vs_generic("pad", [x, torch.tensor([[3, 2], [5, 1], [0, 7]])], [2])
    _platform_vs_generic("mirror_pad", [x, torch.tensor([[1, 1], [2, 2], [3, 3]])], [0])
    _platform_vs_generic("mirror_pad", [x, torch.tensor([[1, 1], [2, 2], [3, 3]])], [1])
    _platform_vs_generic("pad", [x.bfloat16(), p], [0], [-2.0])
    _platform_vs_generic("pad", [torch.arange(24).reshape(2, 3, 4), torch.tensor([[1, 0], [0, 1], [2, 2]])], [0], [7.0])
    _platform_vs_generic("reverse", [x], [0, 2])
    _platform_vs_generic("reverse", [x], [])
    _platform_vs_generic("tile", [x], [2, 1, 3])
    _platform_vs_generic("tile", [x], [2, 2, 1, 3])
    _platform_vs_generic("roll", [x], [3, 2])
    _platform_vs_generic("roll", [x], [-2, 1])
    _platform_vs_generic("concat", [x, torch.randn(3, 2, 7), torch.randn(3, 1, 7)], [1])
    _platform_vs_generic("concat", [x, x * 2], [-1])
    from deeplearning4j_amd.ops import helpers as H
    xc = x.to(dev())
    assert torch.equal(H.permute_copy(xc, (2, 0, 1)).cpu(), x.permute(2, 0, 1).contiguous())
    assert torch.equal(H.strided_slice_nd(xc, [1, 0, 6], [1, 2, -2], [2, 3, 3]).cpu(), x[1:3, 0:5:2][:, :, [6, 4, 2]])
    xt = xc.transpose(0, 2)                     # non-contiguous input view
    assert torch.equal(H.reverse_dims(xt, [1]).cpu(), torch.flip(x.transpose(0, 2), [1]))
    # indices that leave the input extent are never dereferenced: the kernel writes the pad value instead
    oob = H.strided_slice_nd(xc, [0, 0, 0], [1, 1, 3], [3, 5, 7]).cpu()
    assert torch.equal(oob[:, :, :3], x[:, :, 0:7:3]) and float(oob[:, :, 3:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.uint8, torch.bfloat16, torch.float32, torch.int64])
def test_index_map_vector_paths(dtype):
    """The 16-bytes-per-thread form of index_map.hip: aligned runs (vector load + store), runs at odd element offsets (dword-aligned or
    not), rows shorter than / not a multiple of a vector, reversed and wrapped innermost dimensions, strided inputs; bit-exact vs torch."""
    from deeplearning4j_amd.ops import helpers as H
    g = torch.Generator().manual_seed(11)

    def mk(*shape):
        if dtype.is_floating_point:
            return torch.randn(*shape, generator=g).to(dtype)
        return torch.randint(0, 120, shape, generator=g).to(dtype)
    for shape in ((5, 64), (3, 7, 33), (4, 3, 130), (2, 3, 5, 48), (9,), (6, 1), (2, 5000), (70001,)):
        x = mk(*shape)
        xc = x.to(dev())
        last = len(shape) - 1
        assert torch.equal(H.reverse_dims(xc, [last]).cpu(), torch.flip(x, [last]))
        assert torch.equal(H.reverse_dims(xc, [0]).cpu(), torch.flip(x, [0]))
        for sh in (1, 3, 8, 17):
            assert torch.equal(H.roll_nd(xc, [sh], [last]).cpu(), torch.roll(x, sh, last)), (shape, sh)
        assert torch.equal(H.roll_nd(xc, [2], [0]).cpu(), torch.roll(x, 2, 0))
        reps = [2] * len(shape)
        assert torch.equal(H.tile_nd(xc, reps).cpu(), x.repeat(*reps))
        pads = [(1, 2)] * len(shape)
        ref = torch.nn.functional.pad(x.float() if dtype == torch.bfloat16 else x, [v for p in reversed(pads) for v in p], value=3).to(dtype)
        assert torch.equal(H.pad_nd(xc, pads, "constant", 3).cpu(), ref), shape
        assert torch.equal(H.pad_nd(xc, [(0, 0)] * last + [(-1, 0)], "constant", 0).cpu(), x[..., 1:]) if shape[last] > 1 else True
        y = mk(*shape)
        for ax in range(len(shape)):
            assert torch.equal(H.concat_nd([xc, y.to(dev()), xc], ax).cpu(), torch.cat([x, y, x], ax)), (shape, ax)
        if len(shape) >= 2:
            perm = list(range(len(shape)))[::-1]
            assert torch.equal(H.permute_copy(xc, perm).cpu(), x.permute(*perm).contiguous())
            xt = xc.transpose(0, last)
            assert torch.equal(H.tile_nd(xt, [1] * len(shape)).cpu(), x.transpose(0, last).contiguous())
        n = shape[last]
        if n >= 9:
            cnt = (n - 1) // 2
            assert torch.equal(H.strided_slice_nd(xc, [0] * last + [1], [1] * last + [2], list(shape[:last]) + [cnt]).cpu(), x[..., 1:1 + 2 * cnt:2])
            assert torch.equal(H.strided_slice_nd(xc, [0] * last + [3], [1] * len(shape), list(shape[:last]) + [n - 3]).cpu(), x[..., 3:])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_biasadd_vector_paths(dtype):
    """helpers_nn.hip bias_add_vec_kernel: channels-first pieces inside one channel, channels-last pieces over 8 / 4 channels, and the
    element-wise fallback for shapes that are neither."""
    tol = dict(atol=3e-2, rtol=2e-2) if dtype == torch.bfloat16 else {}
    g = torch.Generator().manual_seed(3)
    for shape, nchw in (((2, 6, 4, 8), True), ((3, 5, 8, 16), True), ((4, 7, 3, 5), True), ((5, 16), False), ((9, 64), False), ((2, 3, 4, 24), False),
                        ((7, 12), False)):
        C = shape[1] if nchw else shape[-1]
        x = torch.randn(*shape, generator=g).to(dtype)
        _platform_vs_generic("biasadd", [x, torch.randn(C, generator=g).to(dtype)], [], [], [True] if nchw else [], **tol)


def test_gather_rows_vector_path():
    """helpers_data.hip gather_rows_vec_kernel (rows that are whole 16-byte pieces) and the element-wise fallback; negative indices wrap."""
    g = torch.Generator().manual_seed(5)
    for shape, dt in (((50, 64), torch.float32), ((50, 24), torch.bfloat16), ((33, 7), torch.float32), ((40, 16), torch.int64), ((9, 48), torch.uint8)):
        table = (torch.randn(*shape, generator=g) * 40).to(dt)
        for idt in (torch.int64, torch.int32):
            idx = torch.randint(-shape[0], shape[0], (3, 29), generator=g).to(idt)
            ref = table[idx.long() % shape[0]]
            if dt.is_floating_point:
                _platform_vs_generic("embedding_lookup", [table, idx.clamp(min=0)])
            from deeplearning4j_amd.ops import helpers as H
            got = H.gather_rows(table.to(dev()), idx.to(dev()).reshape(-1)) if hasattr(H, "gather_rows") else None
            if got is not None:
                assert torch.equal(got.cpu().reshape(ref.shape), ref), (shape, dt, idt)


def test_onehot_vector_path():
    """helpers_data.hip one_hot_vec_kernel (depth a multiple of the 16-byte vector) next to the element-wise form; out-of-range and negative
    indices give an all-off row."""
    g = torch.Generator().manual_seed(2)
    for depth in (16, 1024, 10):
        for idt in (torch.int64, torch.int32):
            idx = torch.randint(-2, depth + 2, (7, 33), generator=g).to(idt)
            got = _platform_vs_generic("onehot", [idx], [-1, depth], [2.0, -1.0])[0]
            ref = torch.where(torch.arange(depth).view(1, 1, -1) == idx.long().unsqueeze(-1), 2.0, -1.0)
            assert torch.equal(got.float().cpu(), ref)
