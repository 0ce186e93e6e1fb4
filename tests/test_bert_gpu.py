"""BERT SameDiff graph on the GPU in bf16: attention goes through the fused flash kernels (forward + backward), layer norm through
the HIP helper; the tiny model learns."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_tiny_bert_bf16_trains_through_flash_attention():
    from deeplearning4j_amd.autodiff.samediff import SameDiff, TrainingConfig
    from deeplearning4j_amd.models.bert import BertConfig, build_bert_classifier
    from deeplearning4j_amd.nd4j.learning import Adam
    from deeplearning4j_amd.ops import attention as A
    from test_bert_cpu import _data
    calls = {"fwd": 0}
    orig = A._FlashSDPA.apply

    def spy(*a):
        calls["fwd"] += 1
        return orig(*a)
    A._FlashSDPA.apply = spy
    try:
        torch.manual_seed(0)
        T = 32
        sd = SameDiff.create("cuda")
        build_bert_classifier(sd, BertConfig.tiny(vocab=50, hidden=128, layers=2, heads=2, intermediate=256, maxPositions=64), T, 2,
                              computeDataType="BFLOAT16")
        sd.setTrainingConfig(TrainingConfig.builder().updater(Adam(2e-3)).build())
        idx, mask, lab = _data(64, T, 50)
        ph = {"tokenIdx": idx.cuda(), "mask": mask.cuda(), "label": lab.cuda()}
        l0 = float(sd.output(ph, ["loss"])["loss"])
        for _ in range(60):
            sd.fit(ph)
        l1 = float(sd.output(ph, ["loss"])["loss"])
        assert calls["fwd"] >= 2 * 60, "flash SDPA kernel was not used by dot_product_attention_v2"
        assert l1 < 0.5 * l0, (l0, l1)
    finally:
        A._FlashSDPA.apply = orig


@pytest.mark.parametrize("upd", ["Adam", "Nesterovs", "RmsProp"])
def test_samediff_fused_updater_matches_per_variable_path(upd):
    """SameDiff.fit on the GPU: one segmented-updater launch over the flat variable vector == the per-variable updaters."""
    from deeplearning4j_amd.autodiff.samediff import SameDiff, TrainingConfig
    from deeplearning4j_amd.nd4j import learning as LR

    def run(fused):
        torch.manual_seed(3)
        sd = SameDiff.create("cuda")
        x = sd.placeHolder("in", "FLOAT", -1, 6)
        lab = sd.placeHolder("label", "FLOAT", -1, 3)
        w0, b0 = sd.var("w0", torch.randn(6, 16) * 0.3), sd.var("b0", torch.zeros(16))
        w1, b1 = sd.var("w1", torch.randn(16, 3) * 0.3), sd.var("b1", torch.zeros(3))
        logits = sd.nn.linear(sd.nn.tanh(sd.nn.linear(x, w0, b0)), w1, b1)
        sd.loss.softmaxCrossEntropy(lab, logits, None, 3, name="loss").markAsLoss()
        u = {"Adam": LR.Adam(1e-2), "Nesterovs": LR.Nesterovs(0.05, 0.9), "RmsProp": LR.RmsProp(1e-2)}[upd]
        sd.setTrainingConfig(TrainingConfig.builder().updater(u).l2(1e-3).weightDecay(1e-3).build())
        if not fused:
            sd._fused_update = lambda grads: False
        g = torch.Generator().manual_seed(4)
        ph = {"in": torch.randn(32, 6, generator=g).cuda(), "label": torch.eye(3)[torch.randint(0, 3, (32,), generator=g)].cuda()}
        for _ in range(12):
            sd.fit(ph)
        return {n: sd.getArrForVarName(n).tensor().detach().cpu().clone() for n in ("w0", "b0", "w1", "b1")}, sd

    a, sda = run(True)
    b, _ = run(False)
    assert getattr(sda, "_fused", None) is not None, "fused path was not taken"
    for n in a:
        assert torch.allclose(a[n], b[n], atol=2e-6, rtol=1e-5), (n, float((a[n] - b[n]).abs().max()))
    # the variables are views of one flat vector now, and still individually addressable / serialisable
    assert sda.getArrForVarName("w1").tensor().data_ptr() > sda.getArrForVarName("w0").tensor().data_ptr()
    buf = sda.asFlatBuffers(True) if hasattr(sda, "asFlatBuffers") else None
    if buf is not None:
        sd2 = SameDiff.fromFlatBuffers(bytes(buf), True, "cuda")
        assert torch.allclose(sd2.getArrForVarName("w0").tensor().cpu(), a["w0"])


def test_samediff_training_step_hip_graph_capture_matches_eager():
    """enableGraphCapture: after two eager warm-up iterations the whole step (interpreter forward, autograd backward, fused update)
    is one HIP-graph replay; parameters follow the eager trajectory, an Adam schedule still takes effect, and new data is used."""
    from deeplearning4j_amd.autodiff.samediff import SameDiff, TrainingConfig
    from deeplearning4j_amd.models.bert import BertConfig, build_bert_classifier
    from deeplearning4j_amd.nd4j.learning import Adam
    from test_bert_cpu import _data

    def run(capture):
        torch.manual_seed(0)
        T = 32
        sd = SameDiff.create("cuda")
        build_bert_classifier(sd, BertConfig.tiny(vocab=50, hidden=128, layers=2, heads=2, intermediate=256, maxPositions=64), T, 2,
                              computeDataType="BFLOAT16")
        sd.setTrainingConfig(TrainingConfig.builder().updater(Adam(1e-3)).l2(1e-4).build())
        if capture:
            sd.enableGraphCapture(True)
        losses = []
        for k in range(12):
            idx, mask, lab = _data(32, T, 50, seed=k % 3)          # the batch changes between iterations
            h = sd.fit({"tokenIdx": idx.cuda(), "mask": mask.cuda(), "label": lab.cuda()})
            losses.append(h.lossCurve.lastMeanLoss()[0])
        return losses, sd

    le, _ = run(False)
    lc, sdc = run(True)
    assert sdc._cap is not None and sdc._cap["graph"] is not None, "the step was not captured"
    assert sdc.trainingConfig.iterationCount == 12
    for a, b in zip(le, lc):
        assert abs(a - b) < 2e-2 * max(1.0, abs(a)), (le, lc)
    assert lc[-1] < lc[0]


@pytest.mark.parametrize("cdt,tol,cos_min", [("FLOAT", 2e-4, 0.9999), ("BFLOAT16", 2e-2, 0.99)])
def test_bert_gpu_matches_transformers(cdt, tol, cos_min):
    """The independent oracle of tests/test_bert_parity_cpu.py on the device paths: fp32 (library / own GEMMs, HIP layer norm) and the
    bf16 flash-attention path against `transformers` in fp32 on the host."""
    pytest.importorskip("transformers")
    from deeplearning4j_amd.autodiff.samediff import SameDiff
    from deeplearning4j_amd.models.bert import BertConfig, build_bert_classifier
    from test_bert_parity_cpu import _batch, _hf, load_hf_weights
    cfg = BertConfig(vocabSize=200, hiddenSize=128, numLayers=2, numHeads=2, intermediateSize=256, maxPositions=64)
    B, T, L = 8, 32, 3
    hf = _hf(cfg, T, L)
    sd = SameDiff.create("cuda")
    build_bert_classifier(sd, cfg, T, L, computeDataType=cdt)
    mp = load_hf_weights(sd, cfg, hf)
    ids, mask, y = _batch(cfg, B, T, L)
    ref = hf(input_ids=ids, attention_mask=mask, labels=y)
    ph = {"tokenIdx": ids.cuda(), "mask": mask.cuda(), "label": torch.eye(L)[y].cuda()}
    out = sd.output(ph, ["logits", "loss"])
    lg = out["logits"].tensor() if hasattr(out["logits"], "tensor") else out["logits"]
    assert (lg.float().cpu() - ref.logits.detach()).abs().max().item() < tol
    ref.loss.backward()
    hp = dict(hf.named_parameters())
    names = [n for n in sorted(mp) if not n.endswith("attn/k/b")]
    grads = sd.calculateGradients(ph, names)
    for n in names:
        hn, tr = mp[n]
        g_ref = (hp[hn].grad.t() if tr else hp[hn].grad).reshape(-1)
        g = grads[n].tensor() if hasattr(grads[n], "tensor") else grads[n]
        g = g.float().cpu().reshape(-1)
        if n in ("embeddings/position", "embeddings/tokenType"):
            g, g_ref = g[:T * cfg.hiddenSize], g_ref[:T * cfg.hiddenSize]
        cos = float(torch.dot(g, g_ref) / (g.norm() * g_ref.norm() + 1e-30))
        assert cos > cos_min, (n, cos)
