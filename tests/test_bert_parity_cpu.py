"""BERT against an INDEPENDENT implementation: `transformers.BertForSequenceClassification` (random init, no download) with its weights
copied into the SameDiff graph of models/bert.py. Reference recipe: platform-tests/.../frameworkimport/tensorflow/TestBERTGraph.java:74-278
compares the imported graph with the framework that produced it; the repo's earlier test compared two outputs of the same in-tree
writer. fp32: logits to 1e-4, one-step loss gradients to 1e-4 relative."""
import math

import pytest
import torch

transformers = pytest.importorskip("transformers")


def _hf(cfg, T, labels):
    from transformers import BertConfig as HFConfig, BertForSequenceClassification
    hc = HFConfig(vocab_size=cfg.vocabSize, hidden_size=cfg.hiddenSize, num_hidden_layers=cfg.numLayers, num_attention_heads=cfg.numHeads,
                  intermediate_size=cfg.intermediateSize, max_position_embeddings=cfg.maxPositions, type_vocab_size=cfg.typeVocabSize,
                  hidden_act="quick_gelu",            # x * sigmoid(1.702 x): libnd4j's GELU transform (the graph's sd.nn.gelu)
                  hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=cfg.layerNormEps, num_labels=labels,
                  attn_implementation="eager")
    torch.manual_seed(123)
    m = BertForSequenceClassification(hc)
    with torch.no_grad():                              # default init leaves biases / LN at 0 / 1: randomise them so they are checked too
        for n, p in m.named_parameters():
            if n.endswith("bias") or "LayerNorm" in n:
                p.add_(torch.randn_like(p) * 0.05)
    return m.eval()


def weight_map(cfg):
    """SameDiff variable name -> (HF parameter name, transpose?)."""
    mp = {"embeddings/word": ("bert.embeddings.word_embeddings.weight", False), "embeddings/position": ("bert.embeddings.position_embeddings.weight", False),
          "embeddings/tokenType": ("bert.embeddings.token_type_embeddings.weight", False),
          "embeddings/ln/gain": ("bert.embeddings.LayerNorm.weight", False), "embeddings/ln/bias": ("bert.embeddings.LayerNorm.bias", False),
          "pooler/W": ("bert.pooler.dense.weight", True), "pooler/b": ("bert.pooler.dense.bias", False),
          "classifier/W": ("classifier.weight", True), "classifier/b": ("classifier.bias", False)}
    for li in range(cfg.numLayers):
        p, h = "layer%d/" % li, "bert.encoder.layer.%d." % li
        for a, b in (("attn/q", "attention.self.query"), ("attn/k", "attention.self.key"), ("attn/v", "attention.self.value"),
                     ("attn/out", "attention.output.dense"), ("ffn/in", "intermediate.dense"), ("ffn/out", "output.dense")):
            mp[p + a + "/W"] = (h + b + ".weight", True)
            mp[p + a + "/b"] = (h + b + ".bias", False)
        for a, b in (("attn/ln", "attention.output.LayerNorm"), ("ffn/ln", "output.LayerNorm")):
            mp[p + a + "/gain"] = (h + b + ".weight", False)
            mp[p + a + "/bias"] = (h + b + ".bias", False)
    return mp


def load_hf_weights(sd, cfg, hf):
    params = dict(hf.named_parameters())
    mp = weight_map(cfg)
    assert sorted(mp) == sorted(sd.trainableVariables()), "every graph variable must be covered by the HF model"
    for name, (hn, tr) in mp.items():
        w = params[hn].detach()
        tgt = sd.getArrForVarName(name).tensor()
        tgt.copy_((w.t() if tr else w).to(tgt.dtype).reshape(tgt.shape))
    return mp


def _batch(cfg, B, T, labels, seed=5):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, cfg.vocabSize, (B, T), generator=g)
    mask = torch.ones(B, T)
    mask[1, T - 5:] = 0
    mask[2, 3:] = 0
    y = torch.randint(0, labels, (B,), generator=g)
    return ids, mask, y


def test_bert_forward_and_gradients_match_transformers_fp32():
    from deeplearning4j_amd.autodiff.samediff import SameDiff
    from deeplearning4j_amd.models.bert import BertConfig, build_bert_classifier
    cfg = BertConfig(vocabSize=120, hiddenSize=64, numLayers=3, numHeads=4, intermediateSize=160, maxPositions=40)
    B, T, L = 4, 16, 3
    hf = _hf(cfg, T, L)
    sd = SameDiff.create("cpu")
    build_bert_classifier(sd, cfg, T, L)
    mp = load_hf_weights(sd, cfg, hf)
    ids, mask, y = _batch(cfg, B, T, L)
    ref = hf(input_ids=ids, attention_mask=mask, labels=y)
    onehot = torch.eye(L)[y]
    out = sd.output({"tokenIdx": ids, "mask": mask, "label": onehot}, ["logits", "loss"])
    logits = out["logits"].tensor() if hasattr(out["logits"], "tensor") else out["logits"]
    loss = out["loss"].tensor() if hasattr(out["loss"], "tensor") else out["loss"]
    assert (logits - ref.logits.detach()).abs().max().item() < 1e-4
    assert abs(float(loss) - float(ref.loss.detach())) < 1e-5
    # one-step gradients of the (mean) cross entropy with respect to every parameter
    ref.loss.backward()
    hp = dict(hf.named_parameters())
    names = sorted(mp)
    grads = sd.calculateGradients({"tokenIdx": ids, "mask": mask, "label": onehot}, names)
    worst = 0.0
    for n in names:
        hn, tr = mp[n]
        g_ref = hp[hn].grad
        g_ref = g_ref.t() if tr else g_ref
        g = grads[n].tensor() if hasattr(grads[n], "tensor") else grads[n]
        denom = g_ref.norm().item()
        if n == "embeddings/position":                 # rows >= T receive no gradient on either side
            assert float(g[T:].abs().max()) == 0.0
        if n == "embeddings/tokenType":
            assert float(g[1:].abs().max()) == 0.0     # only segment 0 is used
        if n.endswith("attn/k/b"):
            # softmax is invariant to a per-query constant: the key bias has an analytically ZERO gradient; both sides hold rounding noise
            assert g.abs().max().item() < 1e-6 and g_ref.abs().max().item() < 1e-6
            continue
        rel = (g.reshape(g_ref.shape) - g_ref).norm().item() / max(denom, 1e-12)
        worst = max(worst, rel)
        assert rel < 1e-4, (n, rel)
    assert worst > 0.0


def test_bert_key_mask_is_what_changes_padded_batches():
    """Sanity of the oracle itself: without the mask the two implementations still agree, and the masked / unmasked logits differ (the
    comparison above is not vacuous with respect to the padding mask)."""
    from deeplearning4j_amd.autodiff.samediff import SameDiff
    from deeplearning4j_amd.models.bert import BertConfig, build_bert_classifier
    cfg = BertConfig(vocabSize=60, hiddenSize=32, numLayers=1, numHeads=2, intermediateSize=64, maxPositions=16)
    B, T, L = 3, 12, 2
    hf = _hf(cfg, T, L)
    sd = SameDiff.create("cpu")
    build_bert_classifier(sd, cfg, T, L)
    load_hf_weights(sd, cfg, hf)
    ids, mask, y = _batch(cfg, B, T, L)
    full = torch.ones(B, T)
    a = sd.output({"tokenIdx": ids, "mask": full}, ["logits"])["logits"]
    b = sd.output({"tokenIdx": ids, "mask": mask}, ["logits"])["logits"]
    a, b = (t.tensor() if hasattr(t, "tensor") else t for t in (a, b))
    assert (a - hf(input_ids=ids, attention_mask=full).logits.detach()).abs().max() < 1e-4
    assert (a[2] - b[2]).abs().max() > 1e-5 and torch.allclose(a[0], b[0], atol=1e-6)       # row 2 masks 9 of 12 keys, row 0 none
    assert (b - hf(input_ids=ids, attention_mask=mask).logits.detach()).abs().max() < 1e-5
