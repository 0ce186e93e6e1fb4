"""BERT encoder for sequence classification as a SameDiff graph (SURVEY config 4: the reference runs BERT through
SameDiff — TF import + fine-tuning, samediff-import tests `TestBERTGraph`, nlp BertIterator feeding `[idx, mask]`).

The graph is built from the same ops the imported TF graph is rewritten to by autodiff/optimize.py: embedding_lookup,
layer_norm, xw_plus_b, gelu, dot_product_attention_v2 (single-output mode -> fused flash attention kernels on the GPU),
softmax cross entropy. ``computeDataType="BFLOAT16"`` keeps fp32 master variables and casts weights and activations to
bf16 at their use (MFMA GEMMs, bf16 layer norm / attention kernels); the loss is computed in fp32.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


@dataclass
class BertConfig:
    vocabSize: int = 30522
    hiddenSize: int = 768
    numLayers: int = 12
    numHeads: int = 12
    intermediateSize: int = 3072
    maxPositions: int = 512
    typeVocabSize: int = 2
    layerNormEps: float = 1e-12
    initStd: float = 0.02

    @staticmethod
    def base():
        return BertConfig()

    @staticmethod
    def tiny(vocab=100, hidden=64, layers=2, heads=1, intermediate=128, maxPositions=32):
        return BertConfig(vocab, hidden, layers, heads, intermediate, maxPositions)


def build_bert_classifier(sd, cfg: BertConfig, seqLen: int, numLabels: int, computeDataType="FLOAT", seed=12345):
    """Adds the graph to ``sd``. Placeholders: ``tokenIdx`` [B, T] int, ``mask`` [B, T] float (1 = token), ``label`` [B, numLabels].
    Outputs: ``logits``, ``probs``; loss variable ``loss``. Returns the logits SDVariable."""
    g = torch.Generator().manual_seed(seed)
    H, nh = cfg.hiddenSize, cfg.numHeads
    hd = H // nh
    assert hd * nh == H and seqLen <= cfg.maxPositions
    bf16 = str(computeDataType).upper() in ("BFLOAT16", "BF16")

    def param(name, *shape, std=cfg.initStd, fill=None):
        t = torch.full(shape, float(fill)) if fill is not None else torch.randn(*shape, generator=g) * std
        return sd.var(name, t)

    def c(v):                      # weight / activation at compute precision
        return v.castTo("BFLOAT16") if bf16 else v

    idx = sd.placeHolder("tokenIdx", "INT64", -1, seqLen)
    mask = sd.placeHolder("mask", "FLOAT", -1, seqLen)
    label = sd.placeHolder("label", "FLOAT", -1, numLabels)

    wemb = param("embeddings/word", cfg.vocabSize, H)
    pemb = param("embeddings/position", cfg.maxPositions, H)
    temb = param("embeddings/tokenType", cfg.typeVocabSize, H)
    x = sd.nn.embeddingLookup(wemb, idx)                                                     # [B, T, H]
    pos = pemb.get(slice(0, seqLen), slice(None)) + temb.get(slice(0, 1), slice(None))        # [T, H] (+ segment 0)
    x = x + pos
    x = c(x).reshape(-1, H)                                                                   # [B*T, H]
    x = sd.nn.layerNorm(x, c(param("embeddings/ln/gain", H, fill=1.0)), c(param("embeddings/ln/bias", H, fill=0.0)), False, 1, eps=cfg.layerNormEps)

    for li in range(cfg.numLayers):
        p = "layer%d/" % li
        wq, bq = param(p + "attn/q/W", H, H), param(p + "attn/q/b", H, fill=0.0)
        wk, bk = param(p + "attn/k/W", H, H), param(p + "attn/k/b", H, fill=0.0)
        wv, bv = param(p + "attn/v/W", H, H), param(p + "attn/v/b", H, fill=0.0)
        wo, bo = param(p + "attn/out/W", H, H), param(p + "attn/out/b", H, fill=0.0)

        def heads(t):              # [B*T, H] -> [B, nh, T, hd]
            return t.reshape(-1, seqLen, nh, hd).permute(0, 2, 1, 3)
        q = heads(sd.nn.linear(x, c(wq), c(bq)))
        k = heads(sd.nn.linear(x, c(wk), c(bk)))
        v = heads(sd.nn.linear(x, c(wv), c(bv)))
        a = sd.nn.dotProductAttentionV2(q, v, k, None, mask, scale=1.0 / math.sqrt(hd), withWeights=False)
        a = a.permute(0, 2, 1, 3).reshape(-1, H)
        a = sd.nn.linear(a, c(wo), c(bo))
        x = sd.nn.layerNorm(x + a, c(param(p + "attn/ln/gain", H, fill=1.0)), c(param(p + "attn/ln/bias", H, fill=0.0)), False, 1, eps=cfg.layerNormEps)
        w1, b1 = param(p + "ffn/in/W", H, cfg.intermediateSize), param(p + "ffn/in/b", cfg.intermediateSize, fill=0.0)
        w2, b2 = param(p + "ffn/out/W", cfg.intermediateSize, H), param(p + "ffn/out/b", H, fill=0.0)
        f = sd.nn.linear(sd.nn.gelu(sd.nn.linear(x, c(w1), c(b1))), c(w2), c(b2))
        x = sd.nn.layerNorm(x + f, c(param(p + "ffn/ln/gain", H, fill=1.0)), c(param(p + "ffn/ln/bias", H, fill=0.0)), False, 1, eps=cfg.layerNormEps)

    cls = x.reshape(-1, seqLen, H).get(slice(None), 0, slice(None))                           # [B, H]: the [CLS] position
    pooled = sd.nn.tanh(sd.nn.linear(cls, c(param("pooler/W", H, H)), c(param("pooler/b", H, fill=0.0))))
    logits = sd.nn.linear(pooled, c(param("classifier/W", H, numLabels)), c(param("classifier/b", numLabels, fill=0.0)))
    logits = (logits.castTo("FLOAT") if bf16 else logits).rename("logits")
    sd.nn.softmax(logits, name="probs")
    sd.loss.softmaxCrossEntropy(label, logits, None, 3, name="loss").markAsLoss()
    return logits
