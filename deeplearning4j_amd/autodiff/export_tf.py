"""TensorFlow GraphDef writer (frozen-graph subset) on the in-tree protobuf wire encoder, and a BERT exporter.

The reference's headline NLP path is *imported* BERT: a frozen TF graph (google-research layout: decomposed layer norm, erf GELU,
BatchMatMul attention with an additive ``(1 - mask) * -10000`` term) mapped to SameDiff, its constants turned into variables and
fine-tuned (platform-tests/.../frameworkimport/tensorflow/TestBERTGraph.java). There is no network here to fetch such a file, so
``bert_graphdef`` writes one with exactly that op vocabulary from a BertConfig + weight dict; ``TFGraphMapper.importGraph`` +
``GraphOptimizer`` then have to rediscover layer_norm / gelu / attention / fused_dense from the primitives, as with a real export.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch

from .protowire import enc_msg

_DT = {torch.float32: 1, torch.int32: 3, torch.int64: 9, torch.bool: 10}


def tf_shape(dims):
    return enc_msg(*[(2, enc_msg((1, int(d)))) for d in dims])


def tf_tensor(t: torch.Tensor):
    return enc_msg((1, _DT[t.dtype]), (2, tf_shape(t.shape)), (4, t.detach().cpu().contiguous().numpy().tobytes()))


def tf_attr(**kw):
    out = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            av = enc_msg((8, tf_tensor(v)))
        elif isinstance(v, bool):
            av = enc_msg((5, v))
        elif isinstance(v, int):
            av = enc_msg((3, v))
        elif isinstance(v, float):
            av = enc_msg((4, v))
        elif isinstance(v, str):
            av = enc_msg((2, v))
        elif isinstance(v, tuple) and v[0] == "type":
            av = enc_msg((6, v[1]))
        elif isinstance(v, tuple) and v[0] == "shape":
            av = enc_msg((7, tf_shape(v[1])))
        elif isinstance(v, list):
            av = enc_msg((1, enc_msg(*[(3, int(i)) for i in v])))
        else:
            raise TypeError(k)
        out.append((5, enc_msg((1, k), (2, av))))
    return out


def tf_node(name, op, inputs=(), **attrs):
    return enc_msg((1, name), (2, op), *[(3, i) for i in inputs], *tf_attr(**attrs))


def tf_graph(*nodes):
    return enc_msg(*[(1, n) for n in nodes])


def tf_const(name, t):
    return tf_node(name, "Const", value=t, dtype=("type", _DT[t.dtype]))


class GraphDefBuilder:
    """Append-only GraphDef builder: every method returns the node name."""

    def __init__(self):
        self.nodes: List[bytes] = []
        self._n = 0

    def _name(self, base):
        self._n += 1
        return "%s_%d" % (base, self._n)

    def node(self, op, inputs, name=None, **attrs):
        name = name or self._name(op)
        self.nodes.append(tf_node(name, op, inputs, **attrs))
        return name

    def const(self, t, name=None):
        if not isinstance(t, torch.Tensor):
            t = torch.tensor(t)
        name = name or self._name("Const")
        self.nodes.append(tf_const(name, t))
        return name

    def placeholder(self, name, dtype, shape):
        self.nodes.append(tf_node(name, "Placeholder", dtype=("type", _DT[dtype]), shape=("shape", shape)))
        return name

    def i32(self, vals):
        return self.const(torch.tensor(list(vals), dtype=torch.int32))

    def f32(self, v):
        return self.const(torch.tensor(float(v), dtype=torch.float32))

    def binary(self, op, a, b, name=None):
        return self.node(op, [a, b], name, T=("type", 1))

    def unary(self, op, a, name=None):
        return self.node(op, [a], name, T=("type", 1))

    def build(self) -> bytes:
        return tf_graph(*self.nodes)


def bert_graphdef(cfg, seqLen: int, numLabels: int, weights: Dict[str, torch.Tensor]) -> bytes:
    """Frozen BERT sequence classifier. ``weights`` uses the variable names of models/bert.py (embeddings/word, layerN/attn/q/W,
    ...); every weight becomes a Const node of that name. Placeholders: ``tokenIdx`` int32 [-1, T], ``mask`` float [-1, T]; output
    node ``logits``."""
    g = GraphDefBuilder()
    H, nh = cfg.hiddenSize, cfg.numHeads
    hd = H // nh
    T = seqLen
    W = {k: g.const(v.detach().float().cpu(), name=k) for k, v in weights.items()}
    ids = g.placeholder("tokenIdx", torch.int32, [-1, T])
    mask = g.placeholder("mask", torch.float32, [-1, T])

    def layer_norm(x, p):
        axes = g.i32([-1])
        mean = g.node("Mean", [x, axes], T=("type", 1), keep_dims=True)
        sqd = g.binary("SquaredDifference", x, mean)
        var = g.node("Mean", [sqd, axes], T=("type", 1), keep_dims=True)
        rs = g.unary("Rsqrt", g.binary("AddV2", var, g.f32(cfg.layerNormEps)))
        m = g.binary("Mul", rs, W[p + "/gain"])
        return g.binary("AddV2", g.binary("Mul", x, m), g.binary("Sub", W[p + "/bias"], g.binary("Mul", mean, m)))

    def dense(x, p):
        return g.node("BiasAdd", [g.node("MatMul", [x, W[p + "/W"]], T=("type", 1), transpose_a=False, transpose_b=False), W[p + "/b"]],
                      T=("type", 1))

    emb = g.node("GatherV2", [W["embeddings/word"], ids, g.const(torch.tensor(0, dtype=torch.int32))], Tparams=("type", 1),
                 Tindices=("type", 3))
    pos = g.node("Slice", [W["embeddings/position"], g.i32([0, 0]), g.i32([T, H])], T=("type", 1))
    typ = g.node("Slice", [W["embeddings/tokenType"], g.i32([0, 0]), g.i32([1, H])], T=("type", 1))
    x = g.binary("AddV2", emb, g.binary("AddV2", pos, typ))
    x = g.node("Reshape", [x, g.i32([-1, H])], T=("type", 1))
    x = layer_norm(x, "embeddings/ln")
    # additive attention mask, shared by all layers: (1 - mask[:, None, None, :]) * -10000
    m4 = g.node("Reshape", [mask, g.i32([-1, 1, 1, T])], T=("type", 1))
    adder = g.binary("Mul", g.binary("Sub", g.f32(1.0), m4), g.f32(-10000.0))
    for li in range(cfg.numLayers):
        p = "layer%d/" % li

        def heads(t):
            t4 = g.node("Reshape", [t, g.i32([-1, T, nh, hd])], T=("type", 1))
            return g.node("Transpose", [t4, g.i32([0, 2, 1, 3])], T=("type", 1))
        q, k, v = heads(dense(x, p + "attn/q")), heads(dense(x, p + "attn/k")), heads(dense(x, p + "attn/v"))
        s = g.node("BatchMatMulV2", [q, k], T=("type", 1), adj_x=False, adj_y=True)
        s = g.binary("AddV2", g.binary("Mul", s, g.f32(1.0 / math.sqrt(hd))), adder)
        pr = g.node("Softmax", [s], T=("type", 1))
        c = g.node("BatchMatMulV2", [pr, v], T=("type", 1), adj_x=False, adj_y=False)
        c = g.node("Reshape", [g.node("Transpose", [c, g.i32([0, 2, 1, 3])], T=("type", 1)), g.i32([-1, H])], T=("type", 1))
        x = layer_norm(g.binary("AddV2", dense(c, p + "attn/out"), x), p + "attn/ln")
        f = dense(x, p + "ffn/in")
        f = g.binary("Mul", g.binary("Mul", f, g.f32(0.5)),
                     g.binary("AddV2", g.unary("Erf", g.binary("RealDiv", f, g.f32(math.sqrt(2.0)))), g.f32(1.0)))
        x = layer_norm(g.binary("AddV2", dense(f, p + "ffn/out"), x), p + "ffn/ln")
    x3 = g.node("Reshape", [x, g.i32([-1, T, H])], T=("type", 1))
    cls = g.node("Squeeze", [g.node("Slice", [x3, g.i32([0, 0, 0]), g.i32([-1, 1, -1])], T=("type", 1))], T=("type", 1), squeeze_dims=[1])
    pooled = g.unary("Tanh", dense(cls, "pooler"))
    g.node("Identity", [dense(pooled, "classifier")], name="logits", T=("type", 1))
    return g.build()
