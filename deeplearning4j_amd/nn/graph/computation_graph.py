"""ComputationGraph (reference: deeplearning4j-nn/.../nn/graph/ComputationGraph.java, 4917 LoC).

DAG of layers and vertices executed in topological order (ComputationGraph.java:1187) with
epsilons summed at fan-out on the way back (:2795-2803). On top of the reference semantics the
init-time *fusion planner* rewrites the execution plan (never the configuration / parameter
layout) the way a ResNet wants it on MI355X:

* Conv(+bias) -> BatchNorm           : bias folded out of the conv (it only shifts BN's mean);
* BatchNorm -> ReLU                  : one fused normalise+ReLU pass, ReLU mask reused in backward;
* BatchNorm -> Add(shortcut) -> ReLU : one fused normalise+add+ReLU pass; the shortcut may itself be
                                       a BatchNorm (conv blocks) and is then normalised on the fly.

This removes ~6 elementwise launches and ~4 full-tensor copies per bottleneck tail that the
reference issues.
"""
from __future__ import annotations

import copy
import os
import time
from collections import OrderedDict
from typing import Dict, List, Optional

import torch

from ...nd4j.activations import ActivationReLU, ActivationIdentity
from ..conf import graph as G
from ..conf import layers as C
from ..conf.base import BackpropType, ConvolutionMode, PoolingType
from ..conf.configuration import ComputationGraphConfiguration
from ..layers import registry
from ..layers.core import BatchNormalizationImpl, ConvolutionLayerImpl, ActivationLayerImpl, SubsamplingLayerImpl, ZeroPaddingLayerImpl
from ..layers.recurrent import LSTMImpl, SimpleRnnImpl
from ..network_base import BaseNetwork
from ...ops.conv_igemm import MaskedGrad
from ...ops.bn_pool import PoolBnGrad
from .vertices import create_vertex, VertexImpl, ElementWiseVertexImpl, LastTimeStepVertexImpl, \
    DuplicateToTimeSeriesVertexImpl, ReverseTimeSeriesVertexImpl


class _Node:
    __slots__ = ("out", "inputs", "impl", "pre", "is_layer", "is_output", "names")

    def __init__(self, out, inputs, impl, pre=None, is_layer=True, is_output=False, names=None):
        self.out, self.inputs, self.impl, self.pre = out, list(inputs), impl, pre
        self.is_layer, self.is_output, self.names = is_layer, is_output, names or [out]

    def forward(self, xs, train, mode=None):
        if self.is_layer:
            x = xs[0]
            if self.pre is not None:
                x = self.pre.preProcess(x, x.shape[0], train)
            l = self.impl
            if mode is not None and isinstance(l, (LSTMImpl, SimpleRnnImpl)):
                if mode == "timestep":
                    st = l.stateMap
                    init = None
                    if st:
                        init = (st["prevActivation"], st["prevMemCell"]) if "prevMemCell" in st else (st["prevActivation"],)
                    return l.forward(x, False, initial_state=init, store_state=True)
                if mode == "tbptt" and isinstance(l, LSTMImpl):
                    y = l.forward(x, train, initial_state=l.tbpttState.get("s"))
                    l.tbpttState["s"] = tuple(t.detach() for t in l._last_state)
                    return y
            return l.forward(x, train)
        return self.impl.forward(xs, train)

    def backward(self, eps, mb):
        if self.is_layer:
            dx = self.impl.backward(eps)
            if dx is not None and self.pre is not None:
                dx = self.pre.backprop(dx, mb)
            return [dx]
        return self.impl.backward(eps)


class _FusedBNNode(_Node):
    """BN [-> Add(shortcut [BN'])] -> ReLU executed as one fused node."""
    __slots__ = ("bn", "bn2", "has_res")

    def __init__(self, out, inputs, bn, bn2, has_res, names):
        super().__init__(out, inputs, bn, None, False, False, names)
        self.bn, self.bn2, self.has_res = bn, bn2, has_res

    def forward(self, xs, train, mode=None):
        bn = self.bn
        if not self.has_res:
            return bn.forward(xs[0], train, relu=True)
        if self.bn2 is None:
            return bn.forward(xs[0], train, relu=True, residual=xs[1])
        rs, rt = self.bn2.stats_only(xs[1], train)
        return bn.forward(xs[0], train, relu=True, residual=xs[1], res_coeffs=(rs, rt))

    def backward(self, eps, mb):
        if not self.has_res:
            return [self.bn.backward(eps)]
        # dz = eps * [out > 0] is not written when the mask exists as bits: the identity shortcut receives a MaskedGrad
        # (consumed by the epilogue of the data-gradient launch that accumulates into it), a shortcut BatchNorm reads eps
        # through the bits
        dxa, dz = self.bn.backward(eps, return_dz=True, lazy_dz=True)
        if self.bn2 is None:
            return [dxa, dz]
        from ...ops.conv_igemm import MaskedGrad
        if isinstance(dz, MaskedGrad):
            return [dxa, self.bn2.backward(dz.src, ext_mask_bits=dz.bits)]
        return [dxa, self.bn2.backward(dz)]


class _FusedBNPoolNode(_Node):
    """BN -> ReLU -> MaxPool (the stem of the ImageNet networks) as one forward pass and a backward that never writes the full-resolution
    activation or its gradient (ops/bn_pool.py); layer by layer whenever the fused kernels do not take the call."""
    __slots__ = ("bn", "pool", "_st")

    def __init__(self, out, inputs, bn, pool, names):
        super().__init__(out, inputs, bn, None, False, False, names)
        self.bn, self.pool, self._st = bn, pool, None

    def forward(self, xs, train, mode=None):
        from ...ops import bn_pool as BP
        bn, pc, x = self.bn, self.pool.conf, xs[0]
        self._st = None
        if x.dim() == 4 and x.is_cuda and x.dtype == torch.bfloat16:
            x2, shp = bn._to2d(x)
            geom = BP.geometry(shp, pc.kernelSize, pc.stride, pc.padding)
            if x2.is_contiguous() and BP.usable(x2, geom):
                scale, shift = bn.coefficients(x2, train)
                y, idx = BP.forward(x2, scale, shift, geom, train)
                if train:
                    bn._x2, bn._shp = x2, shp
                    bn._out2 = bn._z2 = bn._mask_coeffs = bn._mask_bits = None
                    bn._want_dz = False
                    self._st = (y, idx, geom)
                return y.permute(0, 3, 1, 2)
        return self.pool.forward(bn.forward(x, train, relu=True), train)

    def backward(self, eps, mb):
        if self._st is None:
            return [self.bn.backward(self.pool.backward(eps))]
        from ...ops import bn_pool as BP
        bn = self.bn
        y, idx, geom = self._st
        dy = eps.permute(0, 2, 3, 1)
        if not dy.is_contiguous():
            dy = dy.contiguous()
        if dy.dtype != y.dtype:
            dy = dy.to(y.dtype)
        gamma = bn.params["gamma"].reshape(-1) if "gamma" in bn.params else None
        beta = bn.params["beta"].reshape(-1) if "beta" in bn.params else None
        dg = bn.grads["gamma"].reshape(-1) if "gamma" in bn.grads else None
        db = bn.grads["beta"].reshape(-1) if "beta" in bn.grads else None
        if BP.LAZY_DX:
            lazy = BP.backward_lazy(bn._x2, dy, y, idx, gamma, beta, bn._mean, bn._invstd, dg, db, geom, bn._shp, bn._from2d)
            self._st = None
            return [lazy]
        dx2 = BP.backward(bn._x2, dy, y, idx, gamma, beta, bn._mean, bn._invstd, dg, db, geom)
        self._st = None
        return [bn._from2d(dx2, bn._shp)]


class _LstmStackNode(_Node):
    """A chain of LSTM / GravesLSTM layer vertices, each feeding only the next one: run as one layer x chunk wavefront over several HIP
    streams (nn/layers/lstm_stack.py) when the shapes allow it, vertex by vertex otherwise."""
    __slots__ = ("chain", "impls")

    def __init__(self, chain):
        last = chain[-1]
        super().__init__(last.out, chain[0].inputs, last.impl, chain[0].pre, True, False, [c.out for c in chain])
        self.chain, self.impls = chain, [c.impl for c in chain]

    def forward(self, xs, train, mode=None):
        from ..layers import lstm_stack
        x = xs[0]
        if mode != "timestep" and x.is_cuda and all(l.maskArray is None for l in self.impls):
            xp = x if self.pre is None else self.pre.preProcess(x, x.shape[0], train)
            if lstm_stack.ok(self.impls, xp, train):
                return lstm_stack.forward(self.impls, xp, train, mode)
        for c in self.chain:
            x = c.forward([x], train, mode)
        return x

    def backward(self, eps, mb):
        from ..layers import lstm_stack
        if eps.is_cuda and lstm_stack.can_backward(self.impls):
            dx = lstm_stack.backward(self.impls, eps)
        else:
            dx = eps
            for c in reversed(self.chain[1:]):
                dx = c.impl.backward(dx)
            dx = self.chain[0].impl.backward(dx)
        if dx is not None and self.pre is not None:
            dx = self.pre.backprop(dx, mb)
        return [dx]


def _stats_only(self, x, train):
    """Batch statistics / coefficients for a BN whose normalisation is applied inside another fused node."""
    x2, shp = self._to2d(x)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    scale, shift = self.coefficients(x2, train)
    if train:
        self._x2, self._shp, self._out2, self._z2, self._want_dz = x2, shp, None, None, False
        self._mask_coeffs = None
    return scale, shift


BatchNormalizationImpl.stats_only = _stats_only


class ComputationGraph(BaseNetwork):
    def __init__(self, conf: ComputationGraphConfiguration, device=None):
        super().__init__()
        if isinstance(conf, str):
            conf = ComputationGraphConfiguration.fromJson(conf)
        self.conf = conf
        self._seed = conf.defaultConfiguration.seed
        self._minibatch = conf.defaultConfiguration.miniBatch
        self._device = device
        self.iterationCount, self.epochCount = conf.iterationCount, conf.epochCount
        self.vertices: "OrderedDict[str, object]" = OrderedDict()
        self._inputs = self._labels = None
        self.fusion_enabled = True
        self._bn_sinks_enabled = os.environ.get("DL4J_AMD_BN_DGRAD_SINK", "1") == "1"

    # ------------------------------------------------------------------ init
    def init(self, params=None, cloneParametersArray=True):
        if self._initialized:
            return self
        conf = self.conf
        self.ctx = self._make_ctx(conf, self._device)
        self.ctx.iteration, self.ctx.epoch = self.iterationCount, self.epochCount
        self.topologicalOrder = [n for n in conf.topological_order() if n not in conf.networkInputs]
        self._impls: "OrderedDict[str, object]" = OrderedDict()
        for n in self.topologicalOrder:
            v = conf.vertices[n]
            if isinstance(v, G.LayerVertex):
                impl = registry.create(v.layer, self.ctx)
                impl.name = n
            else:
                impl = create_vertex(v, self.ctx)
                impl.name = n
            self._impls[n] = impl
        self.vertices = self._impls
        # an activation "requires grad" only if a trainable parameter sits upstream of it; layers whose
        # input does not require grad skip their backward-data pass (e.g. the stem conv behind ZeroPadding)
        req = {i: False for i in conf.networkInputs}
        for n, impl in self._impls.items():
            in_req = any(req[i] for i in conf.vertexInputs[n])
            if isinstance(conf.vertices[n], G.LayerVertex):
                impl.needs_input_grad = in_req
            req[n] = in_req or len(impl.param_specs()) > 0
        self._requires_grad = req
        self._alloc_params(params, cloneParametersArray)
        self._plan = self._build_plan()
        return self

    def _ordered_layers(self):
        return list(self._impls.values())

    def _layer_key(self, idx):
        return list(self._impls.keys())[idx]

    def _consumers(self):
        cons: Dict[str, List[str]] = {}
        for n, ins in self.conf.vertexInputs.items():
            for i in ins:
                cons.setdefault(i, []).append(n)
        return cons

    def _build_plan(self):
        conf = self.conf
        cons = self._consumers()
        outs = set(conf.networkOutputs)
        impls = self._impls
        fused_away = {}
        nodes: "OrderedDict[str, _Node]" = OrderedDict()

        def single(n):
            return len(cons.get(n, [])) == 1 and n not in outs

        def is_relu_layer(n):
            l = impls.get(n)
            return (isinstance(l, ActivationLayerImpl) and isinstance(l.conf.activationFn, ActivationReLU)
                    and l.conf.idropout is None and conf.vertices[n].preProcessor is None)

        def is_plain_max_pool(n):
            l = impls.get(n)
            if not isinstance(l, SubsamplingLayerImpl) or conf.vertices[n].preProcessor is not None or l.conf.idropout is not None:
                return False
            c = l.conf
            return (c.poolingType == PoolingType.MAX and tuple(c.dilation) == (1, 1)
                    and (c.convolutionMode or ConvolutionMode.Truncate) != ConvolutionMode.Same)

        def is_plain_bn(n):
            l = impls.get(n)
            return isinstance(l, BatchNormalizationImpl) and l.conf.idropout is None and conf.vertices[n].preProcessor is None

        for impl in impls.values():
            if isinstance(impl, ConvolutionLayerImpl):
                impl.fold_bias = False
                impl.want_stats = False
                impl.stats_consumer = None
            elif isinstance(impl, BatchNormalizationImpl):
                impl.conv_bias = None
                impl.producer = None
            elif isinstance(impl, ZeroPaddingLayerImpl):
                impl.passthrough = False
            if isinstance(impl, ConvolutionLayerImpl):
                impl.extra_pad = (0, 0)
        if self.fusion_enabled:
            # ZeroPadding (symmetric) -> Convolution (truncate mode): the convolution pads inside its own kernels, the padded copy
            # of the input (the 230x230 image batch of the ImageNet stems) is never made
            for n in self.topologicalOrder:
                z = impls[n]
                if not (type(z) is ZeroPaddingLayerImpl and single(n) and conf.vertices[n].preProcessor is None and z.conf.idropout is None):
                    continue
                t_, b_, l_, r_ = z.conf.padding
                cn = cons[n][0]
                a = impls.get(cn)
                if (t_ == b_ and l_ == r_ and type(a) is ConvolutionLayerImpl and conf.vertices[cn].preProcessor is None
                        and a.conf.idropout is None and (a.conf.convolutionMode or ConvolutionMode.Truncate) != ConvolutionMode.Same):
                    z.passthrough = True
                    a.extra_pad = (int(t_), int(l_))
            for n in self.topologicalOrder:
                if n in fused_away or not is_plain_bn(n) or not single(n):
                    continue
                c = cons[n][0]
                if is_relu_layer(c):
                    fused_away[n] = c
                    fused_away[c] = c
                    pl = cons[c][0] if single(c) else None
                    if pl is not None and is_plain_max_pool(pl):
                        # BN -> ReLU -> MaxPool (the ImageNet stem): one forward pass, backward without the full-resolution gradient
                        fused_away[c] = pl
                        fused_away[n] = pl
                        fused_away[pl] = pl
                        nodes[pl] = _FusedBNPoolNode(pl, conf.vertexInputs[n], impls[n], impls[pl], [n, c, pl])
                        continue
                    nodes[c] = _FusedBNNode(c, conf.vertexInputs[n], impls[n], None, False, [n, c])
                    continue
                ci = impls.get(c)
                if (isinstance(ci, ElementWiseVertexImpl) and ci.conf.op == "Add" and len(conf.vertexInputs[c]) == 2
                        and single(c) and is_relu_layer(cons[c][0])):
                    r = cons[c][0]
                    other = [i for i in conf.vertexInputs[c] if i != n]
                    if len(other) != 1:
                        continue
                    o = other[0]
                    if o in fused_away and fused_away[o] != o:
                        continue            # o is interior to another fused node (not materialised); a fused node's *output* is fine
                    bn2 = None
                    in2 = o
                    if o in impls and is_plain_bn(o) and single(o):
                        bn2 = impls[o]
                        in2 = conf.vertexInputs[o][0]
                        fused_away[o] = r
                    fused_away[n] = r
                    fused_away[c] = r
                    fused_away[r] = r
                    nodes[r] = _FusedBNNode(r, [conf.vertexInputs[n][0], in2], impls[n], bn2, True, [n, c, r] + ([o] if bn2 else []))
            # conv bias folding
            for n in self.topologicalOrder:
                a = impls[n]
                if (isinstance(a, ConvolutionLayerImpl) and "b" in a.params and isinstance(a.conf.activationFn, ActivationIdentity)
                        and single(n)):
                    b = impls.get(cons[n][0])
                    if isinstance(b, BatchNormalizationImpl) and conf.vertices[cons[n][0]].preProcessor is None and b.conf.idropout is None:
                        a.fold_bias = True
                        b.conv_bias = (lambda a=a: a.params["b"].reshape(-1))
            # conv -> BN (any bias handling): fuse the BN statistics into the conv epilogue
            for n in self.topologicalOrder:
                a = impls[n]
                if isinstance(a, ConvolutionLayerImpl) and isinstance(a.conf.activationFn, ActivationIdentity) and single(n):
                    b = impls.get(cons[n][0])
                    if (isinstance(b, BatchNormalizationImpl) and conf.vertices[cons[n][0]].preProcessor is None
                            and b.conf.idropout is None and (a.fold_bias or "b" not in a.params)):
                        a.want_stats = True
                        b.producer = a
                        a.stats_consumer = b
        plan = []
        for n in self.topologicalOrder:
            if n in fused_away:
                if fused_away[n] == n or (n in nodes):
                    pass
                if n in nodes:
                    plan.append(nodes[n])
                continue
            v = conf.vertices[n]
            impl = impls[n]
            if isinstance(v, G.LayerVertex):
                plan.append(_Node(n, conf.vertexInputs[n], impl, v.preProcessor, True, getattr(impl, "is_output", False) and n in outs))
            else:
                plan.append(_Node(n, conf.vertexInputs[n], impl, None, False, False))
        if self.fusion_enabled:
            plan = self._merge_lstm_chains(plan, cons, outs, fused_away)
        self._fused_alias = fused_away
        # backward bookkeeping for the fused dgrad epilogue: for every tensor name the plan index of its FIRST consumer (the
        # last one to contribute a gradient in the reversed sweep) and, when the tensor is the output of a BatchNorm node, that
        # BatchNorm implementation
        first, bn_of = {}, {}
        for k, nd in enumerate(plan):
            for i in nd.inputs:
                first.setdefault(i, k)
            if isinstance(nd, _FusedBNNode):
                bn_of[nd.out] = nd.bn
            elif type(nd) is _Node and nd.is_layer and nd.pre is None and type(nd.impl) is BatchNormalizationImpl:
                bn_of[nd.out] = nd.impl
        for o in outs:
            first.pop(o, None)                       # a network output also receives the loss gradient
        self._plan_first_consumer, self._plan_bn_of = first, bn_of
        # last-use table for freeing activations in inference
        last = {}
        for k, nd in enumerate(plan):
            for i in nd.inputs:
                last[i] = k
        self._last_use = last
        return plan

    # ------------------------------------------------------------------ accessors
    def getLayer(self, name):
        if isinstance(name, int):
            return [l for l in self._impls.values() if not isinstance(l, VertexImpl)][name]
        return self._impls[name]

    def getLayers(self):
        return [l for l in self._impls.values() if not isinstance(l, VertexImpl)]

    def getVertex(self, name):
        return self._impls[name]

    def getVertices(self):
        return list(self._impls.values())

    def getConfiguration(self):
        return self.conf

    def getNumInputArrays(self):
        return len(self.conf.networkInputs)

    def getNumOutputArrays(self):
        return len(self.conf.networkOutputs)

    def getOutputLayer(self, i=0):
        return self._impls[self.conf.networkOutputs[i]]

    def topologicalSortOrder(self):
        return list(self.topologicalOrder)

    def setInputs(self, *xs):
        self._inputs = list(xs)

    def setInput(self, i, x):
        if self._inputs is None:
            self._inputs = [None] * len(self.conf.networkInputs)
        self._inputs[i] = x

    def setLabels(self, *ys):
        self._labels = list(ys)

    def clear(self):
        self._inputs = self._labels = None
        for l in self._impls.values():
            l.clear()

    def clearLayerMaskArrays(self):
        for l in self._impls.values():
            if hasattr(l, "maskArray"):
                l.maskArray = None
            if hasattr(l, "mask"):
                l.mask = None

    def _check_init(self):
        if not self._initialized:
            self.init()

    # ------------------------------------------------------------------ forward / backward
    def _set_masks(self, fmasks):
        if not fmasks or all(m is None for m in fmasks):
            return
        masks = {n: self._to_dev(m) for n, m in zip(self.conf.networkInputs, fmasks) if m is not None}
        # propagate time-series masks along the graph (first-input rule, ComputationGraph.setLayerMaskArrays)
        for n in self.topologicalOrder:
            ins = self.conf.vertexInputs[n]
            m = next((masks[i] for i in ins if i in masks), None)
            impl = self._impls[n]
            if m is None:
                continue
            if isinstance(impl, (LastTimeStepVertexImpl, ReverseTimeSeriesVertexImpl)):
                nm = impl.conf.maskArrayInputName
                impl.mask = masks.get(nm, m) if nm else m
                if isinstance(impl, LastTimeStepVertexImpl):
                    continue
                masks[n] = m
            elif hasattr(impl, "feedForwardMaskArray"):
                v = self.conf.vertices[n]
                mm = m
                if isinstance(v, G.LayerVertex) and v.preProcessor is not None:
                    mm = v.preProcessor.feedForwardMaskArray(m, m.shape[0])
                out = impl.feedForwardMaskArray(mm, m.shape[0])
                if out is not None:
                    masks[n] = out
            else:
                masks[n] = m

    def _forward(self, inputs, train, mode=None, keep_all=False, fmasks=None):
        acts = dict(zip(self.conf.networkInputs, inputs))
        self._set_masks(fmasks)
        for n in self.topologicalOrder:
            impl = self._impls[n]
            if isinstance(impl, DuplicateToTimeSeriesVertexImpl):
                ref = impl.conf.inputName
                impl.ref_T = acts[ref].shape[2] if ref in acts else 1
        last = self._last_use
        outs = set(self.conf.networkOutputs)
        for k, nd in enumerate(self._plan):
            xs = [acts[i] for i in nd.inputs]
            acts[nd.out] = nd.forward(xs, train, mode)
            if not train and not keep_all:
                for i in nd.inputs:
                    if last.get(i) == k and i not in outs:
                        acts.pop(i, None)
        return acts

    @staticmethod
    def _merge_lstm_chains(plan, cons, outs, fused_away):
        """Chains of plain LSTM layer vertices (each output used by the next vertex only, no preprocessor inside the chain) become one
        _LstmStackNode at the position of the chain's last vertex."""
        by_out = {nd.out: nd for nd in plan}

        def plain(nd):
            return (type(nd) is _Node and nd.is_layer and not nd.is_output and type(nd.impl) is LSTMImpl and not nd.impl.reverse
                    and len(nd.inputs) == 1)
        chains, used = [], set()
        for nd in plan:
            if nd.out in used or not plain(nd):
                continue
            prev = by_out.get(nd.inputs[0])
            if prev is not None and plain(prev) and len(cons.get(prev.out, [])) == 1 and prev.out not in outs and nd.pre is None:
                continue                      # not the head of its chain
            chain = [nd]
            while True:
                c = cons.get(chain[-1].out, [])
                if len(c) != 1 or chain[-1].out in outs:
                    break
                nx = by_out.get(c[0])
                if nx is None or not plain(nx) or nx.pre is not None:
                    break
                chain.append(nx)
            if len(chain) >= 2:
                chains.append(chain)
                used.update(c.out for c in chain)
        if not chains:
            return plan
        last_of = {ch[-1].out: ch for ch in chains}
        out = []
        for nd in plan:
            if nd.out in last_of:
                ch = last_of[nd.out]
                for c in ch:
                    fused_away[c.out] = nd.out
                out.append(_LstmStackNode(ch))
            elif nd.out not in used:
                out.append(nd)
        return out

    def _prep_inputs(self, xs):
        if isinstance(xs, torch.Tensor) or not isinstance(xs, (list, tuple)):
            xs = [xs]
        return [self._prep_features(x) for x in xs]

    def output(self, *xs, train=False, featureMasks=None):
        self._check_init()
        if len(xs) == 1 and isinstance(xs[0], (list, tuple)):
            xs = xs[0]
        if len(xs) == 1 and hasattr(xs[0], "hasNext"):
            it = xs[0]
            it.reset()
            res = []
            while it.hasNext():
                d = it.next()
                f = d.getFeatures()
                res.append(self.output(*(f if isinstance(f, (list, tuple)) else [f])))
            return [torch.cat([r[i] for r in res], 0) for i in range(len(res[0]))]
        with torch.no_grad():
            ins = self._prep_inputs(list(xs))
            cap = getattr(self, "_infer_cap", None)
            if cap is not None and not train and featureMasks is None:
                outs = cap.run(lambda z: [self._wrap(a) for a in (lambda acts: [acts[o] for o in self.conf.networkOutputs])(
                    self._forward(z, False, fmasks=None))], ins)
                if outs is not None:
                    return outs
            with self._ws_scope(False):
                acts = self._forward(ins, train, fmasks=featureMasks)
                outs = [self._wrap(acts[o]) for o in self.conf.networkOutputs]
        self.clearLayerMaskArrays()
        return outs

    def outputSingle(self, *xs, train=False):
        return self.output(*xs, train=train)[0]

    def _wrap(self, y):
        if y.is_floating_point() and y.dtype != self.ctx.dtype:
            y = y.to(self.ctx.dtype)
        if y.dim() == 4 and not y.is_contiguous():
            y = y.contiguous()
        return y

    def feedForward(self, *xs, train=False):
        """Map of ALL vertex activations (unfused execution, API parity)."""
        self._check_init()
        if len(xs) == 1 and isinstance(xs[0], (list, tuple)):
            xs = xs[0]
        old_plan, old_last = self._plan, self._last_use
        fe = self.fusion_enabled
        try:
            self.fusion_enabled = False
            self._plan = self._build_plan()
            with torch.no_grad():
                acts = self._forward(self._prep_inputs(list(xs)), train, keep_all=True)
        finally:
            self.fusion_enabled = fe
            self._plan, self._last_use = old_plan, old_last
            self._build_plan_restore()
        return acts

    def _build_plan_restore(self):
        # restore fold flags after an unfused feedForward
        self._plan = self._build_plan()

    def _backward(self, mb):
        from ...ops import streams
        with streams.backward_scope():          # conv wgrads run on a side stream, joined when the scope closes
            return self._backward_impl(mb)

    def _backward_impl(self, mb):
        eps: Dict[str, torch.Tensor] = {}
        net_in = set(self.conf.networkInputs)
        prog = self._on_backward_progress
        if prog is not None:
            if self._on_backward_begin is not None:
                self._on_backward_begin()
            offs = self._node_min_offsets()
        for k_rev, nd in enumerate(reversed(self._plan)):
            if prog is not None and k_rev > 0:
                prog(offs[len(self._plan) - k_rev])
            if nd.is_output:
                dxs = [nd.impl.backward(None)]
                if dxs[0] is not None and nd.pre is not None:
                    dxs[0] = nd.pre.backprop(dxs[0], mb)
            else:
                e = eps.pop(nd.out, None)
                if e is None:
                    continue
                if isinstance(e, MaskedGrad):
                    e = e.materialize()
                elif isinstance(e, PoolBnGrad) and not (type(nd) is _Node and nd.is_layer and nd.pre is None
                                                        and type(nd.impl) is ConvolutionLayerImpl):
                    e = e.materialize()                   # only a plain convolution layer knows how to consume it unevaluated
                # fan-out accumulation fused into the conv dgrad epilogue: the gradient already collected for this node's
                # input is updated in place instead of materialising dx and running a separate add
                acc_in = sink = sink_bn = None
                if (type(nd) is _Node and nd.is_layer and nd.pre is None and len(nd.inputs) == 1
                        and isinstance(nd.impl, ConvolutionLayerImpl) and type(nd.impl) is ConvolutionLayerImpl
                        and nd.impl.supports_accum()):
                    inp = nd.inputs[0]
                    acc_in = eps.get(inp)
                    if acc_in is not None and not isinstance(acc_in, MaskedGrad):
                        # the target must be exclusively owned: vertices such as Add hand the *same* tensor to several inputs
                        sp = acc_in.untyped_storage().data_ptr()
                        if sp == e.untyped_storage().data_ptr() or any(
                                isinstance(t, torch.Tensor) and t is not acc_in and t.untyped_storage().data_ptr() == sp
                                for t in eps.values()):
                            acc_in = None
                    # this conv is the LAST contributor to its input's gradient and that input comes out of a BatchNorm node:
                    # the dgrad epilogue then also emits that BatchNorm's backward partial sums (no column-reduction pass)
                    k_fwd = len(self._plan) - 1 - k_rev
                    if (self._bn_sinks_enabled and self._plan_first_consumer.get(inp) == k_fwd
                            and (acc_in is not None) == (eps.get(inp) is not None)):
                        sink_bn = self._plan_bn_of.get(inp)
                        if sink_bn is not None:
                            sink = sink_bn.backward_sink()
                if acc_in is not None or sink is not None:
                    dx = nd.impl.backward(e, accum=acc_in, bn_sink=sink)
                    if sink is not None and sink.partial is not None:
                        sink_bn._bwd_sink = sink
                    if dx is not None:
                        prev = eps.get(inp) if acc_in is None else None
                        eps[inp] = dx if prev is None else prev + dx
                    continue
                dxs = nd.backward(e, mb)
            for inp, dx in zip(nd.inputs, dxs):
                if dx is None or inp in net_in or not self._requires_grad.get(inp, True):
                    continue
                prev = eps.get(inp)
                eps[inp] = dx if prev is None else prev + dx
        return eps

    def _node_min_offsets(self):
        """offs[k] = smallest T such that NO unfinished plan node (0..k-1) owns a flat-gradient offset >= T, i.e. every
        gradient at offset >= offs[k] is final once plan nodes k..end have run backward. Fused nodes own parameters that
        sit earlier in the flat vector than parameters of nodes that run later in backward (a fused BN node at the trailing
        ReLU owns the shortcut BN while the shortcut conv is still pending), so the bound is the highest END offset among
        the unfinished nodes, not the lowest start offset among the finished ones."""
        if getattr(self, "_node_offs", None) is not None and len(self._node_offs) == len(self._plan) + 1:
            return self._node_offs
        by_impl = {}
        for e in self._entries:
            by_impl.setdefault(id(e.layer), []).append(e.off + e.spec.numel)
        offs = [0] * (len(self._plan) + 1)
        hi = 0                                    # highest end offset owned by plan nodes 0..k-1
        for k, nd in enumerate(self._plan):
            offs[k] = hi
            for impl in self._node_impls(nd):
                for end in by_impl.get(id(impl), []):
                    hi = max(hi, end)
        offs[len(self._plan)] = hi
        self._node_offs = offs
        return offs

    @staticmethod
    def _node_impls(nd):
        """Every layer implementation whose parameters a plan node's backward writes (fused nodes own several)."""
        out = []
        for attr in ("impl", "bn", "bn2", "conv", "act"):
            v = getattr(nd, attr, None)
            if v is not None and not any(v is o for o in out):
                out.append(v)
        for v in getattr(nd, "impls", None) or []:
            if not any(v is o for o in out):
                out.append(v)
        return out

    def computeGradientAndScore(self, inputs=None, labels=None, fmasks=None, lmasks=None, mode=None):
        self._check_init()
        inputs = self._prep_inputs(self._inputs if inputs is None else inputs)
        labels = self._labels if labels is None else labels
        if isinstance(labels, torch.Tensor) or not isinstance(labels, (list, tuple)):
            labels = [labels]
        labels = [self._prep_labels(y) for y in labels]
        mb = inputs[0].shape[0]
        self._last_batch = mb
        if self._grads.is_cuda:
            # one clear of the flat gradient vector replaces a memset in front of every split-K wgrad (53 on ResNet-50). It has to
            # come BEFORE the forward pass: BatchNormalization writes its running-statistics deltas (the "gradients" of mean /
            # var, BatchNormalization.java:330-360) into this vector while it normalises.
            self._grads.zero_()
            self.ctx.grads_zeroed = True
        self._forward(inputs, True, mode, fmasks=fmasks)
        total = None
        for i, o in enumerate(self.conf.networkOutputs):
            ol = self._impls[o]
            if not getattr(ol, "is_output", False):
                raise RuntimeError("Cannot compute score: output vertex '%s' is not an output layer" % o)
            lm = self._to_dev(lmasks[i]) if lmasks is not None and lmasks[i] is not None else None
            ol.setLabels(labels[i], lm)
            s = ol.compute_score_sum()
            total = s if total is None else total + s
        self._score = (total / mb) if self._minibatch else total
        self._backward(mb)
        for lst in self.listeners:
            if hasattr(lst, "onGradientCalculation"):
                lst.onGradientCalculation(self)
        return self._grads, self._score

    # ------------------------------------------------------------------ fit
    def fit(self, data, labels=None, numEpochs=1):
        self._check_init()
        if labels is not None or isinstance(data, torch.Tensor):
            self._fit_batch(data if isinstance(data, (list, tuple)) else [data],
                            labels if isinstance(labels, (list, tuple)) else [labels], None, None)
            return self
        if hasattr(data, "getFeatures") and not hasattr(data, "hasNext"):
            self._fit_ds(data)
            return self
        from ...datasets.iterators import AsyncDataSetIterator
        it = data
        for _ in range(numEpochs):
            if it.resetSupported():
                it.reset()
            src = it
            if it.asyncSupported() and not isinstance(it, AsyncDataSetIterator):
                src = AsyncDataSetIterator(it, 2, device=self.ctx.device)
            for l in self.listeners:
                if hasattr(l, "onEpochStart"):
                    l.onEpochStart(self)
            t0 = time.time()
            while src.hasNext():
                ds = src.next()
                self.lastEtlTime = int((time.time() - t0) * 1000)
                self._fit_ds(ds)
                t0 = time.time()
            if src is not it and hasattr(src, "shutdown"):
                src.shutdown()
            for l in self.listeners:
                if hasattr(l, "onEpochEnd"):
                    l.onEpochEnd(self)
            self.incrementEpochCount()
        return self

    def _fit_ds(self, ds):
        f, l = ds.getFeatures(), ds.getLabels()
        fm = getattr(ds, "getFeaturesMaskArrays", None)
        if isinstance(f, (list, tuple)):
            self._fit_batch(list(f), list(l), ds.getFeaturesMaskArrays(), ds.getLabelsMaskArrays())
        else:
            fmk, lmk = ds.getFeaturesMaskArray(), ds.getLabelsMaskArray()
            self._fit_batch([f], [l], [fmk] if fmk is not None else None, [lmk] if lmk is not None else None)

    def _fit_batch(self, xs, ys, fmasks, lmasks):
        x0 = torch.as_tensor(xs[0]) if not hasattr(xs[0], "shape") else xs[0]
        if (self.conf.backpropType == BackpropType.TruncatedBPTT and len(x0.shape) == 3
                and x0.shape[2] > self.conf.tbpttFwdLength):
            return self._fit_tbptt(xs, ys, fmasks, lmasks)
        if self._graph_step is not None and fmasks is None and lmasks is None and len(xs) == 1 and len(ys) == 1:
            self._graph_step.run(xs[0], ys[0])
        else:
            with self._ws_scope(True):
                self.computeGradientAndScore(xs, ys, fmasks, lmasks)
                self._apply_update(self._last_batch)
        self.clearLayerMaskArrays()
        self._after_iteration()

    def _fit_tbptt(self, xs, ys, fmasks, lmasks):
        xs = [self._to_dev(x) for x in xs]
        ys = [self._to_dev(y) for y in ys]
        T, L = xs[0].shape[2], self.conf.tbpttFwdLength
        self.rnnClearPreviousState()
        for s in range(0, T, L):
            e = min(s + L, T)
            sx = [x[:, :, s:e] if x.dim() == 3 else x for x in xs]
            sy = [y[:, :, s:e] if y.dim() == 3 else y for y in ys]
            fm = [m[:, s:e] if m is not None else None for m in fmasks] if fmasks else None
            lm = [m[:, s:e] if m is not None and m.dim() == 2 and m.shape[1] == T else m for m in lmasks] if lmasks else None
            with self._ws_scope(True):
                self.computeGradientAndScore(sx, sy, fm, lm, mode="tbptt")
                self._apply_update(self._last_batch)
            self._after_iteration()
        self.rnnClearPreviousState()
        self.clearLayerMaskArrays()

    # ------------------------------------------------------------------ rnn
    def rnnTimeStep(self, *xs):
        self._check_init()
        ins = self._prep_inputs(list(xs))
        squeeze = ins[0].dim() == 2
        if squeeze:
            ins = [x.unsqueeze(2) for x in ins]
        with torch.no_grad():
            acts = self._forward(ins, False, mode="timestep")
        outs = [self._wrap(acts[o]) for o in self.conf.networkOutputs]
        if squeeze:
            outs = [o[:, :, 0] if o.dim() == 3 else o for o in outs]
        return outs

    def rnnClearPreviousState(self):
        for l in self._impls.values():
            if hasattr(l, "rnnClearPreviousState"):
                l.rnnClearPreviousState()

    # ------------------------------------------------------------------ scoring / evaluation
    def _score_dataset(self, ds, training=False):
        f, l = ds.getFeatures(), ds.getLabels()
        f = list(f) if isinstance(f, (list, tuple)) else [f]
        l = list(l) if isinstance(l, (list, tuple)) else [l]
        with torch.no_grad():
            ins = self._prep_inputs(f)
            self._forward(ins, training)
            tot = 0.0
            for i, o in enumerate(self.conf.networkOutputs):
                ol = self._impls[o]
                ol.setLabels(self._prep_labels(l[i]), None)
                tot = tot + ol.compute_score_sum()
            s = tot / ins[0].shape[0]
            r = self.calcRegularizationScore()
        return float(s + r) if r is not None else float(s)

    def evaluate(self, iterator, labelsList=None, topN=1):
        from ...eval.evaluation import Evaluation
        return self.doEvaluation(iterator, Evaluation(labels=labelsList, topN=topN))[0]

    def evaluateRegression(self, iterator):
        from ...eval.evaluation import RegressionEvaluation
        return self.doEvaluation(iterator, RegressionEvaluation())[0]

    def evaluateROC(self, iterator, thresholdSteps=0):
        from ...eval.evaluation import ROC
        return self.doEvaluation(iterator, ROC(thresholdSteps))[0]

    def doEvaluation(self, iterator, *evals):
        if iterator.resetSupported():
            iterator.reset()
        while iterator.hasNext():
            ds = iterator.next()
            f = ds.getFeatures()
            out = self.output(*(f if isinstance(f, (list, tuple)) else [f]))[0]
            lab = ds.getLabels()
            lab = lab[0] if isinstance(lab, (list, tuple)) else lab
            lm = ds.getLabelsMaskArray() if hasattr(ds, "getLabelsMaskArray") else None
            for e in evals:
                e.eval(self._to_dev(lab), out, self._to_dev(lm))
        return list(evals)

    # ------------------------------------------------------------------ misc
    def summary(self):
        lines = ["=" * 110, "%-36s %-22s %-12s %-14s %s" % ("VertexName (VertexType)", "nIn,nOut", "TotalParams", "ParamsShape", "Inputs"),
                 "=" * 110]
        tot = 0
        for n, impl in self._impls.items():
            c = impl.conf
            np_ = sum(s.numel for s in impl.param_specs())
            tot += np_
            lines.append("%-36s %-22s %-12d %-14s %s" % ("%s (%s)" % (n, type(c).__name__), "%s,%s" % (getattr(c, "nIn", "-"), getattr(c, "nOut", "-")),
                                                         np_, ",".join(s.key for s in impl.param_specs()), self.conf.vertexInputs[n]))
        lines += ["-" * 110, "%30s %d" % ("Total Parameters:", tot), "=" * 110]
        return "\n".join(lines)

    def clone(self):
        net = ComputationGraph(self.conf.clone(), device=self.ctx.device if self._initialized else self._device)
        if self._initialized:
            net.init(self._params, True)
            net.setUpdaterState(self.getUpdaterState())
        return net

    def save(self, path, saveUpdater=True):
        from ...utils.model_serializer import ModelSerializer
        ModelSerializer.writeModel(self, path, saveUpdater)

    @staticmethod
    def load(path, loadUpdater=True, device=None):
        from ...utils.model_serializer import ModelSerializer
        return ModelSerializer.restoreComputationGraph(path, loadUpdater, device=device)

    def enableGraphCapture(self, enabled=True):
        from ..graph_capture import CapturedTrainStep
        self._graph_step = CapturedTrainStep(self) if enabled and self.ctx.device.type == "cuda" else None
        return self

    def enableInferenceCapture(self, enabled=True):
        """Replay output() as a HIP graph per input signature (GPU only; calls with masks or train=True stay eager)."""
        from ..graph_capture import CapturedInference
        self._infer_cap = CapturedInference() if enabled and self.ctx.device.type == "cuda" else None
        return self


from ...utils.crash_reporting import install as _install_crash_reporting  # noqa: E402

_install_crash_reporting(ComputationGraph)
