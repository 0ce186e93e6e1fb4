"""MultiLayerNetwork (reference: deeplearning4j-nn/.../nn/multilayer/MultiLayerNetwork.java, 4131 LoC).

fit / output / feedForward / computeGradientAndScore / rnnTimeStep / truncated BPTT / evaluate /
save / load over the shared flat-vector engine in network_base.py.
Call-stack parity with SURVEY §3.2: fit -> (AsyncDataSetIterator) -> per minibatch:
forward (all layers) -> output-layer score+delta (fused softmax/MCXENT) -> backward (reverse) ->
single fused updater launch -> listeners. No host synchronisation inside the iteration.
"""
from __future__ import annotations

import copy
import time
from typing import List, Optional

import torch

from ..nd4j.activations import ActivationReLU, ActivationIdentity
from .conf.base import BackpropType, DataType
from .conf.configuration import MultiLayerConfiguration
from .conf import layers as C
from .layers import registry
from .layers.base import LayerImpl
from .layers.core import BatchNormalizationImpl, ConvolutionLayerImpl, ActivationLayerImpl
from .layers.recurrent import LSTMImpl, SimpleRnnImpl
from .network_base import BaseNetwork


class MultiLayerNetwork(BaseNetwork):
    def __init__(self, conf: MultiLayerConfiguration, params=None, device=None):
        super().__init__()
        if isinstance(conf, str):
            conf = MultiLayerConfiguration.fromJson(conf)
        self.conf = conf
        self._seed = conf.confs[0].seed if conf.confs else 12345
        self._minibatch = conf.confs[0].miniBatch if conf.confs else True
        self._device = device
        self._init_params = params
        self.layers: List[LayerImpl] = []
        self.iterationCount, self.epochCount = conf.iterationCount, conf.epochCount
        self._input = self._labels = None
        self._fmask = self._lmask = None

    # ------------------------------------------------------------------ init
    def init(self, params=None, cloneParametersArray=True):
        if self._initialized:
            return self
        if params is None:
            params = self._init_params
        self.ctx = self._make_ctx(self.conf, self._device)
        self.ctx.iteration, self.ctx.epoch = self.iterationCount, self.epochCount
        self.layers = [registry.create(c.layer, self.ctx) for c in self.conf.confs]
        for i, l in enumerate(self.layers):
            l.index = i
            if l.name is None:
                l.name = "layer%d" % i
        seen_params = False
        for l in self.layers:
            l.needs_input_grad = seen_params
            seen_params = seen_params or len(l.param_specs()) > 0
        self._alloc_params(params, cloneParametersArray)
        self._plan = self._build_plan()
        return self

    def _ordered_layers(self):
        return self.layers

    def _build_plan(self):
        """Fusion planner: conv(bias) -> BN folds the bias; BN -> ActivationLayer(ReLU) fuses the ReLU."""
        plan = []
        i, n = 0, len(self.layers)
        pre = self.conf.inputPreProcessors
        while i < n:
            l = self.layers[i]
            if (isinstance(l, BatchNormalizationImpl) and i + 1 < n and isinstance(self.layers[i + 1], ActivationLayerImpl)
                    and isinstance(self.layers[i + 1].conf.activationFn, ActivationReLU) and (i + 1) not in pre
                    and self.layers[i + 1].conf.idropout is None):
                plan.append(("bn_relu", i))
                i += 2
                continue
            plan.append(("layer", i))
            i += 1
        for i in range(n - 1):
            a, b = self.layers[i], self.layers[i + 1]
            if (isinstance(a, ConvolutionLayerImpl) and isinstance(b, BatchNormalizationImpl) and "b" in a.params
                    and isinstance(a.conf.activationFn, ActivationIdentity) and (i + 1) not in pre
                    and b.conf.idropout is None):
                a.fold_bias = True
                b.conv_bias = (lambda a=a: a.params["b"].reshape(-1))
            if (isinstance(a, ConvolutionLayerImpl) and isinstance(b, BatchNormalizationImpl)
                    and isinstance(a.conf.activationFn, ActivationIdentity) and (i + 1) not in pre
                    and b.conf.idropout is None and (a.fold_bias or "b" not in a.params)):
                a.want_stats = True
                b.producer = a
                a.stats_consumer = b
        return plan

    # ------------------------------------------------------------------ accessors
    def getLayer(self, i):
        if isinstance(i, str):
            for l in self.layers:
                if l.name == i:
                    return l
            raise KeyError(i)
        return self.layers[i]

    def getLayers(self):
        return self.layers

    def getnLayers(self):
        return len(self.layers)

    def getOutputLayer(self):
        return self.layers[-1]

    def getLayerWiseConfigurations(self):
        return self.conf

    def getConfiguration(self):
        return self.conf

    def setInput(self, x):
        self._input = x

    def setLabels(self, y):
        self._labels = y

    def getInput(self):
        return self._input

    def getLabels(self):
        return self._labels

    def setLayerMaskArrays(self, fmask, lmask):
        self._fmask, self._lmask = fmask, lmask

    def clearLayerMaskArrays(self):
        self._fmask = self._lmask = None
        for l in self.layers:
            l.maskArray = None

    def clear(self):
        self._input = self._labels = None
        for l in self.layers:
            l.clear()

    # ------------------------------------------------------------------ forward
    def _layer_forward(self, l, x, train, mode=None):
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

    def _forward(self, x, train, upto=None, mode=None, collect=None, fmask=None):
        """Run the (fused) plan. Returns the activation of the last executed layer."""
        n = len(self.layers) if upto is None else upto + 1
        pre = self.conf.inputPreProcessors
        mb = x.shape[0]
        mask = fmask
        plan = self._plan if collect is None else [("layer", i) for i in range(len(self.layers))]
        runs = self._lstm_runs() if (collect is None and mode != "timestep" and x.is_cuda) else {}
        skip = -1
        for kind, i in plan:
            if i >= n:
                break
            if i <= skip:
                continue
            if i in pre:
                x = pre[i].preProcess(x, mb, train)
                mask = pre[i].feedForwardMaskArray(mask, mb) if mask is not None else None
            l = self.layers[i]
            if mask is not None:
                mask = l.feedForwardMaskArray(mask, mb)
            else:
                l.maskArray = None
            run = runs.get(i)
            if run is not None and mask is None and run[-1] < n:
                from .layers import lstm_stack
                ls = [self.layers[j] for j in run]
                if lstm_stack.ok(ls, x, train):
                    # consecutive LSTM layers: one layer x chunk wavefront over several HIP streams (nn/layers/lstm_stack.py)
                    x = lstm_stack.forward(ls, x, train, mode)
                    skip = run[-1]
                    continue
            if kind == "bn_relu" and i + 1 < n:
                x = l.forward(x, train, relu=True)
            else:
                x = self._layer_forward(l, x, train, mode)
            if collect is not None:
                collect.append(x)
        return x

    def _lstm_runs(self):
        if getattr(self, "_lstm_runs_cache", None) is None:
            from .layers import lstm_stack
            self._lstm_runs_cache = lstm_stack.find_runs(self.layers, self._plan, self.conf.inputPreProcessors)
        return self._lstm_runs_cache

    def output(self, x, train=False, featuresMask=None, labelsMask=None):
        if hasattr(x, "getFeatures") and not isinstance(x, torch.Tensor):
            x = x.getFeatures()
        if hasattr(x, "hasNext"):
            outs = []
            x.reset()
            while x.hasNext():
                outs.append(self.output(x.next().getFeatures(), train))
            return torch.cat(outs, 0)
        self._check_init()
        with torch.no_grad():
            x = self._prep_features(x)
            cap = getattr(self, "_infer_cap", None)
            if cap is not None and not train and featuresMask is None and labelsMask is None:
                outs = cap.run(lambda z: [self._wrap_out(self._forward(z[0], False, fmask=None))], [x])
                if outs is not None:
                    return outs[0]
            with self._ws_scope(False):
                y = self._wrap_out(self._forward(x, train, fmask=self._to_dev(featuresMask)))
        self.clearLayerMaskArrays()
        return y

    def _wrap_out(self, y):
        if y.dtype != self.ctx.dtype and y.is_floating_point():
            y = y.to(self.ctx.dtype)
        if y.dim() == 4 and not y.is_contiguous():
            y = y.contiguous()
        return y

    def feedForward(self, x=None, train=False):
        """All activations: [input, act(layer0), ..., act(layerN-1)] (unfused, API parity)."""
        self._check_init()
        if x is None:
            x = self._input
        with torch.no_grad():
            x = self._prep_features(x)
            acts = [x]
            self._forward(x, train, collect=acts)
        return acts

    def feedForwardToLayer(self, layerNum, x, train=False):
        return self.feedForward(x, train)[:layerNum + 2]

    def activate(self, x, train=False):
        return self.output(x, train)

    def activateSelectedLayers(self, frm, to, x):
        with torch.no_grad():
            x = self._prep_features(x)
            pre = self.conf.inputPreProcessors
            for i in range(frm, to + 1):
                if i in pre:
                    x = pre[i].preProcess(x, x.shape[0], False)
                x = self.layers[i].forward(x, False)
        return x

    def predict(self, x):
        if hasattr(x, "getFeatures"):
            x = x.getFeatures()
        return self.output(x).argmax(dim=1).tolist()

    def _check_init(self):
        if not self._initialized:
            self.init()

    # ------------------------------------------------------------------ backward
    def _backward(self, mb):
        from ..ops import streams
        with streams.backward_scope():          # conv wgrads run on a side stream, joined when the scope closes
            return self._backward_impl(mb)

    def _backward_impl(self, mb):
        pre = self.conf.inputPreProcessors
        eps = None
        steps = list(self._plan)
        prog = self._on_backward_progress
        if prog is not None:
            if self._on_backward_begin is not None:
                self._on_backward_begin()
            offs = self._layer_min_offsets()
        first = True
        ends = {r[-1]: r for r in self._lstm_runs().values()}
        skip_lo = len(self.layers)
        for kind, i in reversed(steps):
            if i >= skip_lo:
                continue                # part of an LSTM stack that has been back-propagated as a whole
            l = self.layers[i]
            if prog is not None and not first:
                prog(offs[i + 1])       # layers i+1.. have finished writing their gradients
            first = False
            run = ends.get(i)
            if run is not None and eps is not None and eps.is_cuda:
                from .layers import lstm_stack
                ls = [self.layers[j] for j in run]
                if lstm_stack.can_backward(ls):
                    eps = lstm_stack.backward(ls, eps)
                    skip_lo = i = run[0]
                    if eps is not None and i in pre:
                        eps = pre[i].backprop(eps, mb)
                    if eps is None:
                        break
                    continue
            if getattr(l, "is_output", False) and i == len(self.layers) - 1:
                eps = l.backward(None)
            else:
                if eps is None:
                    break
                if not l.needs_input_grad and not l.param_specs():
                    break   # nothing trainable at or below this layer
                eps = l.backward(eps)
            if eps is not None and i in pre:
                eps = pre[i].backprop(eps, mb)
            if eps is None:
                break
        return eps

    def _layer_min_offsets(self):
        """offs[i] = lowest flat-gradient offset owned by layers i..end."""
        if getattr(self, "_layer_offs", None) is None:
            by_impl = {}
            for e in self._entries:
                by_impl.setdefault(id(e.layer), []).append(e.off)
            offs = [self._num_params] * (len(self.layers) + 1)
            for i in range(len(self.layers) - 1, -1, -1):
                offs[i] = min([offs[i + 1]] + by_impl.get(id(self.layers[i]), []))
            self._layer_offs = offs
        return self._layer_offs

    def computeGradientAndScore(self, x=None, y=None, fmask=None, lmask=None, mode=None):
        """Forward + backward; fills the flat gradient vector and the device-resident score."""
        self._check_init()
        x = self._prep_features(self._input if x is None else x)
        y = self._prep_labels(self._labels if y is None else y)
        fmask = self._to_dev(self._fmask if fmask is None else fmask)
        lmask = self._to_dev(self._lmask if lmask is None else lmask)
        mb = x.shape[0]
        self._last_batch = mb
        out_layer = self.layers[-1]
        if not getattr(out_layer, "is_output", False):
            raise RuntimeError("Cannot compute gradient and score: final layer is not an output layer (%s)" %
                               type(out_layer.conf).__name__)
        for lst in self.listeners:
            if hasattr(lst, "onForwardPass"):
                pass
        if self._grads.is_cuda:
            # one clear of the flat gradient vector replaces a memset in front of every split-K wgrad (53 on ResNet-50). It has to
            # come BEFORE the forward pass: BatchNormalization writes its running-statistics deltas (the "gradients" of mean /
            # var, BatchNormalization.java:330-360) into this vector while it normalises.
            self._grads.zero_()
            self.ctx.grads_zeroed = True
        self._forward(x, True, mode=mode, fmask=fmask)
        out_layer.setLabels(y, lmask)
        ssum = out_layer.compute_score_sum()
        self._score = (ssum / mb) if self._minibatch else ssum
        self._backward(mb)
        for lst in self.listeners:
            if hasattr(lst, "onGradientCalculation"):
                lst.onGradientCalculation(self)
        return self._grads, self._score

    # ------------------------------------------------------------------ fit
    def fit(self, data, labels=None, numEpochs=1, featuresMask=None, labelsMask=None):
        self._check_init()
        if labels is not None or isinstance(data, torch.Tensor):
            self._fit_batch(data, labels, featuresMask, labelsMask)
            return self
        if hasattr(data, "getFeatures") and not hasattr(data, "hasNext"):
            self._fit_batch(data.getFeatures(), data.getLabels(), data.getFeaturesMaskArray(), data.getLabelsMaskArray())
            return self
        from ..datasets.iterators import AsyncDataSetIterator
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
                self._fit_batch(ds.getFeatures(), ds.getLabels(), ds.getFeaturesMaskArray(), ds.getLabelsMaskArray())
                t0 = time.time()
            if src is not it and hasattr(src, "shutdown"):
                src.shutdown()
            for l in self.listeners:
                if hasattr(l, "onEpochEnd"):
                    l.onEpochEnd(self)
            self.incrementEpochCount()
        return self

    def _fit_batch(self, x, y, fmask=None, lmask=None):
        if (self.conf.backpropType == BackpropType.TruncatedBPTT and torch.as_tensor(x).dim() == 3
                and torch.as_tensor(x).shape[2] > self.conf.tbpttFwdLength):
            return self._fit_tbptt(x, y, fmask, lmask)
        if self._graph_step is not None and fmask is None and lmask is None:
            self._graph_step.run(x, y)
        else:       # masked / variable-length batches are not part of the captured signature: run them eagerly
            with self._ws_scope(True):
                self.computeGradientAndScore(x, y, fmask, lmask)
                self._apply_update(self._last_batch)
        self.clearLayerMaskArrays()
        self._after_iteration()

    def _fit_tbptt(self, x, y, fmask, lmask):
        """Truncated BPTT (MultiLayerNetwork.doTruncatedBPTT, :2083)."""
        x, y = self._to_dev(x), self._to_dev(y)
        T, L = x.shape[2], self.conf.tbpttFwdLength
        self.rnnClearPreviousState()
        for s in range(0, T, L):
            e = min(s + L, T)
            xs = x[:, :, s:e]
            ys = y[:, :, s:e] if y.dim() == 3 else y
            fm = fmask[:, s:e] if fmask is not None else None
            lm = lmask[:, s:e] if lmask is not None and lmask.dim() == 2 and lmask.shape[1] == T else lmask
            with self._ws_scope(True):
                self.computeGradientAndScore(xs, ys, fm, lm, mode="tbptt")
                self._apply_update(self._last_batch)
            self._after_iteration()
        self.rnnClearPreviousState()
        self.clearLayerMaskArrays()

    # ------------------------------------------------------------------ rnn
    def rnnTimeStep(self, x):
        self._check_init()
        x = self._prep_features(x)
        squeeze = x.dim() == 2
        if squeeze:
            x = x.unsqueeze(2)
        with torch.no_grad():
            y = self._forward(x, False, mode="timestep")
        if squeeze and y.dim() == 3:
            y = y[:, :, 0]
        return self._wrap_out(y)

    def rnnClearPreviousState(self):
        for l in self.layers:
            if hasattr(l, "rnnClearPreviousState"):
                l.rnnClearPreviousState()

    def rnnGetPreviousState(self, layer):
        return self.layers[layer].rnnGetPreviousState()

    def rnnSetPreviousState(self, layer, state):
        self.layers[layer].rnnSetPreviousState(state)

    # ------------------------------------------------------------------ scoring / evaluation
    def _score_dataset(self, ds, training=False):
        with torch.no_grad():
            x = self._prep_features(ds.getFeatures())
            y = self._prep_labels(ds.getLabels())
            self._forward(x, training, fmask=self._to_dev(ds.getFeaturesMaskArray()))
            out = self.layers[-1]
            out.setLabels(y, self._to_dev(ds.getLabelsMaskArray()))
            s = out.compute_score_sum() / x.shape[0]
            r = self.calcRegularizationScore()
        self.clearLayerMaskArrays()
        return float(s + r) if r is not None else float(s)

    def scoreExamples(self, ds, addRegularizationTerms=True):
        with torch.no_grad():
            x = self._prep_features(ds.getFeatures())
            self._forward(x, False)
            out = self.layers[-1]
            out.setLabels(self._prep_labels(ds.getLabels()), None)
            s = out.score_array().float()
            if addRegularizationTerms:
                r = self.calcRegularizationScore()
                if r is not None:
                    s = s + r.float()
        return s

    def evaluate(self, iterator, labelsList=None, topN=1):
        from ..eval.evaluation import Evaluation
        return self.doEvaluation(iterator, Evaluation(labels=labelsList, topN=topN))[0]

    def evaluateRegression(self, iterator):
        from ..eval.evaluation import RegressionEvaluation
        return self.doEvaluation(iterator, RegressionEvaluation())[0]

    def evaluateROC(self, iterator, thresholdSteps=0):
        from ..eval.evaluation import ROC
        return self.doEvaluation(iterator, ROC(thresholdSteps))[0]

    def evaluateROCMultiClass(self, iterator, thresholdSteps=0):
        from ..eval.evaluation import ROCMultiClass
        return self.doEvaluation(iterator, ROCMultiClass(thresholdSteps))[0]

    def doEvaluation(self, iterator, *evals):
        if iterator.resetSupported():
            iterator.reset()
        while iterator.hasNext():
            ds = iterator.next()
            out = self.output(ds.getFeatures(), False, ds.getFeaturesMaskArray())
            lab = self._to_dev(ds.getLabels())
            for e in evals:
                e.eval(lab, out, self._to_dev(ds.getLabelsMaskArray()))
        return list(evals)

    # ------------------------------------------------------------------ pretrain (AutoEncoder / VAE layers)
    def pretrain(self, iterator, numEpochs=1):
        for i, l in enumerate(self.layers):
            if l.isPretrainLayer():
                self.pretrainLayer(i, iterator, numEpochs)

    def pretrainLayer(self, idx, iterator, numEpochs=1):
        l = self.layers[idx]
        if not l.isPretrainLayer():
            return
        for _ in range(numEpochs):
            iterator.reset()
            while iterator.hasNext():
                ds = iterator.next()
                x = self._prep_features(ds.getFeatures())
                with torch.no_grad():
                    if idx > 0:
                        x = self._forward(x, False, upto=idx - 1)
                    if idx in self.conf.inputPreProcessors:
                        x = self.conf.inputPreProcessors[idx].preProcess(x, x.shape[0], False)
                self._grads.zero_()
                leaves = {k: l.w(k).detach().requires_grad_(True) for k in l.params}
                with torch.enable_grad():
                    loss = l.pretrain_loss(x, leaves)
                gs = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
                for (k, _), g in zip(leaves.items(), gs):
                    if g is not None:
                        l.grads[k].copy_(g.reshape(l.grads[k].shape))
                self._score = loss.detach() / x.shape[0]
                self._last_batch = x.shape[0]
                self._apply_update(x.shape[0])
                self._after_iteration()

    # ------------------------------------------------------------------ misc API
    def summary(self):
        lines = ["=" * 100, "%-32s %-28s %-14s %-12s" % ("LayerName (LayerType)", "nIn,nOut", "TotalParams", "ParamsShape"),
                 "=" * 100]
        tot = 0
        for i, l in enumerate(self.layers):
            c = l.conf
            n = l.numParams()
            tot += n
            shapes = ", ".join("%s:%s" % (s.key, list(s.shape)) for s in l.param_specs())
            lines.append("%-32s %-28s %-14d %s" % ("%s (%s)" % (l.name, type(c).__name__),
                                                   "%s,%s" % (getattr(c, "nIn", "-"), getattr(c, "nOut", "-")), n, shapes))
        lines += ["-" * 100, "%30s %d" % ("Total Parameters:", tot), "=" * 100]
        return "\n".join(lines)

    def clone(self):
        net = MultiLayerNetwork(self.conf.clone(), device=self.ctx.device if self._initialized else self._device)
        if self._initialized:
            net.init(self._params, True)
            net.setUpdaterState(self.getUpdaterState())
        return net

    def save(self, path, saveUpdater=True):
        from ..utils.model_serializer import ModelSerializer
        ModelSerializer.writeModel(self, path, saveUpdater)

    @staticmethod
    def load(path, loadUpdater=True, device=None):
        from ..utils.model_serializer import ModelSerializer
        return ModelSerializer.restoreMultiLayerNetwork(path, loadUpdater, device=device)

    def toComputationGraph(self):
        from .graph.computation_graph import ComputationGraph
        from .conf.configuration import NeuralNetConfiguration
        gb = NeuralNetConfiguration.Builder().seed(self._seed).graphBuilder().addInputs("in")
        prev = "in"
        for i, c in enumerate(self.conf.confs):
            name = str(i)
            pre = self.conf.inputPreProcessors.get(i)
            if pre is not None:
                gb.addLayer(name, copy.deepcopy(c.layer), pre, prev)
            else:
                gb.addLayer(name, copy.deepcopy(c.layer), prev)
            prev = name
        gb.setOutputs(prev)
        cgc = gb.build()
        cgc.dataType, cgc.computeDataType = self.conf.dataType, self.conf.computeDataType
        cg = ComputationGraph(cgc, device=self.ctx.device if self._initialized else None)
        cg.init(self._params if self._initialized else None)
        return cg

    def enableGraphCapture(self, enabled=True):
        """Capture the training iteration into a HIP graph (static shapes; GPU only)."""
        from .graph_capture import CapturedTrainStep
        self._graph_step = CapturedTrainStep(self) if enabled and self.ctx.device.type == "cuda" else None
        return self

    def enableInferenceCapture(self, enabled=True):
        """Replay output() as a HIP graph per input signature (GPU only; calls with masks or train=True stay eager)."""
        from .graph_capture import CapturedInference
        self._infer_cap = CapturedInference() if enabled and self.ctx.device.type == "cuda" else None
        return self


from ..utils.crash_reporting import install as _install_crash_reporting  # noqa: E402

_install_crash_reporting(MultiLayerNetwork)
