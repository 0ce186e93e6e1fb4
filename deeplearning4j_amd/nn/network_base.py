"""Shared machinery of MultiLayerNetwork and ComputationGraph.

* ONE flat parameter vector + ONE flat gradient vector + ONE flat updater-state vector; layers hold
  views (reference: MultiLayerNetwork.init, ComputationGraph.java:527,772 — the flat layout is the
  checkpoint ABI of ModelSerializer).
* UpdaterBlocks (contiguous params with equal updater/regularisation config — reference
  nn/updater/BaseMultiLayerUpdater.java:83-141, UpdaterUtils.java:30-56) become rows of a device
  segment table consumed by the single-launch fused updater (ops/updater.py).
* Optional bf16 compute replicas of the weights (fp32 master) refreshed by one cast-table launch.
* The whole training iteration is stream-asynchronous: the score stays on the device until someone
  asks for it; an optional HIP-graph capture replays the iteration with ~zero launch overhead.
"""
from __future__ import annotations

import math
import time
from typing import Dict, List, Optional, Tuple

import os

import torch

from ..nd4j.learning import IUpdater, NoOp, L1Regularization, L2Regularization, WeightDecay
from ..ops import updater as U
from .conf.base import DataType, GradientNormalization
from .conf.layers import ParamSpec
from .layers.base import LayerCtx, LayerImpl


def default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def _strided_view(flat, off, spec: ParamSpec):
    n = spec.numel
    seg = flat[off:off + n]
    shape = spec.shape
    if spec.order == "f" and len(shape) > 1:
        strides, s = [], 1
        for d in shape:
            strides.append(s)
            s *= d
        return seg.as_strided(shape, strides)
    return seg.view(shape)


class ParamEntry:
    __slots__ = ("layer_idx", "layer", "spec", "off", "updater", "regs", "divide")

    def __init__(self, layer_idx, layer, spec, off, updater, regs, divide):
        self.layer_idx, self.layer, self.spec, self.off = layer_idx, layer, spec, off
        self.updater, self.regs, self.divide = updater, regs, divide


class BaseNetwork:
    def __init__(self):
        self.listeners = []
        self._initialized = False
        self.iterationCount = 0
        self.epochCount = 0
        self._score = None          # device scalar (loss/mb, without regularisation) of the last iteration
        self._last_batch = 0
        self.lastEtlTime = 0
        self._graph_step = None
        self._grad_sync = None              # DP hook: called between backward and update
        self._on_backward_progress = None   # DP overlap hook: lowest finished gradient offset
        self._on_backward_begin = None
        self._dp_buckets = None             # DP: GradientBuckets for the segmented HIP-graph capture

    # ------------------------------------------------------------------ to be provided by subclasses
    def _ordered_layers(self) -> List[LayerImpl]:
        raise NotImplementedError

    # ------------------------------------------------------------------ init
    def _make_ctx(self, conf, device=None):
        dt = DataType.to_torch(conf.dataType)
        cdt = DataType.to_torch(conf.computeDataType) if getattr(conf, "computeDataType", None) else None
        dev = torch.device(device) if device is not None else default_device()
        return LayerCtx(dt, cdt, dev, getattr(self, "_seed", 12345))

    def _alloc_params(self, params=None, clone=True):
        """Allocate flat vectors, create views, initialise parameters."""
        ctx = self.ctx
        layers = self._ordered_layers()
        entries: List[ParamEntry] = []
        off = 0
        for li, l in enumerate(layers):
            for spec in l.param_specs():
                upd = l.conf.getUpdaterByParam(spec) if hasattr(l.conf, "getUpdaterByParam") else NoOp()
                regs = l.conf.getRegularizationByParam(spec) if hasattr(l.conf, "getRegularizationByParam") else []
                entries.append(ParamEntry(li, l, spec, off, upd, regs, spec.role != "stat"))
                off += spec.numel
        self._entries = entries
        self._num_params = off
        dev, dt = ctx.device, ctx.dtype
        if params is not None:
            p = params.reshape(-1).to(device=dev, dtype=dt)
            if p.numel() != off:
                raise ValueError("Invalid parameter vector length: expected %d, got %d" % (off, p.numel()))
            self._params = p.clone() if clone else p
        else:
            gen = torch.Generator(device="cpu")
            gen.manual_seed(int(ctx.seed))
            host = torch.zeros(off, dtype=dt)
            for e in entries:
                _strided_view(host, e.off, e.spec).copy_(self._init_param(e, gen, dt))
            self._params = host.to(dev)
        self._grads = torch.zeros(off, dtype=dt, device=dev)
        per_layer_p: Dict[int, Dict[str, torch.Tensor]] = {}
        per_layer_g: Dict[int, Dict[str, torch.Tensor]] = {}
        for e in entries:
            per_layer_p.setdefault(e.layer_idx, {})[e.spec.key] = _strided_view(self._params, e.off, e.spec)
            per_layer_g.setdefault(e.layer_idx, {})[e.spec.key] = _strided_view(self._grads, e.off, e.spec)
        for li, l in enumerate(layers):
            l.index = li
            l.set_views(per_layer_p.get(li, {}), per_layer_g.get(li, {}))
        self._build_replicas(layers)
        self._build_updater()
        self._initialized = True

    def _init_param(self, e: ParamEntry, gen, dt):
        spec, conf = e.spec, e.layer.conf
        if spec.init is not None:
            return spec.init(spec.shape, gen, dt)
        if spec.role == "b":
            return torch.full(spec.shape, float(getattr(conf, "biasInit", 0.0) or 0.0), dtype=dt)
        if spec.role == "gain":
            return torch.full(spec.shape, float(getattr(conf, "gainInit", 1.0) or 1.0), dtype=dt)
        wi = getattr(conf, "weightInitFn", None)
        if spec.role == "RW" and getattr(conf, "weightInitFnRecurrent", None) is not None:
            wi = conf.weightInitFnRecurrent
        if wi is None and hasattr(conf, "underlying") and conf.underlying is not None:
            wi = getattr(conf.underlying, "weightInitFn", None)
        if wi is None and hasattr(conf, "fwd") and conf.fwd is not None:
            wi = getattr(conf.fwd, "weightInitFn", None)
        if wi is None:
            from .conf.base import IWeightInit
            wi = IWeightInit("XAVIER")
        if spec.order == "f":
            # DL4J initialises 'f'-ordered weights by filling the flat view in f order
            return wi.init(spec.fan_in, spec.fan_out, spec.shape, gen, dt)
        return wi.init(spec.fan_in, spec.fan_out, spec.shape, gen, dt)

    # ------------------------------------------------------------------ compute replicas (bf16 MFMA path)
    def _build_replicas(self, layers):
        ctx = self.ctx
        self._cast_table = None
        self._cparams_flat = None
        # mixed precision: every requested parameter gets a compute-dtype replica. Same-dtype networks on the GPU (fp32 / fp16
        # masters) still get the PERMUTED filter images the implicit-GEMM kernels read (OHWI forward, flipped [I][k][O] data
        # gradient); plain parameters are used in place.
        layout_only = not ctx.mixed
        if layout_only and (ctx.device.type != "cuda" or ctx.dtype not in (torch.float32, torch.float16)):
            return
        ents = []
        off = 0
        views = []
        by_layer = {}
        for e in self._entries:
            by_layer.setdefault(e.layer_idx, {})[e.spec.key] = e
        for li, l in enumerate(layers):
            for key, layout in l.compute_replicas():
                e = by_layer.get(li, {}).get(key)
                if e is None:
                    continue
                dense_dgrad = layout == "dgrad" and len(e.spec.shape) == 2 and e.spec.order == "f"
                if layout == "dgrad" and len(e.spec.shape) == 2 and not dense_dgrad:
                    continue
                if layout_only and not dense_dgrad and not (len(e.spec.shape) == 4 and (
                        layout == "dgrad" or (layout == "ohwi" and e.spec.shape[2] * e.spec.shape[3] > 1))):
                    continue
                n = e.spec.numel
                off = (off + 7) // 8 * 8      # 16-byte aligned replicas
                if layout == "ohwi" and len(e.spec.shape) == 4:
                    O, I, kh, kw = e.spec.shape
                    ents.append((e.off, off, n, I, kh * kw))
                    views.append((l, key, off, ("ohwi", (O, I, kh, kw))))
                elif dense_dgrad:
                    # 'f'-ordered [nIn, nOut] is [nOut][nIn] in memory; the input-gradient GEMM reads [nIn][nOut]
                    I, O = e.spec.shape
                    ents.append((e.off, off, n, I, 1, O, 1))
                    views.append((l, key + "_dgrad", off, ("dgrad", (O, I, 1, 1))))
                elif layout == "dgrad" and len(e.spec.shape) == 4:
                    # [I][kh*kw flipped][O]: filters as consumed by the dgrad implicit GEMM
                    O, I, kh, kw = e.spec.shape
                    ents.append((e.off, off, n, I, kh * kw, O, 1))
                    views.append((l, key + "_dgrad", off, ("dgrad", (O, I, kh, kw))))
                else:
                    ents.append((e.off, off, n, 1, 1))
                    views.append((l, key, off, ("plain", e.spec)))
                off += n
        self._cparams_flat = torch.zeros(max(off, 1), dtype=ctx.cdtype, device=ctx.device)
        for l, key, o, (kind, info) in views:
            if kind == "ohwi":
                O, I, kh, kw = info
                l.cparams[key] = self._cparams_flat[o:o + O * I * kh * kw].view(O, kh, kw, I).permute(0, 3, 1, 2)
            elif kind == "dgrad":
                O, I, kh, kw = info
                v = self._cparams_flat[o:o + O * I * kh * kw]
                l.cparams[key] = v.view(I, O) if key == "W_dgrad" and len(l.params["W"].shape) == 2 else v.view(I, kh, kw, O)
            else:
                l.cparams[key] = _strided_view(self._cparams_flat, o, info)
        self._cast_table = U.CastTable(ents, ctx.device)
        self._refresh_replicas()

    def _refresh_replicas(self):
        if self._cast_table is not None:
            self._cast_table.run(self._params, self._cparams_flat)

    # ------------------------------------------------------------------ updater
    def _build_updater(self):
        segs: List[U.Segment] = []
        self._seg_updaters: List[IUpdater] = []
        state_off = 0
        cur_key = None
        it, ep = self.iterationCount, self.epochCount
        self._has_schedule = False
        for e in self._entries:
            upd = e.updater
            l1 = sum(r.l1.valueAt(it, ep) for r in e.regs if isinstance(r, L1Regularization))
            l2 = sum(r.l2.valueAt(it, ep) for r in e.regs if isinstance(r, L2Regularization))
            wds = [r for r in e.regs if isinstance(r, WeightDecay)]
            wd = sum(r.coeff.valueAt(it, ep) for r in wds)
            wd_lr = all(r.applyLR for r in wds) if wds else True
            key = (type(upd).__name__, repr(upd.to_json()), l1, l2, wd, wd_lr, e.divide)
            n = e.spec.numel
            if getattr(upd, "learningRateSchedule", None) is not None:
                self._has_schedule = True
            if key == cur_key and segs and segs[-1].off + segs[-1].length == e.off:
                segs[-1].length += n
            else:
                lr, a, b, eps = upd.segment_hparams(it, ep)
                segs.append(U.Segment(e.off, n, 0, U.KIND[upd.KIND], lr, a, b, eps, l1, l2, wd, e.divide, wd_lr))
                self._seg_updaters.append(upd)
                cur_key = key
        for s in segs:
            s.state_off = state_off
            state_off += U.STATE_MULT[s.kind] * s.length
        self._updater_state = torch.zeros(max(state_off, 1), dtype=self.ctx.dtype, device=self.ctx.device)
        self._state_size = state_off
        for s, u in zip(segs, self._seg_updaters):
            v = u.state_init_value()
            if v != 0.0 and U.STATE_MULT[s.kind]:
                self._updater_state[s.state_off:s.state_off + U.STATE_MULT[s.kind] * s.length].fill_(v)
        self._updater = U.SegmentedUpdater(segs, self.ctx.device)
        self._updater.set_iteration(self.iterationCount)
        if self._has_schedule:
            ups = list(self._seg_updaters)
            self._updater.set_schedule(lambda it, ep: [u.getLearningRate(it, ep) if u.hasLearningRate() else None for u in ups])
        self._gn_table = self._build_grad_norm_table()

    def _build_grad_norm_table(self):
        """Ranges of the flat gradient vector for the configured GradientNormalization modes (one per layer or per parameter)."""
        by_layer = {}
        for e in self._entries:
            by_layer.setdefault(id(e.layer), []).append(e)
        ranges = []
        for l in self._ordered_layers():
            gn = getattr(l.conf, "gradientNormalization", None)
            if not gn or gn == GradientNormalization.NONE:
                continue
            ents = by_layer.get(id(l), [])
            if not ents:
                continue
            name = gn if isinstance(gn, str) else getattr(gn, "name", str(gn))
            if name not in U.GradNormTable.MODES:
                return None
            mode, scope = U.GradNormTable.MODES[name]
            thr = getattr(l.conf, "gradientNormalizationThreshold", 1.0) or 1.0
            if scope == "param":
                ranges += [(e.off, e.spec.numel, mode, thr) for e in ents]
            else:
                lo = min(e.off for e in ents)
                hi = max(e.off + e.spec.numel for e in ents)
                if hi - lo != sum(e.spec.numel for e in ents):
                    return None                       # layer not contiguous in the flat vector: per-layer eager path
                ranges.append((lo, hi - lo, mode, thr))
        return U.GradNormTable(ranges, self.ctx.device)

    def getUpdaterBlocks(self):
        """[(offset, length, IUpdater)] — UpdaterBlock structure of the flat vector."""
        return [(s.off, s.length, u) for s, u in zip(self._updater.segments, self._seg_updaters)]

    def _apply_update(self, batch_size, sync=True):
        if sync and self._grad_sync is not None:
            self._grad_sync()
        if self._has_schedule and not (self._params.is_cuda and torch.cuda.is_current_stream_capturing()):
            self._updater.ensure_schedule(self.iterationCount, self.epochCount)
        self._pre_apply_gradient_normalization()
        mb = batch_size if getattr(self, "_minibatch", True) else 1
        if self._params.dtype == torch.float32 or (self._params.is_cuda and self._params.dtype in (torch.bfloat16, torch.float16)
                                                   and self._grads.dtype == self._params.dtype):
            self._updater.step(self._params, self._grads, self._updater_state, mb)
        else:
            t = float(self.iterationCount + 1)
            for s in self._updater.segments:
                U.apply_segment_reference(s, self._params, self._grads, self._updater_state, 1.0 / max(mb, 1), t)
            self._updater.iter_dev += 1
        self._refresh_replicas()

    def _pre_apply_gradient_normalization(self):
        if getattr(self, "_gn_table", None) is not None:
            self._gn_table.run(self._grads)
            return
        for l in self._ordered_layers():
            gn = getattr(l.conf, "gradientNormalization", None)
            if not gn or gn == GradientNormalization.NONE or not l.grads:
                continue
            thr = getattr(l.conf, "gradientNormalizationThreshold", 1.0) or 1.0
            gs = list(l.grads.values())
            if gn == GradientNormalization.RenormalizeL2PerLayer:
                nrm = torch.sqrt(sum((g.double() ** 2).sum() for g in gs)).clamp_min(1e-12)
                for g in gs:
                    g.div_(nrm.to(g.dtype))
            elif gn == GradientNormalization.RenormalizeL2PerParamType:
                for g in gs:
                    g.div_(g.norm().clamp_min(1e-12))
            elif gn == GradientNormalization.ClipElementWiseAbsoluteValue:
                for g in gs:
                    g.clamp_(-thr, thr)
            elif gn == GradientNormalization.ClipL2PerLayer:
                nrm = torch.sqrt(sum((g.double() ** 2).sum() for g in gs))
                sc = torch.where(nrm > thr, thr / nrm.clamp_min(1e-12), torch.ones_like(nrm))
                for g in gs:
                    g.mul_(sc.to(g.dtype))
            elif gn == GradientNormalization.ClipL2PerParamType:
                for g in gs:
                    nrm = g.norm()
                    g.mul_(torch.where(nrm > thr, thr / nrm.clamp_min(1e-12), torch.ones_like(nrm)))

    # ------------------------------------------------------------------ public parameter API
    def params(self):
        return self._params.view(1, -1)

    def numParams(self):
        return self._num_params

    def setParams(self, p):
        self._params.copy_(torch.as_tensor(p).reshape(-1).to(self._params.dtype))
        self._refresh_replicas()

    setParameters = setParams

    def getFlattenedGradients(self):
        return self._grads.view(1, -1)

    def gradient(self):
        from .gradient import Gradient
        return Gradient(self)

    def paramTable(self):
        out = {}
        for e in self._entries:
            out["%s_%s" % (self._layer_key(e.layer_idx), e.spec.key)] = _strided_view(self._params, e.off, e.spec)
        return out

    def getParam(self, name):
        return self.paramTable()[name]

    def setParam(self, name, value):
        self.paramTable()[name].copy_(torch.as_tensor(value))
        self._refresh_replicas()

    def _layer_key(self, idx):
        return str(idx)

    def getUpdaterState(self):
        return self._updater_state[:self._state_size].view(1, -1) if self._state_size else None

    def setUpdaterState(self, st):
        if st is not None and self._state_size:
            self._updater_state[:self._state_size].copy_(torch.as_tensor(st).reshape(-1).to(self._updater_state.dtype))

    def getLearningRate(self, layer_idx=0):
        for (off, n, u), s in zip(self.getUpdaterBlocks(), self._updater.segments):
            for e in self._entries:
                if e.layer_idx == layer_idx and off <= e.off < off + n:
                    return s.lr
        return float("nan")

    def setLearningRate(self, lr, schedule=None):
        for u in self._seg_updaters:
            if u.hasLearningRate():
                u.setLrAndSchedule(lr, schedule)
        self._has_schedule = schedule is not None
        self._updater.set_learning_rates([lr if u.hasLearningRate() else None for u in self._seg_updaters])

    # ------------------------------------------------------------------ score
    def calcRegularizationScore(self, backpropParamsOnly=True):
        it, ep = self.iterationCount, self.epochCount
        tot = None
        for e in self._entries:
            if not e.regs:
                continue
            p = _strided_view(self._params, e.off, e.spec)
            for r in e.regs:
                s = r.score(p.float() if p.dtype == torch.bfloat16 else p, it, ep)
                tot = s if tot is None else tot + s
        return tot

    def score(self, dataset=None, training=False):
        if dataset is not None:
            return self._score_dataset(dataset, training)
        if self._score is None:
            return 0.0
        s = self._score
        r = self.calcRegularizationScore()
        if r is not None:
            s = s + r.to(s.dtype)
        return float(s)

    def score_tensor(self):
        """Device-resident score of the last iteration (no host sync)."""
        return self._score

    # ------------------------------------------------------------------ listeners / counters
    def setListeners(self, *ls):
        self.listeners = []
        self.addListeners(*ls)

    def addListeners(self, *ls):
        for l in ls:
            if isinstance(l, (list, tuple)):
                self.listeners.extend(l)
            elif l is not None:
                self.listeners.append(l)

    def getListeners(self):
        return self.listeners

    def getIterationCount(self):
        return self.iterationCount

    def getEpochCount(self):
        return self.epochCount

    def setIterationCount(self, n):
        self.iterationCount = int(n)
        self.conf.iterationCount = int(n)
        self.ctx.iteration = int(n)
        if self._initialized:
            self._updater.set_iteration(n)

    def setEpochCount(self, n):
        self.epochCount = int(n)
        self.conf.epochCount = int(n)
        self.ctx.epoch = int(n)

    def incrementEpochCount(self):
        self.setEpochCount(self.epochCount + 1)

    def _ws_scope(self, training: bool):
        """Workspace scope of one eager iteration / inference call (MultiLayerNetwork.java:1688-1705: LayerWorkspaceMgr with
        WS_ALL_LAYERS_ACT & co. unless the WorkspaceMode is NONE). On the GPU the manager's workspaces are private allocator pools
        (nd4j/workspace.py DevicePoolWorkspace): everything the iteration allocates is recycled pool memory. CPU runs, captured
        HIP-graph steps (static graph pool) and WorkspaceMode.NONE get a null scope."""
        import contextlib
        from ..nd4j.workspace import ArrayType, DevicePoolWorkspace, LayerWorkspaceMgr
        from .conf.base import WorkspaceMode
        mode = self.conf.trainingWorkspaceMode if training else self.conf.inferenceWorkspaceMode
        if (str(mode).upper() == str(WorkspaceMode.NONE).upper() or not self._initialized
                or not DevicePoolWorkspace.supported(self.ctx.device) or os.environ.get("DL4J_AMD_WORKSPACES", "1") == "0"):
            return contextlib.nullcontext()
        key = "_ws_mgr_train" if training else "_ws_mgr_infer"
        mgr = getattr(self, key, None)
        if mgr is None:
            mgr = LayerWorkspaceMgr.forNetwork(training, self.ctx.device)
            setattr(self, key, mgr)
        return mgr.notifyScopeEntered(ArrayType.ACTIVATIONS)

    def getLayerWorkspaceMgr(self, training=True):
        return getattr(self, "_ws_mgr_train" if training else "_ws_mgr_infer", None)

    def _after_iteration(self):
        self.iterationCount += 1
        self.conf.iterationCount = self.iterationCount
        self.ctx.iteration = self.iterationCount
        for l in self.listeners:
            l.iterationDone(self, self.iterationCount, self.epochCount)

    def batchSize(self):
        return self._last_batch

    def getInputMiniBatchSize(self):
        return self._last_batch

    # ------------------------------------------------------------------ device / dtype helpers
    def _to_dev(self, x, floating=True):
        if x is None:
            return None
        if hasattr(x, "tensor"):
            x = x.tensor()
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(x)
        if x.device != self.ctx.device:
            x = x.to(self.ctx.device, non_blocking=True)
        return x

    def _prep_features(self, x):
        x = self._to_dev(x)
        if (x.dim() == 4 and self.ctx.channels_last and x.is_cuda and x.dtype == torch.float32 and x.shape[1] <= 4 and x.is_contiguous()
                and self.ctx.cdtype in (torch.bfloat16, torch.float32) and not x.requires_grad):
            # image batches (NCHW fp32 from the iterator) -> NHWC compute type in one pass instead of a cast and a permuting copy
            from ..ops import spatial as SP
            y = SP.nchw_f32_to_nhwc(x, self.ctx.cdtype)
            if y is not None:
                return y
        if x.is_floating_point() and x.dtype != self.ctx.cdtype:
            x = x.to(self.ctx.cdtype)
        if x.dim() == 4 and self.ctx.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        return x

    def _prep_labels(self, y):
        # labels stay in the NETWORK dtype (fp32 master): only activations / weights are rounded to the bf16 / fp16 compute
        # type; the fused softmax-xent kernel casts where it needs matching operands (ops/softmax.py)
        y = self._to_dev(y)
        if y is not None and y.is_floating_point() and y.dtype != self.ctx.dtype:
            y = y.to(self.ctx.dtype)
        return y

    def close(self):
        self._params = self._grads = self._updater_state = None
        self._graph_step = None
