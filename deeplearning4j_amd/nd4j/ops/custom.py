"""Custom (declarable) op framework: registry, DynamicCustomOp, OpContext, OpExecutioner.

Reference: libnd4j ops/declarable/{DeclarableOp.h:67-225, OpRegistrator.h:170-181, impl/DeclarableOp.cpp:662-864},
nd4j-api linalg/api/ops/{DynamicCustomOp,OpContext,CustomOp}.java, executioner/OpExecutioner.java,
nd4j-cuda CudaExecutioner.exec(CustomOp) :1801-1859.

Execution pipeline kept from the reference: validate arg counts against the declared contract
(native registry, rt_ops.cpp — names/hashes/contracts are ABI) -> platform helper if usable (the
hand-written gfx950 kernels registered with ``@platform``) -> generic implementation -> optional
profiling / NaN-Inf checks (OpProfiler). Shape functions (calculateOutputShape) run the op on
``meta`` tensors, so every op gets shape inference without a second implementation.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Dict, List, Optional, Sequence

import torch

from ...native import lib as L
from ..environment import Environment, OpProfiler

_IMPL: Dict[str, Callable] = {}
_PLATFORM: Dict[str, List] = {}
_LOADED = False

# Platform-helper accounting (reference: the helperCountFail / "helper not used" diagnostics of the cuDNN / oneDNN platform ops,
# ops/declarable/PlatformHelper.h + Environment::helpersAllowed): every time an op that HAS a hand-written HIP route is executed
# on GPU tensors, either the route ran ("hit") or the generic ATen implementation did ("fail", with the reason). Nothing falls back
# unseen: OpExecutioner.helperStatistics() returns the counters, DL4J_AMD_LOG_FALLBACKS=1 logs each (op, reason) once.
import collections
import logging
import os

HELPER_HIT: Dict[str, int] = collections.Counter()
HELPER_FAIL: Dict[tuple, int] = collections.Counter()
_LOG_FALLBACKS = os.environ.get("DL4J_AMD_LOG_FALLBACKS", "0") == "1"
_log = logging.getLogger("deeplearning4j_amd.ops")


def _helper_fail(name, reason):
    key = (name, reason)
    if _LOG_FALLBACKS and key not in HELPER_FAIL:
        _log.warning("op %s: platform helper not used (%s) - generic implementation runs", name, reason)
    HELPER_FAIL[key] += 1


def op(*names):
    """Register the generic implementation: fn(ins, i, t, b, d) -> tensor | tuple of tensors."""
    def deco(fn):
        for n in names:
            _IMPL[n] = fn
        return fn
    return deco


def platform(name, usable, differentiable=False):
    """Register a platform helper (hand-written HIP path) with its ``isUsable(ins, i, t, b, d)`` predicate.
    ``differentiable``: the helper is an autograd-aware function (HIP kernels in both directions), so it may also run while a
    SameDiff training session records the graph; other helpers are skipped as soon as an input requires grad."""
    def deco(fn):
        fn._differentiable = bool(differentiable)
        _PLATFORM.setdefault(name, []).append((usable, fn))
        return fn
    return deco


def _cost_model():
    from . import cost_model
    return cost_model


def _load_impls():
    global _LOADED
    if not _LOADED:
        _LOADED = True
        from . import impl_core, impl_nn, impl_misc, impl_platform  # noqa: F401  (populate the registry)


class OpDescriptor:
    def __init__(self, name, nIn, nOut, inplace, nT, nI, hash_):
        self.name, self.numInputs, self.numOutputs, self.inplace, self.numTArgs, self.numIArgs, self.hash = (
            name, nIn, nOut, bool(inplace), nT, nI, hash_)

    def __repr__(self):
        return "OpDescriptor(%s, in=%d, out=%d, tArgs=%d, iArgs=%d, hash=%d)" % (
            self.name, self.numInputs, self.numOutputs, self.numTArgs, self.numIArgs, self.hash)


def describe(name) -> Optional[OpDescriptor]:
    out = (ctypes.c_int * 5)()
    h = ctypes.c_int64(0)
    if not L.rt().ops_describe(name.encode(), out, ctypes.byref(h)):
        return None
    return OpDescriptor(name, out[0], out[1], out[2], out[3], out[4], h.value)


def op_hash(name) -> int:
    return int(L.rt().ops_hash(name.encode()))


def name_by_hash(h) -> Optional[str]:
    r = L.rt().ops_name_by_hash(ctypes.c_int64(h))
    return r.decode() if r else None


class OpContext:
    """Reusable argument carrier (reference: graph::Context as used by Java OpContext)."""

    def __init__(self):
        self.inputs: List[torch.Tensor] = []
        self.outputs: List[torch.Tensor] = []
        self.iArgs: List[int] = []
        self.tArgs: List[float] = []
        self.bArgs: List[bool] = []
        self.dArgs: List[str] = []
        self.inplace = False
        self.allowHelpers = True
        self.rng = None

    def setInputArrays(self, *a):
        self.inputs = [_t(x) for x in (a[0] if len(a) == 1 and isinstance(a[0], (list, tuple)) else a)]
        return self

    def setInputArray(self, idx, a):
        while len(self.inputs) <= idx:
            self.inputs.append(None)
        self.inputs[idx] = _t(a)
        return self

    def setOutputArrays(self, *a):
        self.outputs = [_t(x) for x in (a[0] if len(a) == 1 and isinstance(a[0], (list, tuple)) else a)]
        return self

    def setOutputArray(self, idx, a):
        while len(self.outputs) <= idx:
            self.outputs.append(None)
        self.outputs[idx] = _t(a)
        return self

    def setIArguments(self, *a):
        self.iArgs = [int(v) for v in (a[0] if len(a) == 1 and isinstance(a[0], (list, tuple)) else a)]
        return self

    def setTArguments(self, *a):
        self.tArgs = [float(v) for v in (a[0] if len(a) == 1 and isinstance(a[0], (list, tuple)) else a)]
        return self

    def setBArguments(self, *a):
        self.bArgs = [bool(v) for v in (a[0] if len(a) == 1 and isinstance(a[0], (list, tuple)) else a)]
        return self

    def setDArguments(self, *a):
        self.dArgs = list(a[0] if len(a) == 1 and isinstance(a[0], (list, tuple)) else a)
        return self

    def getInputArray(self, i):
        return self.inputs[i]

    def getOutputArray(self, i):
        return self.outputs[i]

    def numInputArguments(self):
        return len(self.inputs)

    def numOutputArguments(self):
        return len(self.outputs)

    def markInplace(self, b=True):
        self.inplace = b

    def setAllowHelpers(self, b):
        self.allowHelpers = bool(b)

    def purge(self):
        self.__init__()


def _t(x):
    if x is None:
        return None
    if getattr(x, "_nd4j_opaque", False) or type(x).__name__ == "StringArray":
        return x                      # NDArrayList handles and string arrays travel through op inputs / outputs as they are
    if hasattr(x, "tensor"):
        return x.tensor()
    if isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(x)


class DynamicCustomOp:
    """Name + inputs/outputs + i/t/b/d args (reference: linalg/api/ops/DynamicCustomOp.java)."""

    def __init__(self, opName, inputs=None, outputs=None, iArgs=None, tArgs=None, bArgs=None, dArgs=None, inplace=False):
        self.opName = opName
        self.ctx = OpContext()
        self.ctx.inputs = [_t(x) for x in (inputs or [])]
        self.ctx.outputs = [_t(x) for x in (outputs or [])]
        self.ctx.iArgs = [int(v) for v in (iArgs or [])]
        self.ctx.tArgs = [float(v) for v in (tArgs or [])]
        self.ctx.bArgs = [bool(v) for v in (bArgs or [])]
        self.ctx.dArgs = list(dArgs or [])
        self.ctx.inplace = inplace

    class Builder:
        def __init__(self, name):
            self._op = DynamicCustomOp(name)

        def addInputs(self, *a):
            self._op.ctx.inputs += [_t(x) for x in a]
            return self

        def addOutputs(self, *a):
            self._op.ctx.outputs += [_t(x) for x in a]
            return self

        def addIntegerArguments(self, *a):
            self._op.ctx.iArgs += [int(v) for v in a]
            return self

        def addFloatingPointArguments(self, *a):
            self._op.ctx.tArgs += [float(v) for v in a]
            return self

        def addBooleanArguments(self, *a):
            self._op.ctx.bArgs += [bool(v) for v in a]
            return self

        def addDataTypeArguments(self, *a):
            self._op.ctx.dArgs += list(a)
            return self

        def callInplace(self, b=True):
            self._op.ctx.inplace = b
            return self

        def build(self):
            return self._op

    @staticmethod
    def builder(name):
        return DynamicCustomOp.Builder(name)

    def opName_(self):
        return self.opName

    def opHash(self):
        return op_hash(self.opName)

    def getDescriptor(self):
        return describe(self.opName)

    def inputArguments(self):
        return self.ctx.inputs

    def outputArguments(self):
        return self.ctx.outputs

    def iArgs(self):
        return self.ctx.iArgs

    def tArgs(self):
        return self.ctx.tArgs

    def bArgs(self):
        return self.ctx.bArgs

    def calculateOutputShape(self):
        return OpExecutioner.getInstance().calculateOutputShape(self)

    def exec(self):
        return OpExecutioner.getInstance().exec(self)


USER_DEFINED_OPS = set()        # names registered at run time by UserDefinedCustomOp (not part of the built-in op set)


class UserDefinedCustomOp(DynamicCustomOp):
    """UDF ops (ADR 0023): subclass, implement ``doExec(ins, i, t, b, d)``; registered at first use."""

    def doExec(self, ins, i, t, b, d):
        raise NotImplementedError

    def exec(self):
        if self.opName not in _IMPL:
            L.rt().ops_register(self.opName.encode(), -1, -1, 0, -1, -1)
            USER_DEFINED_OPS.add(self.opName)
            _IMPL[self.opName] = lambda ins, i, t, b, d, _s=self: _s.doExec(ins, i, t, b, d)
        return super().exec()


class OpExecutioner:
    _inst = None

    def __init__(self):
        self.profiler = OpProfiler.getInstance()
        self.env = Environment.getInstance()

    @classmethod
    def getInstance(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    # -- registry views
    def getCustomOperations(self) -> Dict[str, OpDescriptor]:
        s = L.rt().ops_all_custom().decode()
        out = {}
        for rec in s.split(";"):
            if not rec:
                continue
            n, h, a, b_, c, d_, e = rec.split(":")
            out[n] = OpDescriptor(n, int(a), int(b_), int(c), int(d_), int(e), int(h))
        return out

    def implementedOps(self):
        _load_impls()
        return sorted(_IMPL)

    def missingOps(self):
        _load_impls()
        return sorted(set(self.getCustomOperations()) - set(_IMPL))

    # -- legacy families
    def legacy(self):
        from . import legacy
        return legacy.executioner()

    # -- execution
    @staticmethod
    def _validate(name, desc: Optional[OpDescriptor], ctx: OpContext):
        if desc is None:
            return
        nin = len([x for x in ctx.inputs if x is not None])
        if desc.numInputs > 0 and nin < desc.numInputs:
            raise ValueError("op %s expects at least %d inputs, got %d" % (name, desc.numInputs, nin))
        if desc.numTArgs > 0 and len(ctx.tArgs) < desc.numTArgs:
            raise ValueError("op %s expects at least %d tArgs, got %d" % (name, desc.numTArgs, len(ctx.tArgs)))
        if desc.numIArgs > 0 and len(ctx.iArgs) < desc.numIArgs:
            raise ValueError("op %s expects at least %d iArgs, got %d" % (name, desc.numIArgs, len(ctx.iArgs)))

    def _run(self, name, ins, i, t, b, d, allow_helpers=True):
        _load_impls()
        if name in _PLATFORM:
            on_gpu = any(isinstance(x, torch.Tensor) and x.is_cuda for x in ins)
            if allow_helpers and self.env.helpersAllowed():
                tracked = torch.is_grad_enabled() and any(isinstance(x, torch.Tensor) and x.requires_grad for x in ins)
                declined = None
                for usable, fn in _PLATFORM[name]:
                    if (not tracked or getattr(fn, "_differentiable", False)) and usable(ins, i, t, b, d):
                        # measured cost model (cost_model.py): usable, but the generic route is faster for these arguments
                        declined = None if tracked else _cost_model().prefer_generic(name, ins, i, t, b, d)
                        if declined is not None:
                            continue
                        HELPER_HIT[name] += 1
                        return fn(ins, i, t, b, d)
                if declined is not None:
                    _helper_fail(name, "cost model: " + declined)
                elif not ins:
                    _helper_fail(name, "no array inputs: the helper cannot tell the device")
                elif on_gpu:
                    _helper_fail(name, "autograd-tracked inputs" if tracked and not any(getattr(f, "_differentiable", False) for _, f in _PLATFORM[name])
                                 else "isUsable predicate rejected the arguments")
            elif on_gpu and allow_helpers:
                _helper_fail(name, "helpers disabled (Environment.allowHelpers=false)")
        if name not in _IMPL:
            canon = name_by_hash(op_hash(name))         # DECLARE_SYN aliases resolve to the canonical op (mmul -> matmul)
            if canon is None or canon not in _IMPL:
                raise NotImplementedError("custom op '%s' has no implementation in this build" % name)
            _IMPL[name] = _IMPL[canon]
            if canon in _PLATFORM:
                _PLATFORM.setdefault(name, _PLATFORM[canon])
                for usable, fn in _PLATFORM[name]:
                    tracked = torch.is_grad_enabled() and any(isinstance(x, torch.Tensor) and x.requires_grad for x in ins)
                    if allow_helpers and self.env.helpersAllowed() and (not tracked or getattr(fn, "_differentiable", False)) \
                            and usable(ins, i, t, b, d):
                        return fn(ins, i, t, b, d)
        return _IMPL[name](ins, i, t, b, d)

    def exec(self, op_or_name, *args, **kw):
        """exec(DynamicCustomOp) / exec(name, inputs, iArgs=..., tArgs=..., bArgs=..., dArgs=...) / exec(op, OpContext) /
        exec(Aggregate) / exec(Batch) (OpExecutioner.java: exec(Aggregate), exec(Batch<T>) -> execAggregate / execAggregateBatch)."""
        from . import aggregates as _agg
        if isinstance(op_or_name, _agg.Batch):
            return _agg.exec_batch(op_or_name)
        if isinstance(op_or_name, _agg.Aggregate):
            return _agg.exec_aggregate(op_or_name)
        if isinstance(op_or_name, (list, tuple)) and op_or_name and all(isinstance(a, _agg.Aggregate) for a in op_or_name):
            for bt in _agg.Batch.getBatches(list(op_or_name)):          # exec(List<Aggregate>)
                _agg.exec_batch(bt)
            return None
        if isinstance(op_or_name, str):
            o = DynamicCustomOp(op_or_name, args[0] if args else kw.get("inputs"), kw.get("outputs"), kw.get("iArgs"),
                                kw.get("tArgs"), kw.get("bArgs"), kw.get("dArgs"))
        else:
            o = op_or_name
            if args and isinstance(args[0], OpContext):
                o = DynamicCustomOp(o.opName)
                o.ctx = args[0]
        ctx, name = o.ctx, o.opName
        self._validate(name, describe(name), ctx)
        ins = ctx.inputs
        with self.profiler.op(name, *ins):
            res = self._run(name, ins, ctx.iArgs, ctx.tArgs, ctx.bArgs, ctx.dArgs, ctx.allowHelpers)
        outs = list(res) if isinstance(res, (tuple, list)) else [res]
        self.profiler.checkOutputs(name, *outs)
        if ctx.outputs:
            for dst, src in zip(ctx.outputs, outs):
                if dst is not None and src is not None and dst is not src:
                    dst.copy_(src.reshape(dst.shape) if src.numel() == dst.numel() else src)
            outs = [dst if dst is not None else src for dst, src in zip(ctx.outputs, outs)] + outs[len(ctx.outputs):]
        elif ctx.inplace and outs and ins and isinstance(outs[0], torch.Tensor) and isinstance(ins[0], torch.Tensor) \
                and outs[0].shape == ins[0].shape and outs[0].dtype == ins[0].dtype:
            ins[0].copy_(outs[0])
            outs[0] = ins[0]
        ctx.outputs = outs
        return outs

    def execAndReturn(self, op_):
        return self.exec(op_)[0]

    @staticmethod
    def helperStatistics():
        """{'hit': {op: n}, 'fail': {(op, reason): n}, 'legacy': {(family, reason): n}} since the last reset."""
        from . import legacy
        return {"hit": dict(HELPER_HIT), "fail": dict(HELPER_FAIL), "legacy": dict(legacy.FALLBACKS)}

    @staticmethod
    def resetHelperStatistics():
        from . import legacy
        HELPER_HIT.clear()
        HELPER_FAIL.clear()
        legacy.FALLBACKS.clear()

    def calculateOutputShape(self, op_) -> List[List[int]]:
        """Shape inference by meta execution (DECLARE_SHAPE_FN analogue)."""
        ctx = op_.ctx
        meta = [None if x is None else torch.empty(x.shape, dtype=x.dtype, device="meta") for x in ctx.inputs]
        try:
            res = self._run(op_.opName, meta, ctx.iArgs, ctx.tArgs, ctx.bArgs, ctx.dArgs, allow_helpers=False)
        except Exception:
            # data-dependent shapes (unique, where, ...) need real values
            res = self._run(op_.opName, ctx.inputs, ctx.iArgs, ctx.tArgs, ctx.bArgs, ctx.dArgs, allow_helpers=False)
        outs = list(res) if isinstance(res, (tuple, list)) else [res]
        return [list(o.shape) for o in outs]

    def commit(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def setProfilingConfig(self, cfg):
        self.profiler.setConfig(cfg)

    def enableDebugMode(self, b):
        self.env.setDebug(b)

    def enableVerboseMode(self, b):
        self.env.setVerbose(b)

    def printEnvironmentInformation(self):
        info = {"backend": "gfx950-hip" if torch.cuda.is_available() else "cpu-torch", "torch": torch.__version__,
                "hip": torch.version.hip, "devices": torch.cuda.device_count() if torch.cuda.is_available() else 0,
                "customOps": L.rt().ops_count()}
        print(info)
        return info


def exec_op(name, inputs, iArgs=None, tArgs=None, bArgs=None, dArgs=None):
    """Convenience: run a custom op by name, return the list of outputs."""
    return OpExecutioner.getInstance().exec(name, inputs, iArgs=iArgs, tArgs=tArgs, bArgs=bArgs, dArgs=dArgs)
