"""SameDiff: define-then-run graph container, variables, sessions, training.

Reference: nd4j-api autodiff/samediff/{SameDiff,SDVariable,VariableType,TrainingConfig}.java,
internal/{AbstractSession,InferenceSession,TrainingSession,SameDiffOp,Variable}.java, History/LossCurve.

MI355X-first design decisions
* Graph nodes name ops of the native registry (custom ops by name/hash, legacy ops by (OpType, opNum)) —
  the same ABI the FlatBuffers file stores — and execute through the nd4j executioners, i.e. the
  hand-written gfx950 kernels on GPU tensors.
* Execution is a dependency-driven dataflow interpreter with TF-style frames (Enter / Exit / Merge /
  Switch / NextIteration / LoopCond), like AbstractSession, so imported control flow runs unchanged.
* Gradients: the reference builds a second graph by calling every op's doDiff. Here the training
  session runs the forward graph under local autograd on device tensors and asks for d(loss)/d(vars)
  — one backward sweep fused by ATen, no Java-side graph interpretation on the hot path.
  ``sd.grad(name)`` / ``calculateGradients`` expose the same information as the reference API.
"""
from __future__ import annotations

import json
import math
import time
from collections import OrderedDict, defaultdict, deque
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Union

import os

import numpy as np
import torch

from ..nd4j import learning as LR
from ..nd4j.ndarray import DataType, INDArray
from ..nd4j.ops import legacy as LG
from ..nd4j.ops.custom import OpExecutioner, op_hash


class VariableType:
    VARIABLE, CONSTANT, ARRAY, PLACEHOLDER = "VARIABLE", "CONSTANT", "ARRAY", "PLACEHOLDER"


# FlatBuffers OpType values (graph/scheme/utils.fbs:43-71)
class OpType:
    TRANSFORM_FLOAT, TRANSFORM_SAME, TRANSFORM_BOOL, TRANSFORM_STRICT, TRANSFORM_ANY = 0, 1, 2, 3, 4
    REDUCE_FLOAT, REDUCE_SAME, REDUCE_LONG, REDUCE_BOOL, INDEX_REDUCE = 5, 6, 7, 8, 9
    SCALAR, SCALAR_BOOL, BROADCAST, BROADCAST_BOOL, PAIRWISE, PAIRWISE_BOOL, REDUCE_3, SUMMARYSTATS = 10, 11, 12, 13, 14, 15, 16, 17
    SHAPE, AGGREGATION, RANDOM, CUSTOM, GRAPH, VARIABLE, BOOLEAN, LOGIC, UDF = 18, 19, 20, 21, 22, 40, 60, 119, 120


# LOGIC opNums (system/op_enums.h:92-105)
LOGIC = {0: "while", 10: "scope", 20: "conditional", 30: "switch", 40: "return", 60: "merge", 70: "loop_cond",
         80: "next_iteration", 90: "exit", 100: "enter"}
LOGIC_NUM = {v: k for k, v in LOGIC.items()}
_CONTROL = {"switch", "merge", "enter", "exit", "next_iteration", "loop_cond"}


def _tt(x, device=None, dtype=None):
    if isinstance(x, INDArray):
        x = x.tensor()
    if not isinstance(x, torch.Tensor):
        a = np.asarray(x)
        x = torch.as_tensor(a)
        if x.dtype == torch.float64 and dtype is None:
            x = x.float()
    if dtype is not None:
        x = x.to(dtype)
    if device is not None:
        x = x.to(device)
    return x


class SDVariable:
    def __init__(self, sd: "SameDiff", name: str, vtype: str, dtype=None, shape=None):
        self.sd, self.name, self.variableType = sd, name, vtype
        self.dtype = dtype
        self.shape = None if shape is None else list(shape)

    # -- info
    def getVarName(self):
        return self.name

    name_ = getVarName

    def dataType(self):
        return None if self.dtype is None else DataType.from_torch(self.dtype)

    def getShape(self):
        a = self.sd._arrays.get(self.name)
        if a is not None:
            return list(a.shape)
        if self.shape is None and self.variableType == VariableType.ARRAY:
            return self.sd.inferShape(self.name)
        return self.shape

    def isPlaceHolder(self):
        return self.variableType == VariableType.PLACEHOLDER

    def isConstant(self):
        return self.variableType == VariableType.CONSTANT

    def getArr(self):
        a = self.sd._arrays.get(self.name)
        return None if a is None else INDArray(a)

    def setArray(self, arr):
        self.sd.setArrayForVariable(self.name, arr)
        return self

    def eval(self, placeholders=None):
        return INDArray(self.sd.output(placeholders or {}, [self.name])[self.name])

    def gradient(self):
        return self.sd.grad(self.name)

    getGradient = gradient

    def rename(self, newName):
        self.sd.renameVariable(self.name, newName)
        return self

    def markAsLoss(self):
        self.sd.addLossVariable(self.name)
        return self

    def convertToConstant(self):
        self.sd.convertToConstants([self.name])
        return self

    def convertToVariable(self):
        self.sd.convertToVariables([self.name])
        return self

    def addControlDependency(self, other):
        op_ = self.sd._producer.get(self.name)
        if op_ is not None:
            self.sd.ops[op_].controlDeps.append(other.name)
        return self

    # -- arithmetic sugar
    def _b(self, opn, o, rev=False):
        o = o if isinstance(o, SDVariable) else self.sd.constant(None, o)
        a, b = (o, self) if rev else (self, o)
        return self.sd._op(opn, [a, b])[0]

    def add(self, o): return self._b("add", o)
    def sub(self, o): return self._b("subtract", o)
    def mul(self, o): return self._b("multiply", o)
    def div(self, o): return self._b("divide", o)
    def rsub(self, o): return self._b("subtract", o, True)
    def rdiv(self, o): return self._b("divide", o, True)
    def pow(self, o): return self._b("Pow", o)
    def mmul(self, o, transposeA=False, transposeB=False): return self.sd._op("matmul", [self, o], iArgs=[int(transposeA), int(transposeB), 0])[0]
    def dot(self, o, *dims): return self.sd._op("reduce_sum", [self.mul(o)], iArgs=list(dims))[0]
    def neg(self): return self.sd.math.neg(self)
    def gt(self, o): return self._b("greater", o)
    def gte(self, o): return self._b("greater_equal", o)
    def lt(self, o): return self._b("less", o)
    def lte(self, o): return self._b("less_equal", o)
    def eq(self, o): return self._b("equals", o)
    def neq(self, o): return self._b("not_equals", o)
    __add__ = add
    __radd__ = add
    __sub__ = sub
    __rsub__ = rsub
    __mul__ = mul
    __rmul__ = mul
    __truediv__ = div
    __rtruediv__ = rdiv
    __neg__ = neg
    __matmul__ = mmul
    __pow__ = pow

    def _r(self, opn, dims, keepDims=False):
        dims = list(dims[0]) if len(dims) == 1 and isinstance(dims[0], (list, tuple)) else list(dims)
        return self.sd._op(opn, [self], iArgs=dims, bArgs=[keepDims])[0]

    def sum(self, *dims, keepDims=False): return self._r("reduce_sum", dims, keepDims)
    def mean(self, *dims, keepDims=False): return self._r("reduce_mean", dims, keepDims)
    def max(self, *dims, keepDims=False): return self._r("reduce_max", dims, keepDims)
    def min(self, *dims, keepDims=False): return self._r("reduce_min", dims, keepDims)
    def prod(self, *dims, keepDims=False): return self._r("reduce_prod", dims, keepDims)
    def norm1(self, *dims, keepDims=False): return self._r("reduce_norm1", dims, keepDims)
    def norm2(self, *dims, keepDims=False): return self._r("reduce_norm2", dims, keepDims)
    def std(self, biasCorrected=True, *dims): return self.sd._op("reduce_stdev", [self], iArgs=list(dims), bArgs=[False, biasCorrected])[0]
    def argmax(self, *dims): return self.sd._op("argmax", [self], iArgs=list(dims))[0]
    def argmin(self, *dims): return self.sd._op("argmin", [self], iArgs=list(dims))[0]
    def reshape(self, *shape): return self.sd._op("reshape", [self], iArgs=list(shape[0]) if len(shape) == 1 and isinstance(shape[0], (list, tuple)) else list(shape))[0]
    def permute(self, *dims): return self.sd._op("permute", [self], iArgs=list(dims))[0]
    def castTo(self, dt): return self.sd._op("cast", [self], dArgs=[str(dt).upper()])[0]

    def get(self, *idx):
        """Strided-slice view with python ints / slices."""
        begin, end, stride, shrink, bm, em = [], [], [], 0, 0, 0
        for k, ix in enumerate(idx):
            if isinstance(ix, int):
                begin.append(ix), end.append(ix + 1), stride.append(1)
                shrink |= 1 << k
            else:
                if ix.start is None:
                    bm |= 1 << k
                if ix.stop is None:
                    em |= 1 << k
                begin.append(ix.start or 0), end.append(ix.stop or 0), stride.append(ix.step or 1)
        return self.sd._op("strided_slice", [self], iArgs=[bm, 0, em, 0, shrink] + begin + end + stride)[0]

    __getitem__ = lambda self, k: self.get(*(k if isinstance(k, tuple) else (k,)))

    def __repr__(self):
        return "SDVariable(name=%r, type=%s, dtype=%s, shape=%s)" % (self.name, self.variableType, self.dataType(), self.getShape())


class SameDiffOp:
    def __init__(self, name, opName, inputs, outputs, iArgs=None, tArgs=None, bArgs=None, dArgs=None, opType=OpType.CUSTOM, opNum=None,
                 dimensions=None, scalar=None, sArgs=None):
        self.name, self.opName = name, opName
        self.inputs, self.outputs = list(inputs), list(outputs)
        self.iArgs, self.tArgs, self.bArgs, self.dArgs = list(iArgs or []), list(tArgs or []), list(bArgs or []), list(dArgs or [])
        self.sArgs = list(sArgs or [])
        self.opType = opType
        self.opNum = opNum if opNum is not None else (op_hash(opName) if opType == OpType.CUSTOM else 0)
        self.dimensions = list(dimensions or [])
        self.scalar = scalar
        self.controlDeps: List[str] = []
        self.frameName: Optional[str] = None       # for enter
        self.isConstantEnter = False

    def __repr__(self):
        return "SameDiffOp(%s: %s(%s) -> %s)" % (self.name, self.opName, ", ".join(self.inputs), ", ".join(self.outputs))


class TrainingConfig:
    def __init__(self, updater=None, regularization=None, dataSetFeatureMapping=None, dataSetLabelMapping=None, minimize=True,
                 lossVariables=None, l1=None, l2=None, weightDecay=None, dataSetFeatureMaskMapping=None, dataSetLabelMaskMapping=None,
                 initialLossDataType="FLOAT"):
        self.updater = updater or LR.Adam(1e-3)
        self.regularization = list(regularization or [])
        if l1:
            self.regularization.append(LR.L1Regularization(l1))
        if l2:
            self.regularization.append(LR.L2Regularization(l2))
        if weightDecay:
            self.regularization.append(LR.WeightDecay(weightDecay, True))
        self.dataSetFeatureMapping = list(dataSetFeatureMapping or [])
        self.dataSetLabelMapping = list(dataSetLabelMapping or [])
        self.dataSetFeatureMaskMapping = list(dataSetFeatureMaskMapping or [])
        self.dataSetLabelMaskMapping = list(dataSetLabelMaskMapping or [])
        self.minimize = minimize
        self.lossVariables = list(lossVariables or [])
        self.iterationCount = 0
        self.epochCount = 0

    class Builder:
        def __init__(self):
            self._kw = {}

        def updater(self, u): self._kw["updater"] = u; return self
        def l1(self, v): self._kw["l1"] = v; return self
        def l2(self, v): self._kw["l2"] = v; return self
        def weightDecay(self, v, applyLR=True): self._kw["weightDecay"] = v; return self
        def regularization(self, *r): self._kw["regularization"] = list(r); return self
        def dataSetFeatureMapping(self, *n): self._kw["dataSetFeatureMapping"] = list(n); return self
        def dataSetLabelMapping(self, *n): self._kw["dataSetLabelMapping"] = list(n); return self
        def dataSetFeatureMaskMapping(self, *n): self._kw["dataSetFeatureMaskMapping"] = list(n); return self
        def dataSetLabelMaskMapping(self, *n): self._kw["dataSetLabelMaskMapping"] = list(n); return self
        def minimize(self, *names):
            if names and isinstance(names[0], bool):
                self._kw["minimize"] = names[0]
            else:
                self._kw["lossVariables"] = list(names)
            return self
        def build(self): return TrainingConfig(**self._kw)

    @staticmethod
    def builder():
        return TrainingConfig.Builder()

    def toJson(self):
        return json.dumps({"updater": self.updater.to_json(), "regularization": [r.to_json() for r in self.regularization],
                           "dataSetFeatureMapping": self.dataSetFeatureMapping, "dataSetLabelMapping": self.dataSetLabelMapping,
                           "dataSetFeatureMaskMapping": self.dataSetFeatureMaskMapping,
                           "dataSetLabelMaskMapping": self.dataSetLabelMaskMapping, "minimize": self.minimize,
                           "lossVariables": self.lossVariables, "iterationCount": self.iterationCount, "epochCount": self.epochCount})

    @staticmethod
    def fromJson(s):
        d = json.loads(s)
        tc = TrainingConfig(LR.updater_from_json(d.get("updater")), LR.regularization_from_json(d.get("regularization")),
                            d.get("dataSetFeatureMapping"), d.get("dataSetLabelMapping"), d.get("minimize", True), d.get("lossVariables"),
                            dataSetFeatureMaskMapping=d.get("dataSetFeatureMaskMapping"), dataSetLabelMaskMapping=d.get("dataSetLabelMaskMapping"))
        tc.iterationCount, tc.epochCount = d.get("iterationCount", 0), d.get("epochCount", 0)
        return tc


class LossCurve:
    def __init__(self, names):
        self.lossNames = list(names)
        self.values: List[List[float]] = []

    def addLossAndCopy(self, vals):
        self.values.append(list(vals))
        return self

    def meanLoss(self, epoch):
        return self.values[epoch]

    def lastMeanLoss(self):
        return self.values[-1]

    def lastMeanDelta(self):
        return [a - b for a, b in zip(self.values[-1], self.values[-2])] if len(self.values) > 1 else None


class History:
    def __init__(self):
        self.lossCurve: Optional[LossCurve] = None
        self.trainingTimeMillis = 0
        self.evaluations = {}
        self.epochs = 0

    def getLossCurve(self):
        return self.lossCurve

    def trainingEval(self, name):
        return self.evaluations.get(name)


# ----------------------------------------------------------------------------- SameDiff
class SameDiff:
    def __init__(self, device=None):
        self.variables: "OrderedDict[str, SDVariable]" = OrderedDict()
        self.ops: "OrderedDict[str, SameDiffOp]" = OrderedDict()
        self._arrays: Dict[str, torch.Tensor] = {}
        self._producer: Dict[str, str] = {}
        self._consumers: Dict[str, List[str]] = defaultdict(list)
        self.lossVariables: List[str] = []
        self.trainingConfig: Optional[TrainingConfig] = None
        self.listeners: list = []
        self.updaterState: Dict[str, torch.Tensor] = {}
        self._updaters: Dict[str, LR.GradientUpdater] = {}
        self._scope: List[str] = []
        self._last_grads: Dict[str, torch.Tensor] = {}
        self.device = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        from . import namespaces as NS
        self.math, self.nn, self.cnn, self.rnn = NS.SDMath(self), NS.SDNN(self), NS.SDCNN(self), NS.SDRNN(self)
        self.loss, self.linalg, self.bitwise, self.random, self.image = NS.SDLoss(self), NS.SDLinalg(self), NS.SDBitwise(self), NS.SDRandom(self), NS.SDImage(self)
        self.base = NS.SDBaseOps(self)

    @staticmethod
    def create(device=None):
        return SameDiff(device)

    # -- naming
    def withNameScope(self, scope):
        sd = self

        class _Scope:
            def __enter__(self_):
                sd._scope.append(scope)
                return self_

            def __exit__(self_, *exc):
                sd._scope.pop()
                return False
        return _Scope()

    def _unique(self, base):
        if self._scope:
            base = "/".join(self._scope) + "/" + base
        if base not in self.variables and base not in self.ops:
            return base
        k = 1
        while "%s_%d" % (base, k) in self.variables or "%s_%d" % (base, k) in self.ops:
            k += 1
        return "%s_%d" % (base, k)

    # -- variable creation
    def _add_var(self, name, vtype, dtype=None, shape=None, arr=None):
        v = SDVariable(self, name, vtype, dtype, shape)
        self.variables[name] = v
        if arr is not None:
            self._arrays[name] = arr
        return v

    def var(self, name=None, *args, dtype=None, weightInit=None):
        """var(name, array) | var(name, shape...) | var(name, weightInitFn, shape...)"""
        name = self._unique(name or "variable")
        if args and isinstance(args[0], (torch.Tensor, INDArray, np.ndarray)) or (args and isinstance(args[0], (list, tuple)) and args[0]
                                                                                 and isinstance(args[0][0], (list, float))):
            arr = _tt(args[0], self.device, DataType.to_torch(dtype) if dtype else None).clone()
            return self._add_var(name, VariableType.VARIABLE, arr.dtype, arr.shape, arr)
        if args and callable(args[0]):
            weightInit, args = args[0], args[1:]
        shape = [int(s) for s in (args[0] if len(args) == 1 and isinstance(args[0], (list, tuple)) else args)]
        td = DataType.to_torch(dtype) if dtype else torch.float32
        if weightInit is not None:
            arr = _tt(weightInit(shape), self.device, td)
        else:
            fan_in = shape[0] if shape else 1
            fan_out = shape[1] if len(shape) > 1 else 1
            g = torch.Generator().manual_seed(abs(hash(name)) % (2 ** 31))
            arr = (torch.randn(shape, generator=g) * math.sqrt(2.0 / (fan_in + fan_out))).to(self.device, td)      # XAVIER default
        return self._add_var(name, VariableType.VARIABLE, td, shape, arr)

    def zero(self, name, *shape, dtype=None):
        return self.var(name, torch.zeros([int(s) for s in shape], dtype=DataType.to_torch(dtype) if dtype else torch.float32))

    def one(self, name, *shape, dtype=None):
        return self.var(name, torch.ones([int(s) for s in shape], dtype=DataType.to_torch(dtype) if dtype else torch.float32))

    def constant(self, name, value=None, dtype=None):
        if value is None and not isinstance(name, str):
            name, value = None, name
        arr = _tt(value, self.device, DataType.to_torch(dtype) if dtype else None)
        name = self._unique(name or "const")
        return self._add_var(name, VariableType.CONSTANT, arr.dtype, arr.shape, arr)

    def scalar(self, name, value, dtype=None):
        return self.constant(name, torch.tensor(value), dtype)

    def placeHolder(self, name, dtype="FLOAT", *shape):
        shape = list(shape[0]) if len(shape) == 1 and isinstance(shape[0], (list, tuple)) else list(shape)
        return self._add_var(self._unique(name), VariableType.PLACEHOLDER, DataType.to_torch(dtype), shape)

    # -- graph construction
    def _op(self, opName, inputs, iArgs=None, tArgs=None, bArgs=None, dArgs=None, nOut=1, name=None, opType=OpType.CUSTOM, opNum=None,
            dimensions=None, scalar=None, outNames=None):
        ins = [v if isinstance(v, SDVariable) else self.constant(None, v) for v in inputs]
        icpt = getattr(self, "_arg_interceptor", None)
        if icpt is not None:
            ins = [icpt(v) for v in ins]
        oname = self._unique(name or opName)
        outs = outNames or ([oname] if nOut == 1 else [oname if k == 0 else "%s:%d" % (oname, k) for k in range(nOut)])
        o = SameDiffOp(oname, opName, [v.name for v in ins], outs, iArgs, tArgs, bArgs, dArgs, opType, opNum, dimensions, scalar)
        self._add_op(o)
        return [self.variables[n] for n in outs]

    def _add_op(self, o: SameDiffOp):
        self.ops[o.name] = o
        for n in o.outputs:
            if n not in self.variables:
                self._add_var(n, VariableType.ARRAY)
            self._producer[n] = o.name
        for n in o.inputs:
            self._consumers[n].append(o.name)
        return o

    def invokeGraphOn(self, other: "SameDiff"):
        for n, v in self.variables.items():
            if n not in other.variables:
                other._add_var(n, v.variableType, v.dtype, v.shape, self._arrays.get(n))
        for o in self.ops.values():
            if o.name not in other.ops:
                other._add_op(o)
        return other

    # -- lookups / edits
    def getVariable(self, name):
        return self.variables[name]

    def hasVariable(self, name):
        return name in self.variables

    def getArrForVarName(self, name):
        a = self._arrays.get(name)
        return None if a is None else INDArray(a)

    def setArrayForVariable(self, name, arr):
        self._arrays[name] = _tt(arr, self.device)
        v = self.variables[name]
        v.dtype, v.shape = self._arrays[name].dtype, list(self._arrays[name].shape)

    associateArrayWithVariable = lambda self, arr, var: self.setArrayForVariable(var.name if isinstance(var, SDVariable) else var, arr)

    def inputs(self):
        return [n for n, v in self.variables.items() if v.variableType == VariableType.PLACEHOLDER]

    def outputs(self):
        """Variables that no op consumes (ARRAY type) — the reference's default output inference."""
        return [n for n, v in self.variables.items() if v.variableType == VariableType.ARRAY and not self._consumers.get(n)]

    def variableNames(self):
        return list(self.variables)

    def trainableVariables(self):
        return [n for n, v in self.variables.items() if v.variableType == VariableType.VARIABLE and
                (v.dtype is None or v.dtype.is_floating_point)]

    def renameVariable(self, old, new):
        v = self.variables.pop(old)
        v.name = new
        self.variables[new] = v
        if old in self._arrays:
            self._arrays[new] = self._arrays.pop(old)
        if old in self._producer:
            p = self._producer.pop(old)
            self._producer[new] = p
            self.ops[p].outputs = [new if n == old else n for n in self.ops[p].outputs]
        for c in self._consumers.pop(old, []):
            self.ops[c].inputs = [new if n == old else n for n in self.ops[c].inputs]
            self._consumers[new].append(c)
        self.lossVariables = [new if n == old else n for n in self.lossVariables]

    def convertToConstants(self, names):
        for n in names:
            self.variables[n].variableType = VariableType.CONSTANT

    def convertToVariables(self, names):
        for n in names:
            self.variables[n].variableType = VariableType.VARIABLE

    def addLossVariable(self, name):
        name = name.name if isinstance(name, SDVariable) else name
        if name not in self.lossVariables:
            self.lossVariables.append(name)

    def setLossVariables(self, *names):
        self.lossVariables = [n.name if isinstance(n, SDVariable) else n for n in names]

    def getLossVariables(self):
        return list(self.lossVariables)

    def setTrainingConfig(self, tc: TrainingConfig):
        self.trainingConfig = tc
        if tc.lossVariables and not self.lossVariables:
            self.lossVariables = list(tc.lossVariables)

    def getTrainingConfig(self):
        return self.trainingConfig

    def setListeners(self, *ls):
        self.listeners = list(ls)

    def addListeners(self, *ls):
        self.listeners += list(ls)

    def numElements(self):
        return sum(a.numel() for n, a in self._arrays.items() if self.variables[n].variableType == VariableType.VARIABLE)

    def summary(self):
        rows = ["--- Summary ---", "Variables: %d  Ops: %d  Trainable params: %d" % (len(self.variables), len(self.ops), self.numElements()),
                "%-40s %-12s %-10s %s" % ("name", "type", "dtype", "shape")]
        for n, v in self.variables.items():
            rows.append("%-40s %-12s %-10s %s" % (n[:40], v.variableType, v.dataType(), v.getShape()))
        rows.append("%-40s %-28s %s" % ("op", "opName", "inputs -> outputs"))
        for o in self.ops.values():
            rows.append("%-40s %-28s %s -> %s" % (o.name[:40], o.opName, o.inputs, o.outputs))
        return "\n".join(rows)

    # ------------------------------------------------------------------ execution
    def _exec_op(self, o: SameDiffOp, ins: List[torch.Tensor], allow_helpers=True, mmgr=None):
        """Run one op. ``mmgr`` (an inference session's SessionMemMgr) supplies the output buffer of same-shape legacy
        transforms / pairwise ops, so recycled arrays are written in place of fresh allocations."""
        ex = OpExecutioner.getInstance()
        t = o.opType
        if o.opName == "__vjp__":
            from .gradfn import exec_vjp
            return exec_vjp(self, o, ins)
        if t in (OpType.CUSTOM, OpType.UDF, OpType.BOOLEAN, OpType.SHAPE):
            res = ex._run(o.opName, ins, o.iArgs, o.tArgs, o.bArgs, o.dArgs, allow_helpers)
            return list(res) if isinstance(res, (tuple, list)) else [res]
        lg = LG.executioner()
        dims = o.dimensions or None
        if t in (OpType.REDUCE_FLOAT, OpType.REDUCE_SAME, OpType.REDUCE_LONG, OpType.REDUCE_BOOL, OpType.INDEX_REDUCE, OpType.SUMMARYSTATS,
                 OpType.REDUCE_3):
            # an explicit axis input (TF-imported graphs) wins over the stored dimensions field
            k = 2 if t == OpType.REDUCE_3 else 1
            if len(ins) > k and ins[k] is not None and not ins[k].is_floating_point():
                dims = [int(v) for v in ins[k].reshape(-1).tolist()]
        keep = bool(o.bArgs[0]) if o.bArgs else False
        ext = o.tArgs or None
        n = o.opNum
        x = ins[0] if ins else None
        z = None
        if mmgr is not None and x is not None and x.is_floating_point() and not torch.is_grad_enabled() and (
                t in (OpType.TRANSFORM_SAME, OpType.TRANSFORM_STRICT) or
                (t == OpType.PAIRWISE and ins[1].shape == x.shape and ins[1].dtype == x.dtype)):
            z = mmgr.allocate(False, x.dtype, *x.shape, device=x.device)
        if t == OpType.TRANSFORM_SAME: return [lg.execTransformSame(n, x, z, ext)]
        if t == OpType.TRANSFORM_STRICT: return [lg.execTransformStrict(n, x, z, ext)]
        if t == OpType.TRANSFORM_FLOAT: return [lg.execTransformFloat(n, x, None, ext)]
        if t == OpType.TRANSFORM_BOOL: return [lg.execTransformBool(n, x, None, ext)]
        if t == OpType.TRANSFORM_ANY: return [lg.execTransformAny(n, x, None, ext)]
        if t == OpType.PAIRWISE: return [lg.execPairwiseTransform(n, x, ins[1], z, ext)]
        if t == OpType.PAIRWISE_BOOL: return [lg.execPairwiseTransformBool(n, x, ins[1], None, ext)]
        if t == OpType.BROADCAST: return [lg.execBroadcast(n, x, ins[1], None, dims)]
        if t == OpType.BROADCAST_BOOL: return [lg.execBroadcastBool(n, x, ins[1], None, dims)]
        if t in (OpType.SCALAR, OpType.SCALAR_BOOL):
            sc = o.scalar if o.scalar is not None else (ins[1] if len(ins) > 1 else (o.tArgs[0] if o.tArgs else 0.0))
            sc = float(sc.reshape(-1)[0]) if isinstance(sc, torch.Tensor) else float(sc)
            return [lg.execScalar(n, x, sc) if t == OpType.SCALAR else lg.execScalarBool(n, x, sc)]
        if t == OpType.REDUCE_FLOAT: return [lg.execReduceFloat(n, x, dims, keep, ext)]
        if t == OpType.REDUCE_SAME: return [lg.execReduceSame(n, x, dims, keep, ext)]
        if t == OpType.REDUCE_BOOL: return [lg.execReduceBool(n, x, dims, keep, ext)]
        if t == OpType.REDUCE_LONG: return [lg.execReduceLong(n, x, dims, keep, ext)]
        if t == OpType.REDUCE_3: return [lg.execReduce3(n, x, ins[1], dims, keep, ext)]
        if t == OpType.INDEX_REDUCE: return [lg.execIndexReduce(n, x, dims, keep, ext)]
        if t == OpType.SUMMARYSTATS: return [lg.execSummaryStats(n, x, dims, bool(o.bArgs[0]) if o.bArgs else True)]
        if t == OpType.RANDOM:
            name = LG.RANDOM[n]
            if name == "Linspace":
                if len(ins) >= 3:
                    return [torch.linspace(float(ins[0]), float(ins[1]), int(ins[2]), device=ins[0].device)]
                return [torch.linspace(o.tArgs[0], o.tArgs[1], int(o.iArgs[0]) if o.iArgs else int(o.tArgs[2]), device=self.device)]
            from ..nd4j import rng as R
            shape = [int(v) for v in ins[0].tolist()] if ins and not ins[0].is_floating_point() else list(o.iArgs)
            z = torch.empty(shape, device=self.device)
            return [R.exec_random(name, None, z, None, None, o.tArgs)]
        raise NotImplementedError("op type %s (%s)" % (t, o.opName))

    # ------------------------------------------------------------------ static shape inference
    def _probe(self, name, probe, cache):
        """Value of ``name`` with unknown placeholder dims set to ``probe``: real tensors for variables/constants,
        ``meta`` tensors downstream of placeholders (no compute, no memory)."""
        if name in cache:
            return cache[name]
        v = self.variables[name]
        t = None
        if v.variableType in (VariableType.VARIABLE, VariableType.CONSTANT):
            t = self._arrays.get(name)
        elif v.variableType == VariableType.PLACEHOLDER:
            if v.shape is not None:
                t = torch.empty([probe if (s is None or s < 0) else int(s) for s in v.shape], dtype=v.dtype or torch.float32, device="meta")
        else:
            pn = self._producer.get(name)
            if pn is not None:
                o = self.ops[pn]
                ins = [self._probe(i, probe, cache) for i in o.inputs]
                outs = None
                if all(x is not None for x in ins) and o.opName.lower() not in _CONTROL:
                    if any(x.device.type == "meta" for x in ins):
                        ins = [x if (x.device.type == "meta" or not x.is_floating_point()) else
                               torch.empty(x.shape, dtype=x.dtype, device="meta") for x in ins]
                    else:
                        ins = [x.detach() for x in ins]
                    try:
                        with torch.no_grad():
                            outs = self._exec_op(o, ins, allow_helpers=False)
                    except Exception:
                        outs = None
                for k, on in enumerate(o.outputs):
                    cache[on] = outs[k] if outs is not None and k < len(outs) and isinstance(outs[k], torch.Tensor) else None
                return cache.get(name)
        cache[name] = t
        return t

    def inferShape(self, name):
        """Static shape of any variable (``-1`` for dims that depend on unknown placeholder dims), or None."""
        import sys
        sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
        key = (len(self.ops), len(self.variables))
        if getattr(self, "_shape_key", None) != key:
            self._shape_key, self._shape_caches = key, ({}, {})
        a = self._probe(name, 2, self._shape_caches[0])
        b = self._probe(name, 3, self._shape_caches[1])
        if a is None or b is None or a.dim() != b.dim():
            return None
        return [int(x) if x == y else -1 for x, y in zip(a.shape, b.shape)]

    def _required_ops(self, outputs: Sequence[str]) -> set:
        need, stack = set(), list(outputs)
        seen = set()
        while stack:
            v = stack.pop()
            if v in seen:
                continue
            seen.add(v)
            p = self._producer.get(v)
            if p is None or p in need:
                continue
            need.add(p)
            o = self.ops[p]
            stack += o.inputs + o.controlDeps
        return need

    def _run_graph(self, placeholders: Dict[str, torch.Tensor], outputs: Sequence[str], training=False, listeners=(), at=None, lifetime=None):
        """Dataflow interpreter with frames. Values are keyed by (variable, frame, iteration).

        ``lifetime`` (sessions.ArrayLifetime, set by InferenceSession) tracks, per produced array, the ops that still have to read it
        (a DependencyTracker: array -> consumer ops) and hands the array to the session's memory manager as soon as the last one has
        run — the reference's InferenceSession.arrayUseTracker (internal/InferenceSession.java:96-140). Only graphs without frames."""
        need = self._required_ops(outputs)
        ROOT = ("", 0)
        vals: Dict[tuple, torch.Tensor] = {}
        frame_parent: Dict[str, tuple] = {}
        for n, v in self.variables.items():
            if v.variableType in (VariableType.VARIABLE, VariableType.CONSTANT):
                if n in self._arrays:
                    vals[(n,) + ROOT] = self._arrays[n]
            elif v.variableType == VariableType.PLACEHOLDER and n in placeholders:
                vals[(n,) + ROOT] = placeholders[n]
        missing = [n for n in self.inputs() if n not in placeholders and any(n in self.ops[o].inputs for o in need)]
        if missing:
            raise ValueError("no value provided for placeholder(s) %s" % missing)
        has_control = any(self.ops[o].opName.lower() in _CONTROL for o in need)
        if lifetime is not None and (has_control or torch.is_grad_enabled()):
            lifetime = None
        if lifetime is not None:
            lifetime.begin(set(outputs))
            for v in vals.values():
                lifetime.pin(v)                   # parameters / constants / fed arrays are not the session's to recycle

        def lookup(var, frame, it):
            f, i = frame, it
            while True:
                k = (var, f, i)
                if k in vals:
                    return vals[k]
                if f == "":
                    return None
                # loop-invariant values entered with is_constant are visible in every iteration
                k0 = (var, f, -1)
                if k0 in vals:
                    return vals[k0]
                f, i = frame_parent[f]

        executed = set()
        results: Dict[str, torch.Tensor] = {}

        def run(o: SameDiffOp, frame, it):
            key = (o.name, frame, it)
            if key in executed:
                return []
            kind = o.opName.lower() if o.opName.lower() in _CONTROL else None
            ins = [lookup(n, frame, it) for n in o.inputs]
            if kind == "merge":
                avail = [x for x in ins if x is not None]
                if not avail:
                    return []
                executed.add(key)
                # prefer the value produced in this very iteration (NextIteration) over the loop-entry value
                exact = [vals.get((n, frame, it)) for n in o.inputs]
                exact = [x for x in exact if x is not None]
                outs = [(exact or avail)[-1 if it > 0 else 0]]
                oframe, oit = frame, it
            else:
                if any(x is None for x in ins):
                    return []
                if any(lookup(c, frame, it) is None for c in o.controlDeps if c in self.variables):
                    return []
                executed.add(key)
                oframe, oit = frame, it
                if kind == "enter":
                    fname = o.frameName or (o.sArgs[0] if o.sArgs else o.name)
                    child = fname if frame == "" else "%s@%s#%d" % (fname, frame, it)     # one frame instance per parent iteration
                    if child not in frame_parent:
                        frame_parent[child] = (frame, it)
                    # loop variables go Enter -> Merge; an Enter that feeds no Merge is loop-invariant
                    # (TF is_constant=true) and must be visible in every iteration of the frame
                    invariant = o.isConstantEnter or not any(self.ops[c].opName.lower() == "merge"
                                                             for c in self._consumers.get(o.outputs[0], ()))
                    outs, oframe, oit = [ins[0]], child, (-1 if invariant else 0)
                elif kind == "exit":
                    pf, pi = frame_parent.get(frame, ROOT)
                    outs, oframe, oit = [ins[0]], pf, pi
                elif kind == "next_iteration":
                    outs, oit = [ins[0]], it + 1
                    if it > 100000:
                        raise RuntimeError("loop did not terminate")
                elif kind == "loop_cond":
                    outs = [ins[0]]
                elif kind == "switch":
                    pred = bool(ins[1].reshape(-1)[0] != 0) if ins[1].numel() else False
                    outs = [None, ins[0]] if pred else [ins[0], None]
                else:
                    t0 = time.perf_counter_ns()
                    for l in listeners:
                        l.preOpExecution(self, at, o, ins)
                    outs = self._exec_op(o, ins, allow_helpers=True, mmgr=lifetime.mmgr if lifetime is not None else None)
                    for l in listeners:
                        l.opExecution(self, at, o, ins, outs, time.perf_counter_ns() - t0)
            produced = []
            for n, val in zip(o.outputs, outs):
                if val is None:
                    continue
                vals[(n, oframe, oit)] = val
                if oframe == "":
                    results[n] = val
                produced.append((n, oframe, oit))
            if lifetime is not None:
                for n, val in zip(o.outputs, outs):
                    if val is not None:
                        lifetime.produced(n, val, [c for c in self._consumers.get(n, ()) if c in need])
                for dead in lifetime.executed(o.name):
                    vals.pop((dead, "", 0), None)
                    results.pop(dead, None)
            return produced

        # exact-context propagation with a worklist ...
        queue = deque(vals.keys())

        def live_contexts():
            return {(f, i) for (_, f, i) in vals if i >= 0} | {ROOT}

        def drain():
            while queue:
                n, f, i = queue.popleft()
                for cn in self._consumers.get(n, ()):
                    if cn not in need:
                        continue
                    ctxs = [(f, i)] if i >= 0 else [c for c in live_contexts() if c[0] == f]
                    for c in ctxs:
                        queue.extend(run(self.ops[cn], *c))

        for on in need:                          # ops without inputs (random / create with only iArgs)
            if not self.ops[on].inputs:
                queue.extend(run(self.ops[on], "", 0))
        drain()
        # ... plus a fixpoint over live (frame, iteration) contexts for values that cross frame boundaries
        # (ancestor-frame values consumed inside loops, loop-invariant Enter values)
        if has_control:
            for _ in range(10_000_000):
                progressed = False
                for (f, i) in sorted(live_contexts()):
                    for on in need:
                        o = self.ops[on]
                        if (o.name, f, i) in executed:
                            continue
                        if (f, i) != ROOT and not any((n, f, i) in vals or (n, f, -1) in vals for n in o.inputs):
                            continue
                        p = run(o, f, i)
                        if p:
                            queue.extend(p)
                            drain()
                            progressed = True
                if not progressed:
                    break
        out = {}
        for n in outputs:
            v = results.get(n)
            if v is None:
                v = vals.get((n, "", 0))
            if v is None:
                raise RuntimeError("output '%s' was not produced (dead branch or missing input)" % n)
            out[n] = v
        return out

    def _prep_placeholders(self, placeholders):
        out = {}
        for k, v in (placeholders or {}).items():
            k = k.name if isinstance(k, SDVariable) else k
            t = _tt(v, self.device)
            pv = self.variables.get(k)
            if pv is not None and pv.dtype is not None and t.dtype != pv.dtype and pv.variableType == VariableType.PLACEHOLDER:
                t = t.to(pv.dtype)
            out[k] = t
        return out

    def output(self, placeholders=None, outputs=None, *more) -> Dict[str, torch.Tensor]:
        if isinstance(outputs, str):
            outputs = [outputs] + list(more)
        outputs = [o.name if isinstance(o, SDVariable) else o for o in (outputs or self.outputs())]
        from .listeners import At, Operation
        at = At(self.trainingConfig.epochCount if self.trainingConfig else 0, self.trainingConfig.iterationCount if self.trainingConfig else 0,
                Operation.INFERENCE)
        with torch.no_grad():
            return self._run_graph(self._prep_placeholders(placeholders), outputs, False, self.listeners, at)

    def outputSingle(self, placeholders, output):
        output = output.name if isinstance(output, SDVariable) else output
        return INDArray(self.output(placeholders, [output])[output])

    def outputAll(self, placeholders=None):
        return self.output(placeholders, [n for n, v in self.variables.items() if v.variableType == VariableType.ARRAY])

    exec = output
    batchOutput = output

    # ------------------------------------------------------------------ gradients / training
    def _bf16_shadowable(self):
        """Trainable fp32 variables whose ONLY consumers are ``cast`` ops to BFLOAT16 (mixed-precision graphs: fp32 master
        variable, bf16 at every use). During a training step their leaves can be views of one bf16 copy of the flat variable
        vector: one cast launch instead of one per variable, and the per-variable cast ops become no-ops."""
        key = (len(self.ops), len(self.variables))
        c = getattr(self, "_shadow_cache", None)
        if c is not None and c[0] == key:
            return c[1]
        users: Dict[str, list] = {}
        for o in self.ops.values():
            for i_ in o.inputs:
                users.setdefault(i_, []).append(o)
        ok = set()
        for n in self.trainableVariables():
            us = users.get(n, [])
            if us and all(u.opName == "cast" and [str(a).upper() for a in (u.dArgs or [])] in (["BFLOAT16"], ["BF16"]) for u in us):
                ok.add(n)
        self._shadow_cache = (key, ok)
        return ok

    def _mixed_precision_state(self, fu):
        """Persistent compute-precision image of the flat fp32 variable vector for mixed-precision graphs (fp32 master variables
        whose only consumers are casts to BFLOAT16): [bf16 copy of the whole vector | transposed replicas of the 2-D weights that
        feed xw_plus_b / fused_dense / matmul], refreshed by two launches per step, plus the ops/linear.py Links (leaf data pointer ->
        fp32 master view, fp32 gradient view, transposed replica) through which the kernels skip the per-variable glue."""
        shadowable = self._bf16_shadowable()
        if not shadowable:
            return None
        mp = fu.get("mp")
        if mp is not None and mp["key"] == (len(self.ops), len(self.variables)):
            return mp
        from ..ops import linear as LIN
        flat = fu["flat"]
        if fu.get("flat_g") is None:
            fu["flat_g"] = torch.zeros_like(flat)
            o_ = 0
            fu["g_views"] = []
            for sz in fu["sizes"]:
                fu["g_views"].append(fu["flat_g"][o_:o_ + sz])
                o_ += sz
        offs, off = {}, 0
        for n_, sz in zip(fu["names"], fu["sizes"]):
            offs[n_] = (off, sz)
            off += sz
        total = off
        # which shadowable 2-D variables are the weight operand of a linear-type op (through their cast)?
        as_weight = set()
        for o in self.ops.values():
            if o.opName in ("xw_plus_b", "fused_dense", "matmul") and len(o.inputs) >= 2:
                p = self._producer.get(o.inputs[1])
                src = self.ops[p].inputs[0] if (p is not None and self.ops[p].opName == "cast") else o.inputs[1]
                if src in shadowable:
                    as_weight.add(src)
        t_entries, t_off, cur = [], {}, (total + 7) // 8 * 8
        for n_ in fu["names"]:
            a = self._arrays[n_]
            if n_ in as_weight and n_ in offs and a.dim() == 2 and a.shape[0] % 8 == 0 and a.shape[1] % 8 == 0:
                t_entries.append((offs[n_][0], cur, int(a.shape[0]), int(a.shape[1])))
                t_off[n_] = cur
                cur += (a.numel() + 7) // 8 * 8
        shadow = torch.empty(max(cur, 8), dtype=torch.bfloat16, device=flat.device)
        views, gviews, links, by_name = {}, {}, {}, {}
        for k, n_ in enumerate(fu["names"]):
            o0, sz = offs[n_]
            if n_ not in shadowable:             # fp32 leaf = the flat view itself: linked for direct gradient accumulation only
                links[flat.data_ptr() + 4 * o0] = by_name[n_] = LIN.Link(flat[o0:o0 + sz], fu["g_views"][k], None, sz, torch.float32)
                continue
            a = self._arrays[n_]
            v = shadow[o0:o0 + sz]
            views[n_] = v
            gviews[n_] = fu["g_views"][k]
            wt = shadow[t_off[n_]:t_off[n_] + sz].view(a.shape[1], a.shape[0]) if n_ in t_off else None
            links[v.data_ptr()] = by_name[n_] = LIN.Link(flat[o0:o0 + sz], fu["g_views"][k], wt, sz, torch.bfloat16)
        mp = fu["mp"] = {"key": (len(self.ops), len(self.variables)), "shadow": shadow, "total": total, "offs": offs, "views": views,
                         "gviews": gviews, "links": links, "link_by_name": by_name, "ttable": LIN.TransposeTable(t_entries, flat.device)}
        return mp

    def _loss_and_grads(self, placeholders, wrt: Sequence[str], extra_outputs=(), listeners=(), at=None, device_losses=False,
                        shadow_ok=False):
        if not self.lossVariables:
            raise RuntimeError("no loss variables set: call setLossVariables / markAsLoss")
        gf = self._gradient_graph_for(wrt, extra_outputs, shadow_ok)
        if gf is not None:
            losses, gd = gf.execute(placeholders, list(wrt), listeners, at, training=True)
            if device_losses:
                return [l.detach().sum() for l in losses], gd, {}
            return [float(l.detach().sum()) for l in losses], gd, {}
        ph = self._prep_placeholders(placeholders)
        saved = {}
        leaves = []
        fu = getattr(self, "_fused", None) if shadow_ok else None
        mp = self._mixed_precision_state(fu) if (fu is not None and self.device.type == "cuda") else None
        if mp is not None:
            # ONE cast launch + ONE transpose-cast launch refresh every compute-precision replica; gradients of linked variables are
            # accumulated by the kernels straight into the (zeroed) flat gradient vector
            mp["shadow"][:mp["total"]].copy_(fu["flat"])
            mp["ttable"].run(fu["flat"], mp["shadow"])
            fu["flat_g"].zero_()
            from ..ops import linear as LIN
            LIN.set_links(mp["links"])
        for n in wrt:
            if n in ph:
                t = ph[n].detach().clone().requires_grad_(True)
                ph[n] = t
            else:
                saved[n] = self._arrays[n]
                a = self._arrays[n]
                if mp is not None and n in mp["views"] and a.data_ptr() == fu["flat"].data_ptr() + 4 * mp["offs"][n][0] \
                        and a.numel() == mp["offs"][n][1]:
                    t = mp["views"][n].view(a.shape).detach().requires_grad_(True)
                else:
                    t = a.detach().requires_grad_(True)
                self._arrays[n] = t
            leaves.append(t)
        try:
            with torch.enable_grad(), LG.force_torch():
                res = self._run_graph(ph, list(self.lossVariables) + list(extra_outputs), True, listeners, at)
                losses = [res[n] for n in self.lossVariables]
                total = sum(l.sum() for l in losses)
                from ..ops import streams as _streams
                with _streams.dense_scope():      # dense weight gradients overlap the dx chain on a side stream; joined here
                    grads = torch.autograd.grad(total, leaves, allow_unused=True)
        finally:
            for n, a in saved.items():
                self._arrays[n] = a
            if mp is not None:
                LIN.set_links(None)
        if mp is not None:
            # linked variables: the kernels already accumulated into the flat gradient views; add what autograd returned on top
            grads = list(grads)
            for k, (n, g) in enumerate(zip(wrt, grads)):
                lk = mp["link_by_name"].get(n)
                if lk is None or (lk.hits == 0 and g is not None):
                    continue                      # untouched by the kernels: the multi-tensor copy of _fused_update takes g
                if g is not None:
                    lk.grad.add_(g.reshape(-1))
                grads[k] = lk.grad.view(leaves[k].shape)
        gd = {n: (g if g is not None else torch.zeros_like(t)) for n, g, t in zip(wrt, grads, leaves)}
        if device_losses:          # no host synchronisation (HIP-graph capture of the training step)
            return [l.detach().sum() for l in losses], gd, {n: res[n].detach() for n in extra_outputs}
        return [float(l.detach().sum()) for l in losses], gd, {n: res[n].detach() for n in extra_outputs}

    def calculateGradients(self, placeholders, *variables) -> Dict[str, INDArray]:
        names = [v.name if isinstance(v, SDVariable) else v for v in (variables[0] if len(variables) == 1 and isinstance(variables[0], (list, tuple)) else variables)]
        _, g, _ = self._loss_and_grads(placeholders, names)
        self._last_grads.update(g)
        return {n: INDArray(t) for n, t in g.items()}

    def calculateGradientsAndOutputs(self, placeholders, outputs, variables):
        _, g, o = self._loss_and_grads(placeholders, list(variables), list(outputs))
        return {n: INDArray(t) for n, t in o.items()}, {n: INDArray(t) for n, t in g.items()}

    def createGradFunction(self, *variables):
        """Build (and cache) the gradient function as a graph of ``_bp`` ops (autodiff/gradfn.py; SameDiff.java:5013). The result's
        ``sd`` holds forward + backward ops and a ``<name>-grad`` variable per differentiated variable."""
        from .gradfn import create_grad_function
        names = [v.name if isinstance(v, SDVariable) else v for v in
                 (variables[0] if len(variables) == 1 and isinstance(variables[0], (list, tuple)) else variables)]
        key = (len(self.ops), len(self.variables), tuple(self.lossVariables), tuple(names))
        c = getattr(self, "_gradfn_cache", None)
        if c is None or c[0] != key:
            self._gradfn_cache = (key, create_grad_function(self, names or None))
        return self._gradfn_cache[1]

    def setGradientMode(self, mode):
        """How gradients are computed (SameDiff.java:5013 createGradFunction is the reference's only way):

        * ``'graph'``: execute the gradient graph of :meth:`createGradFunction` - every backward op is a node (``*_bp`` / ``*_grad``
          custom ops, derivative transforms; ops without a rule as single-op ``__vjp__`` nodes);
        * ``'autograd'``: a tape over the session interpreter;
        * ``'auto'`` (default, also ``DL4J_AMD_SD_GRAD_MODE``): the gradient graph whenever EVERY op on the loss path has an analytic
          rule (no ``__vjp__`` node: those re-run their forward op, which is wrong for random ops and wasteful otherwise), the graph
          has no control-flow frames and the call does not run on the mixed-precision replicas of the fused GPU training step
          (bf16 shadow weights with kernels accumulating straight into the flat gradient vector: that machinery hangs off autograd
          Functions); the tape otherwise."""
        assert mode in ("autograd", "graph", "auto")
        self._grad_mode = mode
        return self

    def gradientMode(self):
        return getattr(self, "_grad_mode", None) or os.environ.get("DL4J_AMD_SD_GRAD_MODE", "auto")

    def _gradient_graph_for(self, wrt, extra_outputs=(), shadow_ok=False):
        """The GradFunction to execute for this call, or None when the tape is the route (see setGradientMode)."""
        mode = self.gradientMode()
        if mode == "autograd" or extra_outputs or not wrt or not all(n in self._arrays for n in wrt):
            return None
        if mode == "auto":
            if shadow_ok and getattr(self, "_fused", None) is not None and self.device.type == "cuda":
                return None
            key = (len(self.ops), len(self.variables), tuple(self.lossVariables), tuple(wrt))
            c = getattr(self, "_auto_grad_cache", None)
            if c is not None and c[0] == key:
                return c[1]
            try:
                gf = self.createGradFunction(list(wrt))
                if gf.vjp_ops:
                    gf = None
            except (NotImplementedError, KeyError, RuntimeError):
                gf = None
            self._auto_grad_cache = (key, gf)
            return gf
        return self.createGradFunction(list(wrt))

    def grad(self, name):
        """Gradient variable handle ``<name>-grad`` (values filled by calculateGradients / fit)."""
        gname = name + "-grad"
        if gname not in self.variables:
            self._add_var(gname, VariableType.ARRAY)
        if name in self._last_grads:
            self._arrays[gname] = self._last_grads[name]
        return self.variables[gname]

    def _ensure_updaters(self, names):
        tc = self.trainingConfig
        for n in names:
            if n not in self._updaters:
                p = self._arrays[n]
                size = tc.updater.stateSize(p.numel())
                st = self.updaterState.get(n)
                fresh = st is None
                if fresh:
                    st = torch.zeros(size, dtype=torch.float32, device=p.device)
                    self.updaterState[n] = st
                self._updaters[n] = tc.updater.instantiate(st if size else None, fresh)

    def fit(self, data=None, numEpochs=1, validationData=None, validationFrequency=1, listeners=()):
        """fit(DataSet | MultiDataSet | iterator, numEpochs) -> History (reference: SameDiff.fit / TrainingSession.trainingIteration)."""
        from .listeners import At, Operation
        tc = self.trainingConfig
        if tc is None:
            raise RuntimeError("no TrainingConfig set")
        ls = list(self.listeners) + list(listeners)
        trainable = self.trainableVariables()
        self._ensure_updaters(trainable)
        hist = History()
        curve = LossCurve(self.lossVariables)
        t_start = time.time()

        def batches():
            if hasattr(data, "hasNext"):
                if data.resetSupported():
                    data.reset()
                while data.hasNext():
                    yield data.next()
            elif isinstance(data, (list, tuple)):
                yield from data
            else:
                yield data

        required = set(ll for l in ls for ll in (l.requiredVariables(self) if hasattr(l, "requiredVariables") else []))
        for ep in range(numEpochs):
            at = At(tc.epochCount, tc.iterationCount, Operation.TRAINING)
            for l in ls:
                l.epochStart(self, at)
            sums, nb = [0.0] * len(self.lossVariables), 0
            for ds in batches():
                at = At(tc.epochCount, tc.iterationCount, Operation.TRAINING)
                ph = self._placeholders_from_dataset(ds)
                for l in ls:
                    l.iterationStart(self, at, ds, 0)
                losses = self._captured_step(ph, trainable) if self._capture_eligible(ls, required) else None
                extra = {}
                if losses is None:
                    losses, grads, extra = self._loss_and_grads(ph, trainable, sorted(required), ls, at, shadow_ok=True)
                    self._apply_updates(grads, at)
                    self._last_grads.update(grads)
                for k, v in enumerate(losses):
                    sums[k] += v
                nb += 1
                for l in ls:
                    l.iterationDone(self, at, ds, losses)
                    for n, v in extra.items():
                        l.activationAvailable(self, at, ds, None, n, v)
                tc.iterationCount += 1
            mean = [s / max(nb, 1) for s in sums]
            curve.addLossAndCopy(mean)
            stop = False
            for l in ls:
                r = l.epochEnd(self, at, curve, int((time.time() - t_start) * 1000))
                stop = stop or (r == "STOP")
            if validationData is not None and (ep + 1) % validationFrequency == 0:
                for l in ls:
                    if hasattr(l, "validationDone"):
                        l.validationDone(self, at, validationData)
            tc.epochCount += 1
            hist.epochs += 1
            if stop:
                break
        hist.lossCurve = curve
        hist.trainingTimeMillis = int((time.time() - t_start) * 1000)
        for l in ls:
            l.operationEnd(self, Operation.TRAINING)
        return hist

    def _placeholders_from_dataset(self, ds):
        tc = self.trainingConfig
        if isinstance(ds, dict):
            return ds
        ph = {}
        feats = ds.getFeatures() if not isinstance(ds.getFeatures(), list) else ds.getFeatures()
        labs = ds.getLabels()
        feats = feats if isinstance(feats, list) else [feats]
        labs = labs if isinstance(labs, list) else [labs]
        for n, f in zip(tc.dataSetFeatureMapping, feats):
            ph[n] = f
        for n, l in zip(tc.dataSetLabelMapping, labs):
            ph[n] = l
        if tc.dataSetFeatureMaskMapping:
            fm = ds.getFeaturesMaskArrays() if hasattr(ds, "getFeaturesMaskArrays") else [ds.getFeaturesMaskArray()]
            for n, m in zip(tc.dataSetFeatureMaskMapping, fm or []):
                if m is not None:
                    ph[n] = m
        if tc.dataSetLabelMaskMapping:
            lm = ds.getLabelsMaskArrays() if hasattr(ds, "getLabelsMaskArrays") else [ds.getLabelsMaskArray()]
            for n, m in zip(tc.dataSetLabelMaskMapping, lm or []):
                if m is not None:
                    ph[n] = m
        return ph

    # -- fused update path: all trainable variables become views of ONE flat fp32 vector (the DL4J network layout) the first
    # time they are trained on the GPU, their updater states views of one flat state vector (per-variable layout kept, so
    # the .fb serialization is unchanged), and a training step applies updater + l1/l2/weight decay for every variable in a
    # single launch of the segmented updater kernel (native/csrc/updaters.hip) instead of ~6 launches and a table upload per
    # variable. Reference: TrainingSession.trainingIteration applies one GradientUpdater per variable.
    def _fused_update(self, grads: Dict[str, torch.Tensor], in_capture=False) -> bool:
        from ..native import lib as NL
        from ..ops import updater as U
        tc = self.trainingConfig
        names = list(grads.keys())
        kind = U.KIND.get(getattr(tc.updater, "KIND", None))
        if kind is None or not names:
            return False
        if not all(isinstance(r, (LR.L1Regularization, LR.L2Regularization, LR.WeightDecay)) for r in tc.regularization):
            return False
        arrs = [self._arrays[n] for n in names]
        if not all(a.dtype == torch.float32 and NL.use_hip(a) for a in arrs):
            return False
        it, ep = tc.iterationCount, tc.epochCount
        fu = getattr(self, "_fused", None)
        sizes = [int(a.numel()) for a in arrs]
        if fu is None or fu["names"] != names or fu["sizes"] != sizes:
            mult = U.STATE_MULT[kind]
            total = sum(sizes)
            flat = torch.empty(max(total, 1), dtype=torch.float32, device=arrs[0].device)
            state = torch.zeros(max(mult * total, 1), dtype=torch.float32, device=arrs[0].device)
            segs, off, soff = [], 0, 0
            init = tc.updater.state_init_value() if hasattr(tc.updater, "state_init_value") else 0.0
            for n, a, sz in zip(names, arrs, sizes):
                segs.append(U.Segment(off, sz, soff, kind, divide_by_batch=False))
                view = state[soff:soff + mult * sz]
                old = self.updaterState.get(n)
                if old is not None and old.numel() == view.numel():
                    view.copy_(old.reshape(-1))
                elif init != 0.0 and mult:
                    view.fill_(init)
                if mult:
                    self.updaterState[n] = view
                off += sz
                soff += mult * sz
            fu = self._fused = {"names": names, "sizes": sizes, "flat": flat, "state": state, "su": U.SegmentedUpdater(segs, flat.device),
                                "key": None}
        flat, off = fu["flat"], 0
        base = flat.data_ptr()
        for n, a, sz in zip(names, arrs, sizes):            # (re)bind variables that are not views of the flat vector (first step, setArray)
            if a.data_ptr() != base + 4 * off or not a.is_contiguous():
                flat[off:off + sz].copy_(a.reshape(-1))
                self._arrays[n] = flat[off:off + sz].view(a.shape)
            off += sz
        lr, a_, b_, eps = tc.updater.segment_hparams(it, ep)
        l1 = sum(r.l1.valueAt(it, ep) for r in tc.regularization if isinstance(r, LR.L1Regularization))
        l2 = sum(r.l2.valueAt(it, ep) for r in tc.regularization if isinstance(r, LR.L2Regularization))
        wds = [r for r in tc.regularization if isinstance(r, LR.WeightDecay)]
        wd = sum(r.coeff.valueAt(it, ep) for r in wds)
        wd_lr = all(r.applyLR for r in wds) if wds else True
        key = (lr, a_, b_, eps, l1, l2, wd, wd_lr)
        su = fu["su"]
        if key != fu["key"]:
            for sg in su.segments:
                sg.lr, sg.a, sg.b, sg.eps, sg.l1, sg.l2, sg.wd, sg.wd_apply_lr = lr, a_, b_, eps, l1, l2, wd, wd_lr
            su._tables_dirty = True
            fu["key"] = key
        with torch.no_grad():
            if fu.get("flat_g") is None:
                fu["flat_g"] = torch.empty_like(flat)
                offs, o_ = [], 0
                for sz in sizes:
                    offs.append(o_)
                    o_ += sz
                fu["g_views"] = [fu["flat_g"][o_:o_ + sz] for o_, sz in zip(offs, sizes)]
            flat_g = fu["flat_g"]
            # multi-tensor copy (with the bf16 -> fp32 conversion of shadow-leaf gradients) instead of one cast + cat per variable
            pairs = [(v, grads[n].detach().reshape(-1)) for v, n in zip(fu["g_views"], names)]
            pairs = [(v, g) for v, g in pairs if g.data_ptr() != v.data_ptr()]
            if pairs:
                torch._foreach_copy_([v for v, _ in pairs], [g for _, g in pairs])
            if not tc.minimize:
                flat_g.neg_()
            if not in_capture:             # a captured step relies on the kernel's own device-side iteration increment
                su.set_iteration(it)
            su.step(flat, flat_g, fu["state"], 1)
        return True

    # -- HIP-graph capture of the whole SameDiff training iteration (forward interpreter + autograd backward + fused update).
    # The session is a Python dataflow interpreter: a BERT-base step issues ~1.5k small launches and is host bound; replaying
    # the captured iteration removes the interpreter and the launch overhead from the steady state. Shapes must be static
    # (one capture per placeholder signature); graphs with random ops or listeners that need activations stay eager.
    def enableGraphCapture(self, on=True):
        self._capture_on = bool(on)
        self._cap = None
        return self

    _RANDOM_OPS = ("dropout", "random", "alpha_dropout", "dropout_inverted", "randomuniform", "randomnormal", "random_bernoulli")

    def _capture_eligible(self, ls, required):
        if not getattr(self, "_capture_on", False) or self.device.type != "cuda" or required:
            return False
        if any(getattr(l, "needsPerOpCallbacks", False) for l in ls):
            return False
        if getattr(self, "_cap_static_ok", None) is None:
            names = [getattr(o, "opName", "") or "" for o in self.ops.values()]
            self._cap_static_ok = not any(any(r in n.lower() for r in self._RANDOM_OPS) for n in names)
        return self._cap_static_ok

    def _captured_step(self, ph, trainable):
        """One training iteration through the captured graph; returns the loss values, or None when this signature cannot be
        captured (the caller then runs the eager step)."""
        tc = self.trainingConfig
        ph = self._prep_placeholders(ph)
        key = tuple(sorted((n, tuple(t.shape), str(t.dtype)) for n, t in ph.items())) + (tuple(trainable),)
        cap = getattr(self, "_cap", None)
        if cap is None or cap["key"] != key:
            cap = self._cap = {"key": key, "warm": 0, "graph": None}
        if cap["warm"] < 2:                       # eager warm-up: lazy allocations, flat variable vector, library handles
            cap["warm"] += 1
            return None
        if cap["graph"] is None:
            try:
                static = {n: t.clone() for n, t in ph.items()}
                self._fused["su"].set_iteration(tc.iterationCount)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    losses, grads, _ = self._loss_and_grads(static, trainable, device_losses=True, shadow_ok=True)
                    ok = self._fused_update(grads, in_capture=True)
                if not ok:
                    raise RuntimeError("fused update not available")
                cap.update(graph=g, static=static, losses=losses, grads=grads, hp=self._fused["key"])
            except Exception as e:            # a host synchronisation inside some op, unsupported updater, ...
                import warnings
                warnings.warn("SameDiff graph capture disabled for this graph: %s" % (e,))
                self._capture_on = False
                self._cap = None
                torch.cuda.synchronize()
                return None
        for n, t in ph.items():
            cap["static"][n].copy_(t)
        it, ep = tc.iterationCount, tc.epochCount
        fu = self._fused
        lr, a_, b_, eps = tc.updater.segment_hparams(it, ep)
        if (lr, a_, b_, eps) != fu["key"][:4]:                  # schedules: refresh the device table the captured kernel reads
            for sg in fu["su"].segments:
                sg.lr, sg.a, sg.b, sg.eps = lr, a_, b_, eps
            fu["key"] = (lr, a_, b_, eps) + tuple(fu["key"][4:])
            fu["su"]._upload()
        cap["graph"].replay()
        self._last_grads.update(cap["grads"])
        return [float(l) for l in cap["losses"]]

    def _apply_updates(self, grads: Dict[str, torch.Tensor], at):
        tc = self.trainingConfig
        if self._fused_update(grads):
            return
        it, ep = tc.iterationCount, tc.epochCount
        lr = tc.updater.getLearningRate(it, ep) if tc.updater.hasLearningRate() else 1.0
        with torch.no_grad():
            for n, g in grads.items():
                p = self._arrays[n]
                g = g.detach().to(torch.float32).clone()
                if not tc.minimize:
                    g = -g
                for r in tc.regularization:
                    if r.applyStep() == LR.Regularization.BEFORE_UPDATER:
                        if isinstance(r, LR.L1Regularization):
                            g += r.l1.valueAt(it, ep) * torch.sign(p)
                        else:
                            g += r.l2.valueAt(it, ep) * p
                self._updaters[n].applyUpdater(g, it, ep)
                for r in tc.regularization:
                    if r.applyStep() == LR.Regularization.POST_UPDATER:
                        g += r.coeff.valueAt(it, ep) * (lr if r.applyLR else 1.0) * p
                p.sub_(g.reshape(p.shape).to(p.dtype))

    def calcRegularizationScore(self):
        tc = self.trainingConfig
        s = 0.0
        if tc:
            for n in self.trainableVariables():
                for r in tc.regularization:
                    s += float(r.score(self._arrays[n], tc.iterationCount, tc.epochCount))
        return s

    def evaluate(self, iterator, outputVariable, *evaluations, labelIndex=0):
        outputVariable = outputVariable.name if isinstance(outputVariable, SDVariable) else outputVariable
        tc = self.trainingConfig
        if iterator.resetSupported():
            iterator.reset()
        while iterator.hasNext():
            ds = iterator.next()
            ph = self._placeholders_from_dataset(ds)
            ph = {k: v for k, v in ph.items() if k in (tc.dataSetFeatureMapping + tc.dataSetFeatureMaskMapping)}
            out = self.output(ph, [outputVariable])[outputVariable]
            labs = ds.getLabels()
            lab = labs[labelIndex] if isinstance(labs, list) else labs
            for e in evaluations:
                e.eval(_tt(lab, out.device), out)
        return evaluations[0] if len(evaluations) == 1 else list(evaluations)

    # ------------------------------------------------------------------ serialization
    def asFlatBuffers(self, saveUpdaterState=True) -> bytes:
        from .serde import to_flatbuffers
        return to_flatbuffers(self, saveUpdaterState)

    def save(self, path, saveUpdaterState=True):
        with open(str(path), "wb") as f:
            f.write(self.asFlatBuffers(saveUpdaterState))

    asFlatFile = save

    @staticmethod
    def fromFlatBuffers(buf: bytes, loadUpdaterState=True, device=None) -> "SameDiff":
        from .serde import from_flatbuffers
        return from_flatbuffers(buf, loadUpdaterState, device)

    @staticmethod
    def load(path, loadUpdaterState=True, device=None) -> "SameDiff":
        with open(str(path), "rb") as f:
            return SameDiff.fromFlatBuffers(f.read(), loadUpdaterState, device)

    fromFlatFile = load

    # ------------------------------------------------------------------ control-flow builders (ControlFlow.java)
    def addArgumentInterceptor(self, interceptor):
        """Every SDVariable argument of ops created while the interceptor is installed passes through it (reference:
        SameDiff.addArgumentInterceptor / ArgumentInterceptor.java). Interceptors nest; the newest runs first."""
        prev = getattr(self, "_arg_interceptor", None)
        self._icpt_stack = getattr(self, "_icpt_stack", []) + [prev]
        self._arg_interceptor = (lambda v: prev(interceptor(v))) if prev is not None else interceptor

    def removeArgumentInterceptor(self):
        self._arg_interceptor = self._icpt_stack.pop() if getattr(self, "_icpt_stack", None) else None

    def _paused_interceptor(self):
        sd = self

        class _P:
            def __enter__(self_):
                self_.saved = getattr(sd, "_arg_interceptor", None)
                sd._arg_interceptor = None

            def __exit__(self_, *exc):
                sd._arg_interceptor = self_.saved
                return False
        return _P()

    def ifCond(self, cond: Callable, trueBody: Callable, falseBody: Callable, name=None):
        """Builds Switch/Merge: both bodies are traced into the graph, only the live one executes. Values from outside a
        body are routed through a Switch on the predicate (an argument interceptor does this while the body is traced, as
        the reference's ifCond does), so a body is gated even when it ignores its Switch-output argument."""
        name = self._unique(name or "cond")
        pred = cond(self)
        sw_f, sw_t = self._op("switch", [pred, pred], nOut=2, name=name + "/switch", opType=OpType.LOGIC, opNum=LOGIC_NUM["switch"])

        def gate(branch, own):
            outside, cache = set(self.variables), {}

            def icpt(v):
                if v.name not in outside or v.name == own:
                    return v
                if v.name not in cache:
                    with self._paused_interceptor():
                        cache[v.name] = self._op("switch", [v, pred], nOut=2, name="sw_" + v.name.replace("/", "_").replace(":", "_"),
                                                 opType=OpType.LOGIC, opNum=LOGIC_NUM["switch"])[branch]
                return cache[v.name]
            return icpt

        with self.withNameScope(name + "/true"):
            t_in = self._op("identity", [sw_t])[0]
            self.addArgumentInterceptor(gate(1, t_in.name))
            try:
                before = set(self.variables)
                tv = trueBody(self, t_in)
                if tv.name in before:
                    tv = self._op("identity", [tv])[0]          # a body returning an outer value as-is still gets gated
            finally:
                self.removeArgumentInterceptor()
        with self.withNameScope(name + "/false"):
            f_in = self._op("identity", [sw_f])[0]
            self.addArgumentInterceptor(gate(0, f_in.name))
            try:
                before = set(self.variables)
                fv = falseBody(self, f_in)
                if fv.name in before:
                    fv = self._op("identity", [fv])[0]
            finally:
                self.removeArgumentInterceptor()
        return self._op("merge", [fv, tv], name=name + "/merge", opType=OpType.LOGIC, opNum=LOGIC_NUM["merge"])[0]

    def whileLoop(self, loopVars: Sequence[SDVariable], cond: Callable, body: Callable, name=None) -> List[SDVariable]:
        """TF-style while: Enter -> Merge -> cond -> LoopCond -> Switch -> body -> NextIteration, Exit."""
        frame = self._unique(name or "while")
        L_ = OpType.LOGIC
        enters = []
        for k, v in enumerate(loopVars):
            e = self._op("enter", [v], name="%s/enter_%d" % (frame, k), opType=L_, opNum=LOGIC_NUM["enter"])[0]
            self.ops[self._producer[e.name]].frameName = frame
            enters.append(e)
        merges, merge_ops = [], []
        for k, e in enumerate(enters):
            m = self._op("merge", [e, e], name="%s/merge_%d" % (frame, k), opType=L_, opNum=LOGIC_NUM["merge"])[0]
            merges.append(m)
            merge_ops.append(self._producer[m.name])
        with self.withNameScope(frame + "/cond"):
            c = cond(self, merges)
        lc = self._op("loop_cond", [c], name=frame + "/loop_cond", opType=L_, opNum=LOGIC_NUM["loop_cond"])[0]
        exits, body_in = [], []
        for k, m in enumerate(merges):
            sf, st = self._op("switch", [m, lc], nOut=2, name="%s/switch_%d" % (frame, k), opType=L_, opNum=LOGIC_NUM["switch"])
            exits.append(self._op("exit", [sf], name="%s/exit_%d" % (frame, k), opType=L_, opNum=LOGIC_NUM["exit"])[0])
            body_in.append(st)
        with self.withNameScope(frame + "/body"):
            new = body(self, body_in)
        for k, nv in enumerate(new):
            ni = self._op("next_iteration", [nv], name="%s/next_%d" % (frame, k), opType=L_, opNum=LOGIC_NUM["next_iteration"])[0]
            mo = self.ops[merge_ops[k]]
            old = mo.inputs[1]
            mo.inputs[1] = ni.name
            if mo.name in self._consumers.get(old, []):
                self._consumers[old].remove(mo.name) if self._consumers[old].count(mo.name) > 1 else None
            self._consumers[ni.name].append(mo.name)
        return exits
