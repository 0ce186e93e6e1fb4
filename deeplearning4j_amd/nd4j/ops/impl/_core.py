"""Base classes of the op-class hierarchy (reference: nd4j-api linalg/api/ops/{BaseOp, BaseTransformOp, BaseTransformSameOp /
StrictOp / FloatOp / BoolOp / AnyOp, BaseReduceOp and its Float / Same / Bool / Long variants, BaseScalarOp, BaseScalarBoolOp,
BaseIndexAccumulation, BaseReduce3Op, DynamicCustomOp}.java). The per-op classes of this package are generated from the op DSL
(codegen/generator.py generate_op_classes) - one class per op, typed constructor, immediate execution through the executioners -
and registered by native op name in ``impl.OP_CLASSES``.

``doDiff`` is the eager counterpart of the reference's ``doDiff(List<SDVariable>)``: gradients of the op's inputs given gradients of
its outputs. A custom op uses its registered ``<name>_bp`` op when there is one (the analytic backward of libnd4j) and the vector-Jacobian
product of its implementation otherwise; legacy ops always take the latter."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from ..custom import DynamicCustomOp, _IMPL, exec_op
from ..legacy import exec_legacy, op_num


def _t(x):
    t = getattr(x, "tensor", None)
    return x if t is None else (t() if callable(t) else t)


def _vjp(fn, inputs: Sequence[torch.Tensor], grads: Sequence[Optional[torch.Tensor]]) -> List[Optional[torch.Tensor]]:
    leaves = [x.detach().clone().requires_grad_(True) if isinstance(x, torch.Tensor) and x.is_floating_point() else x for x in inputs]
    with torch.enable_grad():
        outs = fn(leaves)
    outs = list(outs) if isinstance(outs, (list, tuple)) else [outs]
    pairs = [(o, g) for o, g in zip(outs, grads) if isinstance(o, torch.Tensor) and o.requires_grad and g is not None]
    diff = [x for x in leaves if isinstance(x, torch.Tensor) and x.requires_grad]
    got = torch.autograd.grad([o for o, _ in pairs], diff, [g.to(o.dtype) for o, g in pairs], allow_unused=True) if pairs and diff else ()
    it = iter(got)
    return [next(it, None) if (isinstance(x, torch.Tensor) and x.requires_grad) else None for x in leaves]


class GeneratedCustomOp(DynamicCustomOp):
    """A custom (declarable) op with a typed constructor. ``OP`` = native op name, ``NAMESPACE`` = DSL namespace, ``N_OUT`` = outputs."""
    OP: str = ""
    NAMESPACE: str = ""
    N_OUT = 1

    def opType(self):
        return "CUSTOM"

    def getNumOutputs(self):
        return self.N_OUT

    def execute(self):
        """Run now; one tensor for single-output ops, a tuple otherwise."""
        r = self.exec()
        return r[0] if len(r) == 1 else tuple(r)

    def doDiff(self, *gradOutputs):
        ins = self.inputArguments()
        grads = [None if g is None else _t(g) for g in gradOutputs]
        bp = self.OP + "_bp"
        if bp in _IMPL and all(g is not None for g in grads):
            try:
                res = exec_op(bp, list(ins) + grads, iArgs=self.iArgs(), tArgs=self.tArgs(), bArgs=self.bArgs(), dArgs=self.ctx.dArgs)
                res = list(res)
                if len(res) >= 1 and all(r is None or r.shape == i.shape for r, i in zip(res, ins)):
                    return res[:len(ins)] + [None] * max(0, len(ins) - len(res))
            except Exception:
                pass                      # argument conventions of a few _bp ops differ from "inputs + gradients": fall through to the VJP
        ctx = self.ctx
        return _vjp(lambda xs: exec_op(self.OP, xs, iArgs=ctx.iArgs, tArgs=ctx.tArgs, bArgs=ctx.bArgs, dArgs=ctx.dArgs), ins, grads)


class BaseLegacyOp:
    """A legacy (opNum-addressed) op: family + name + operands (+ extra args / dimensions / scalar)."""
    FAMILY: str = ""
    OP: str = ""
    NAMESPACE: str = ""

    def __init__(self, inputs, tArgs=None, dimensions=None, scalar=None, keepDims=False):
        self.inputs = [_t(v) for v in inputs]
        self.extraArgs = list(tArgs) if tArgs else None
        self.dimensions = list(dimensions) if dimensions else None
        self.scalarValue = scalar
        self.keepDims = bool(keepDims)
        self.z = None

    def opName(self):
        return self.OP

    def opNum(self) -> int:
        return op_num(self.FAMILY, self.OP)

    def opType(self):
        return self.FAMILY.upper()

    def x(self):
        return self.inputs[0] if self.inputs else None

    def y(self):
        return self.inputs[1] if len(self.inputs) > 1 else None

    def _run(self, inputs):
        return exec_legacy(self.FAMILY, self.OP, inputs, tArgs=self.extraArgs, dimensions=self.dimensions, scalar=self.scalarValue,
                           keepDims=self.keepDims)

    def exec(self):
        self.z = self._run(self.inputs)
        return self.z

    execute = exec

    def doDiff(self, *gradOutputs):
        return _vjp(lambda xs: self._run(xs), self.inputs, [None if g is None else _t(g) for g in gradOutputs])


class BaseTransformOp(BaseLegacyOp):
    pass


class BaseTransformSameOp(BaseTransformOp):
    FAMILY = "transform_same"


class BaseTransformStrictOp(BaseTransformOp):
    FAMILY = "transform_strict"


class BaseTransformFloatOp(BaseTransformOp):
    FAMILY = "transform_float"


class BaseTransformBoolOp(BaseTransformOp):
    FAMILY = "transform_bool"


class BaseTransformAnyOp(BaseTransformOp):
    FAMILY = "transform_any"


class BasePairwiseOp(BaseTransformOp):
    FAMILY = "pairwise"


class BasePairwiseBoolOp(BaseTransformOp):
    FAMILY = "pairwise_bool"


class BaseScalarOp(BaseLegacyOp):
    FAMILY = "scalar"


class BaseScalarBoolOp(BaseLegacyOp):
    FAMILY = "scalar_bool"


class BaseReduceOp(BaseLegacyOp):
    def isKeepDims(self):
        return self.keepDims


class BaseReduceFloatOp(BaseReduceOp):
    FAMILY = "reduce_float"


class BaseReduceSameOp(BaseReduceOp):
    FAMILY = "reduce_same"


class BaseReduceBoolOp(BaseReduceOp):
    FAMILY = "reduce_bool"


class BaseReduceLongOp(BaseReduceOp):
    FAMILY = "reduce_long"


class BaseReduce3Op(BaseReduceOp):
    FAMILY = "reduce3"


class BaseIndexAccumulation(BaseReduceOp):
    FAMILY = "index_reduce"


class BaseSummaryStatsOp(BaseReduceOp):
    FAMILY = "summary_stats"


LEGACY_BASES = {c.FAMILY: c for c in (BaseTransformSameOp, BaseTransformStrictOp, BaseTransformFloatOp, BaseTransformBoolOp, BaseTransformAnyOp,
                                      BasePairwiseOp, BasePairwiseBoolOp, BaseScalarOp, BaseScalarBoolOp, BaseReduceFloatOp, BaseReduceSameOp,
                                      BaseReduceBoolOp, BaseReduceLongOp, BaseReduce3Op, BaseIndexAccumulation, BaseSummaryStatsOp)}
