"""Gradient checks and op-coverage accounting (reference: nd4j-api autodiff/validation/{GradCheckUtil,
OpValidation,TestCase,OpTestCase}.java; tolerances GradCheckUtil.java:51-53: eps 1e-5, maxRelError 1e-5, minAbsError 1e-6)."""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch


class GradCheckUtil:
    DEFAULT_EPS, DEFAULT_MAX_REL_ERROR, DEFAULT_MIN_ABS_ERROR = 1e-5, 1e-5, 1e-6

    @staticmethod
    def checkGradients(sd, placeholders: Dict[str, torch.Tensor], eps=1e-5, maxRelError=1e-5, minAbsError=1e-6, printResults=False,
                       exitOnFirstFailure=True, skipVariables: Sequence[str] = (), maxPerParam: Optional[int] = None, seed=12345):
        """Central differences on every trainable fp64 variable vs the session's analytic gradients."""
        names = [n for n in sd.trainableVariables() if n not in skipVariables]
        for n in names:
            if sd._arrays[n].dtype != torch.float64:
                raise ValueError("gradient checks need DOUBLE variables ('%s' is %s)" % (n, sd._arrays[n].dtype))
        analytic = {n: g.tensor().clone() for n, g in sd.calculateGradients(placeholders, names).items()}

        def score():
            res = sd.output(placeholders, list(sd.lossVariables))
            return sum(float(v.double().sum()) for v in res.values())

        g = torch.Generator().manual_seed(seed)
        ok = True
        for n in names:
            p = sd._arrays[n]
            flat = p.reshape(-1)
            idxs = range(flat.numel()) if (maxPerParam is None or flat.numel() <= maxPerParam) else \
                torch.randperm(flat.numel(), generator=g)[:maxPerParam].tolist()
            for k in idxs:
                orig = float(flat[k])
                flat[k] = orig + eps
                sp = score()
                flat[k] = orig - eps
                sm = score()
                flat[k] = orig
                num = (sp - sm) / (2 * eps)
                ana = float(analytic[n].reshape(-1)[k])
                denom = abs(num) + abs(ana)
                rel = abs(num - ana) / denom if denom > 0 else 0.0
                if rel > maxRelError and abs(num - ana) > minAbsError:
                    ok = False
                    if printResults:
                        print("FAIL %s[%d]: numeric %.10g analytic %.10g rel %.3g" % (n, k, num, ana, rel))
                    if exitOnFirstFailure:
                        return False
        return ok


class TestCase:
    def __init__(self, sd):
        self.sd = sd
        self.expected: Dict[str, torch.Tensor] = {}
        self.placeholders: Dict[str, torch.Tensor] = {}
        self.gradCheck = True
        self.eps, self.maxRel, self.minAbs = 1e-5, 1e-5, 1e-6
        self.testName = None

    def expectedOutput(self, name, arr):
        self.expected[name] = arr.tensor() if hasattr(arr, "tensor") else torch.as_tensor(arr)
        return self

    expected_ = expectedOutput

    def placeholderValue(self, name, arr):
        self.placeholders[name] = arr.tensor() if hasattr(arr, "tensor") else torch.as_tensor(arr)
        return self

    def gradientCheck(self, b):
        self.gradCheck = b
        return self

    def testName_(self, n):
        self.testName = n
        return self


class OpTestCase:
    """Single-op test: a DynamicCustomOp plus expected output arrays (values or a predicate) and, optionally, the expected
    output shapes — checked against ``calculateOutputShape`` before execution (reference:
    org/nd4j/autodiff/validation/OpTestCase.java)."""

    def __init__(self, op):
        self.op = op
        self.expected: Dict[int, object] = {}
        self.expShapes: Dict[int, list] = {}
        self.rtol, self.atol = 1e-5, 1e-6

    def expectedOutput(self, outputNum: int, expected):
        """``expected``: array-like, or a callable ``fn(actual_tensor) -> None | error string``."""
        if callable(expected):
            self.expected[outputNum] = expected
        else:
            self.expected[outputNum] = expected.tensor() if hasattr(expected, "tensor") else torch.as_tensor(expected)
            self.expShapes.setdefault(outputNum, list(self.expected[outputNum].shape))
        return self

    def expectedShape(self, outputNum: int, shape):
        self.expShapes[outputNum] = [int(s) for s in shape]
        return self

    def tolerance(self, rtol, atol):
        self.rtol, self.atol = rtol, atol
        return self


class OpValidation:
    """Runs TestCases and keeps per-op coverage counts (forward validated / gradient validated)."""
    fwdPassed: Dict[str, int] = {}
    gradPassed: Dict[str, int] = {}

    @staticmethod
    def validateOp(tc: OpTestCase) -> Optional[str]:
        op = tc.op
        if tc.expShapes:
            shapes = [list(s) for s in op.calculateOutputShape()]
            for k, want in tc.expShapes.items():
                if k >= len(shapes):
                    return "%s: calculateOutputShape returned %d shapes, expected output %d" % (op.opName, len(shapes), k)
                if shapes[k] != list(want):
                    return "%s: calculateOutputShape[%d] = %s, expected %s" % (op.opName, k, shapes[k], list(want))
        outs = op.exec()
        outs = outs if isinstance(outs, (list, tuple)) else [outs]
        for k, e in tc.expected.items():
            if k >= len(outs):
                return "%s: op produced %d outputs, expected output %d" % (op.opName, len(outs), k)
            a = outs[k].tensor() if hasattr(outs[k], "tensor") else outs[k]
            if callable(e):
                err = e(a)
                if err:
                    return "%s: output %d: %s" % (op.opName, k, err)
                continue
            if list(a.shape) != list(e.shape):
                return "%s: output %d shape %s != expected %s" % (op.opName, k, list(a.shape), list(e.shape))
            if not torch.allclose(a.double().cpu(), e.double().cpu(), rtol=tc.rtol, atol=tc.atol, equal_nan=True):
                return "%s: output %d values differ (max abs err %.3g)" % (op.opName, k, float((a.double().cpu() - e.double().cpu()).abs().max()))
        OpValidation.fwdPassed[op.opName] = OpValidation.fwdPassed.get(op.opName, 0) + 1
        return None

    @staticmethod
    def validate(tc) -> Optional[str]:
        if isinstance(tc, OpTestCase):
            return OpValidation.validateOp(tc)
        sd = tc.sd
        if tc.expected:
            res = sd.output(tc.placeholders, list(tc.expected))
            for n, e in tc.expected.items():
                a = res[n]
                if list(a.shape) != list(e.shape):
                    return "%s: shape %s != expected %s" % (n, list(a.shape), list(e.shape))
                if not torch.allclose(a.double().cpu(), e.double().cpu(), rtol=1e-5, atol=1e-6, equal_nan=True):
                    return "%s: values differ (max abs err %.3g)" % (n, float((a.double().cpu() - e.double().cpu()).abs().max()))
            for o in sd.ops.values():
                OpValidation.fwdPassed[o.opName] = OpValidation.fwdPassed.get(o.opName, 0) + 1
        if tc.gradCheck and sd.lossVariables:
            if not GradCheckUtil.checkGradients(sd, tc.placeholders, tc.eps, tc.maxRel, tc.minAbs):
                return "gradient check failed"
            for o in sd.ops.values():
                OpValidation.gradPassed[o.opName] = OpValidation.gradPassed.get(o.opName, 0) + 1
        return None

    @staticmethod
    def coverage():
        from ..nd4j.ops.custom import OpExecutioner
        all_ops = set(OpExecutioner.getInstance().getCustomOperations())
        return {"total": len(all_ops), "forwardValidated": len(all_ops & set(OpValidation.fwdPassed)),
                "gradientValidated": len(all_ops & set(OpValidation.gradPassed)),
                "notValidated": sorted(all_ops - set(OpValidation.fwdPassed))}
