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


class OpValidation:
    """Runs TestCases and keeps per-op coverage counts (forward validated / gradient validated)."""
    fwdPassed: Dict[str, int] = {}
    gradPassed: Dict[str, int] = {}

    @staticmethod
    def validate(tc: TestCase) -> Optional[str]:
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
