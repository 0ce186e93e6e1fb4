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


# ------------------------------------------------------------------------------------------------ whole-registry sweep
def _as_list(o):
    return list(o) if isinstance(o, (list, tuple)) else [o]


def _cmp(a, e, rtol, atol):
    if hasattr(a, "values") and hasattr(e, "values") and not isinstance(a, torch.Tensor):      # StringArray
        return None if (a.values == e.values) else "strings differ: %s vs %s" % (a.values[:6], e.values[:6])
    if not isinstance(a, torch.Tensor):
        return "output is not a tensor"
    e = e if isinstance(e, torch.Tensor) else torch.as_tensor(e)
    if list(a.shape) != list(e.shape):
        return "shape %s != reference %s" % (list(a.shape), list(e.shape))
    a, e = a.detach().cpu(), e.detach().cpu()
    if a.dtype == torch.bool or e.dtype == torch.bool:
        return None if torch.equal(a.bool(), e.bool()) else "boolean values differ"
    if not torch.allclose(a.double(), e.double(), rtol=rtol, atol=atol, equal_nan=True):
        return "values differ (max abs err %.3g)" % float((a.double() - e.double()).abs().max())
    return None


_RANDOM_PREFIX = ("random", "dropout", "alpha_dropout", "get_seed", "set_seed")


def _numeric_grad_check(name, case, device):
    """fp64 central differences of sum(out0 * gy) against the op's ``_bp`` op. Returns None | error string."""
    from ..nd4j.ops.custom import exec_op
    ins = [(t.double() if t.is_floating_point() else t).to(device) for t in case.ins]
    kw = dict(iArgs=case.i, tArgs=case.t, bArgs=case.b)
    out0 = _as_list(exec_op(name, ins, **kw))[0]
    gy = torch.randn(out0.shape, generator=torch.Generator().manual_seed(99), dtype=torch.float64).to(device)
    gin = case.gin if case.gin is not None else [k for k, t in enumerate(ins) if t.is_floating_point()]
    got = _as_list(exec_op(name + "_bp", ins + [gy] + ([out0] if case.bp_out else []), **kw))
    eps = 1e-5
    for pos, k in enumerate(gin):
        if pos >= len(got) or got[pos] is None:
            return "%s_bp returned %d outputs, gradient of input %d missing" % (name, len(got), k)
        x = ins[k]
        num = torch.zeros_like(x)
        flat, nf = x.reshape(-1), num.reshape(-1)
        for j in range(flat.numel()):
            old = float(flat[j])
            flat[j] = old + eps
            fp = float((_as_list(exec_op(name, ins, **kw))[0].double() * gy).sum())
            flat[j] = old - eps
            fm = float((_as_list(exec_op(name, ins, **kw))[0].double() * gy).sum())
            flat[j] = old
            nf[j] = (fp - fm) / (2 * eps)
        g = got[pos].double().reshape(num.shape) if got[pos].numel() == num.numel() else got[pos].double()
        if list(g.shape) != list(num.shape):
            return "gradient of input %d has shape %s, input has %s" % (k, list(g.shape), list(num.shape))
        err = float((g - num).abs().max())
        if err > 1e-5 + 2e-4 * float(num.abs().max()):
            return "gradient of input %d: max abs err %.3g vs central differences (max |grad| %.3g)" % (k, err, float(num.abs().max()))
    return None


def sweep(device="cpu", names=None, check_grad=True):
    """Walk the op registry through the case table (op_cases.py). Per op: status in
    {'reference', 'property', 'executed', 'failed', 'uncovered'} + details; for ops with a registered ``_bp`` and a differentiable
    case, the numeric gradient verdict. Mirrors OpValidation.java's coverage accounting (logCoverageInformation)."""
    from ..nd4j.ops.custom import OpExecutioner, DynamicCustomOp, exec_op, USER_DEFINED_OPS
    from .op_cases import cases
    table = cases()
    all_ops = sorted(n for n in OpExecutioner.getInstance().getCustomOperations() if n not in USER_DEFINED_OPS)
    if names is not None:
        all_ops = [n for n in all_ops if n in set(names)]
    report = {}
    for name in all_ops:
        entry = {"status": "uncovered", "cases": 0, "errors": [], "grad": "n/a"}
        report[name] = entry
        cs = table.get(name)
        if not cs:
            continue
        best = None
        for c in cs:
            entry["cases"] += 1
            try:
                ins = [t.to(device) for t in c.ins]
                kw = dict(iArgs=c.i, tArgs=c.t, bArgs=c.b)
                outs = _as_list(exec_op(name, ins, **kw))
                # shape function (runs the op on meta tensors) against what came out
                try:
                    shapes = [list(s) for s in DynamicCustomOp(name, ins, iArgs=c.i, tArgs=c.t, bArgs=c.b).calculateOutputShape()]
                    real = [list(o.shape) for o in outs if isinstance(o, torch.Tensor)]
                    if shapes and shapes[:len(real)] != real[:len(shapes)] and all(t.is_floating_point() for t in ins):
                        entry["errors"].append("%s: calculateOutputShape %s != executed %s" % (c.tag or "case", shapes, real))
                except Exception:
                    pass                                  # data-dependent output shapes have no static shape function
                level = "executed"
                if c.ref is not None:
                    want = _as_list(c.ref(*c.ins))
                    if getattr(c, "canon", None) is not None:
                        outs, want = c.canon(list(outs)), c.canon(list(want))
                    if len(want) > len(outs):
                        entry["errors"].append("%s: %d outputs, reference has %d" % (c.tag or "case", len(outs), len(want)))
                    for k, e in enumerate(want):
                        if k >= len(outs):
                            entry["errors"].append("%s: %d outputs, reference has %d" % (c.tag or "case", len(outs), len(want)))
                            break
                        err = _cmp(outs[k], e, c.rtol, c.atol)
                        if err:
                            entry["errors"].append("%s: output %d %s" % (c.tag or "case", k, err))
                    level = "reference"
                elif c.check is not None:
                    err = c.check([o.detach().cpu() if isinstance(o, torch.Tensor) else o for o in outs], c.ins)
                    if err:
                        entry["errors"].append("%s: %s" % (c.tag or "case", err))
                    level = "property"
                elif not name.startswith(_RANDOM_PREFIX):
                    again = _as_list(exec_op(name, ins, **kw))
                    for a, b in zip(outs, again):
                        if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and not torch.equal(a, b) and not (a != a).any():
                            entry["errors"].append("%s: not deterministic" % (c.tag or "case"))
                order = ["executed", "property", "reference"]
                best = level if best is None or order.index(level) > order.index(best) else best
                if check_grad and c.grad and (name + "_bp") in report_names(all_ops):
                    g = _numeric_grad_check(name, c, device)
                    entry["grad"] = "ok" if g is None and entry["grad"] != "failed" else "failed"
                    if g:
                        entry["errors"].append("gradient: " + g)
            except Exception as ex:                       # noqa: BLE001 — the sweep reports, the test asserts
                entry["errors"].append("%s: %s: %s" % (c.tag or "case", type(ex).__name__, str(ex)[:160]))
        entry["status"] = "failed" if entry["errors"] else (best or "uncovered")
    from .op_cases import SCENARIOS, EXTERNAL
    from .op_cases_extra import EXTRA_SCENARIOS
    for names_, fn in SCENARIOS + EXTRA_SCENARIOS:
        try:
            err = fn(device)
        except Exception as ex:                           # noqa: BLE001
            err = "%s: %s" % (type(ex).__name__, str(ex)[:160])
        for n in names_:
            if n in report and report[n]["status"] in ("uncovered", "executed"):
                report[n].update(status="failed" if err else "property", cases=max(1, report[n]["cases"]), errors=[err] if err else [])
    for n, where in EXTERNAL.items():
        if n in report and report[n]["status"] == "uncovered":
            report[n].update(status="external", errors=[], where=where)
    # a `_bp` op is validated by the numeric check of its forward op
    for name, e in report.items():
        if name.endswith("_bp") and e["status"] == "uncovered":
            f = report.get(name[:-3])
            if f is not None and f["grad"] == "ok":
                e["status"], e["cases"] = "reference", 1
            elif f is not None and f["grad"] == "failed":
                e["status"], e["errors"] = "failed", ["numeric gradient check of %s failed" % name[:-3]]
    return report


def report_names(all_ops, _cache={}):
    k = id(all_ops)
    if k not in _cache:
        _cache.clear()
        _cache[k] = set(all_ops)
    return _cache[k]


def summarize(report):
    tiers = {"reference": 0, "property": 0, "executed": 0, "external": 0, "failed": 0, "uncovered": 0}
    for e in report.values():
        tiers[e["status"]] += 1
    total = len(report)
    covered = tiers["reference"] + tiers["property"] + tiers["executed"] + tiers["external"]
    grad_ok = sum(1 for e in report.values() if e["grad"] == "ok")
    grad_failed = sum(1 for e in report.values() if e["grad"] == "failed")
    return {"total": total, **tiers, "covered": covered, "covered_pct": 100.0 * covered / max(total, 1), "grad_ok": grad_ok,
            "grad_failed": grad_failed}


def write_report(report, path, title="Op validation sweep"):
    s = summarize(report)
    lines = ["# %s" % title, "",
             "%d registered custom ops: **%d validated against an independent reference**, %d by a property check, %d executed "
             "(+ shape function + determinism), %d graph-level ops covered by named tests, %d failed, %d without a case -> %.1f %% "
             "covered. Numeric fp64 gradient checks of `_bp` ops: %d passed, %d failed." % (
                 s["total"], s["reference"], s["property"], s["executed"], s["external"], s["failed"], s["uncovered"], s["covered_pct"],
                 s["grad_ok"], s["grad_failed"]), "",
             "| op | status | cases | gradient | notes |", "|---|---|---:|---|---|"]
    for n in sorted(report):
        e = report[n]
        lines.append("| %s | %s | %d | %s | %s |" % (n, e["status"], e["cases"], e["grad"], ("; ".join(e["errors"]) or e.get("where", ""))[:200].replace("|", "/")))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return s


OpValidation.sweep = staticmethod(sweep)
OpValidation.summarize = staticmethod(summarize)
OpValidation.writeReport = staticmethod(write_report)
