"""Numerical gradient checking (reference: deeplearning4j-nn/.../gradientcheck/GradientCheckUtil.java:
defaults epsilon 1e-6, maxRelError 1e-3, minAbsoluteError 1e-8; requires DOUBLE networks :201-210)."""
from __future__ import annotations

import torch


class GradientCheckUtil:
    @staticmethod
    def checkGradients(net, x, y, epsilon=1e-6, maxRelError=1e-3, minAbsoluteError=1e-8, featuresMask=None,
                       labelsMask=None, exclude=(), maxPerParam=24, print_results=False, subset_seed=12345):
        if not net._initialized:
            net.init()
        if net._params.dtype != torch.float64:
            raise ValueError("Gradient checks require a DOUBLE network (dataType('DOUBLE'))")
        is_cg = hasattr(net, "topologicalOrder")

        def score_and_grad():
            if is_cg:
                fm = [featuresMask] if featuresMask is not None and not isinstance(featuresMask, list) else featuresMask
                lm = [labelsMask] if labelsMask is not None and not isinstance(labelsMask, list) else labelsMask
                net.computeGradientAndScore(x if isinstance(x, list) else [x], y if isinstance(y, list) else [y], fm, lm)
            else:
                net.computeGradientAndScore(x, y, featuresMask, labelsMask)
            mb = net._last_batch
            s = net._score.double() * (mb if net._minibatch else 1.0)   # un-normalised sum, like the reference
            r = net.calcRegularizationScore()
            tot = s + (r.double() * mb if r is not None else 0.0)
            net.clearLayerMaskArrays()
            return float(tot)

        torch.manual_seed(subset_seed)
        s0 = score_and_grad()
        # analytic gradient incl. regularisation:  dScore/dw = grad_sum + mb * dReg/dw
        mb = net._last_batch
        analytic = net._grads.clone()
        from ..nd4j.learning import L1Regularization, L2Regularization, WeightDecay
        from ..nn.network_base import _strided_view
        for e in net._entries:
            for r in e.regs:
                p = net._params[e.off:e.off + e.spec.numel]
                if isinstance(r, L1Regularization):
                    analytic[e.off:e.off + e.spec.numel] += mb * r.l1.valueAt(0, 0) * torch.sign(p)
                elif isinstance(r, L2Regularization):
                    analytic[e.off:e.off + e.spec.numel] += mb * r.l2.valueAt(0, 0) * p
                elif isinstance(r, WeightDecay):
                    analytic[e.off:e.off + e.spec.numel] += mb * r.coeff.valueAt(0, 0) * p
        g = torch.Generator().manual_seed(subset_seed)
        fails, total = 0, 0
        max_err = 0.0
        for e in net._entries:
            if e.spec.key in exclude or e.spec.role == "stat":
                continue
            n = e.spec.numel
            idxs = range(n) if n <= maxPerParam else torch.randperm(n, generator=g)[:maxPerParam].tolist()
            for j in idxs:
                i = e.off + j
                orig = float(net._params[i])
                net._params[i] = orig + epsilon
                net._refresh_replicas()
                sp = score_and_grad()
                net._params[i] = orig - epsilon
                net._refresh_replicas()
                sm = score_and_grad()
                net._params[i] = orig
                num = (sp - sm) / (2 * epsilon)
                an = float(analytic[i])
                denom = abs(num) + abs(an)
                rel = abs(num - an) / denom if denom > 0 else 0.0
                total += 1
                if rel > maxRelError and abs(num - an) > minAbsoluteError:
                    fails += 1
                    if print_results or fails <= 10:
                        print("Param %s_%s[%d] FAILED: analytic=%.10g numeric=%.10g relError=%.4g" % (
                            net._layer_key(e.layer_idx), e.spec.key, j, an, num, rel))
                max_err = max(max_err, rel if abs(num - an) > minAbsoluteError else 0.0)
        net._refresh_replicas()
        if print_results or fails:
            print("GradientCheckUtil: %d params checked, %d failed, max rel error %.4g" % (total, fails, max_err))
        return fails == 0
