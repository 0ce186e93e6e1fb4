"""ILossFunction implementations (reference: nd4j-api/.../linalg/lossfunctions/impl/*, 19 classes).

API mirrors the Java interface: computeScore / computeScoreArray / computeGradient /
computeGradientAndScore(labels, preOutput, activationFn, mask, average). The softmax+MCXENT
combination takes the fused HIP kernel (ops/softmax.py); everything else is expressed once as a
per-element score in PyTorch and differentiated through the activation with autograd, so every
loss has an exact analytic gradient in any dtype (fp64 gradient checks included).
"""
from __future__ import annotations

import math

import torch

from .activations import ActivationSoftmax, ActivationIdentity, ActivationSigmoid, get_activation

_PKG = "org.nd4j.linalg.lossfunctions.impl."


def _apply_mask(arr, mask):
    if mask is None:
        return arr
    if mask.dim() == 1 or (mask.dim() == 2 and mask.shape[1] == 1 and arr.shape[1] != 1):
        return arr * mask.reshape(-1, 1).to(arr.dtype)
    return arr * mask.to(arr.dtype)


def _loss_dtype(t):
    """Loss math runs in fp32 when the network computes in bf16 / fp16."""
    return t.float() if t.dtype in (torch.bfloat16, torch.float16) else t


class ILossFunction:
    JAVA_CLASS = None
    NAME = None

    def __init__(self, weights=None, **kw):
        self.weights = weights
        for k, v in kw.items():
            setattr(self, k, v)

    # per-element score [mb, nOut] given labels and post-activation output
    def _score_elems(self, labels, out):
        raise NotImplementedError

    def _w(self, arr):
        if self.weights is None:
            return arr
        w = torch.as_tensor(self.weights, dtype=arr.dtype, device=arr.device)
        return arr * w.reshape(1, -1)

    def computeScoreArray(self, labels, preOutput, activationFn, mask=None):
        out = get_activation(activationFn).forward(_loss_dtype(preOutput))
        return _apply_mask(self._w(self._score_elems(labels.to(out.dtype), out)), mask).sum(1)

    def computeScore(self, labels, preOutput, activationFn, mask=None, average=True):
        s = self.computeScoreArray(labels, preOutput, activationFn, mask).sum()
        if average:
            s = s / preOutput.shape[0]
        return s

    def computeGradient(self, labels, preOutput, activationFn, mask=None):
        return self.computeGradientAndScore(labels, preOutput, activationFn, mask, False)[1]

    def computeGradientAndScore(self, labels, preOutput, activationFn, mask=None, average=True):
        """Returns (score tensor (0-d), dL/dpreOutput). Gradient is the SUM over examples (DL4J)."""
        act = get_activation(activationFn)
        # reduced-precision pre-outputs (bf16 / fp16 compute) are scored in fp32 against the un-rounded labels; only the
        # gradient handed back to the layer is cast to the compute type
        z = _loss_dtype(preOutput.detach()).requires_grad_(True)
        with torch.enable_grad():
            out = act.forward(z, True)
            s = _apply_mask(self._w(self._score_elems(labels.to(out.dtype), out)), mask).sum()
        g = torch.autograd.grad(s, z)[0]
        s = s.detach()
        if average:
            s = s / preOutput.shape[0]
        return s, g.to(preOutput.dtype)

    def to_json(self):
        d = {"@class": self.JAVA_CLASS}
        if self.weights is not None:
            d["weights"] = [float(x) for x in self.weights]
        return d

    def name(self):
        return self.NAME

    def __eq__(self, o):
        return type(self) is type(o)

    def __hash__(self):
        return hash(type(self).__name__)


class LossMCXENT(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossMCXENT", "MCXENT"
    softmaxClipEps = 1e-10

    def _score_elems(self, labels, out):
        return -labels * torch.log(out.clamp(self.softmaxClipEps, 1.0 - self.softmaxClipEps)
                                   if self.softmaxClipEps > 0 else out)

    def computeGradientAndScore(self, labels, preOutput, activationFn, mask=None, average=True):
        act = get_activation(activationFn)
        if (isinstance(act, ActivationSoftmax) and self.weights is None and mask is None
                and preOutput.dim() == 2 and labels.shape == preOutput.shape):
            from ..ops.softmax import softmax_xent
            z = preOutput.contiguous()
            score, grad, _ = softmax_xent(z, labels.contiguous())
            s = score[0]
            if average:
                s = s / preOutput.shape[0]
            return s, grad
        return super().computeGradientAndScore(labels, preOutput, activationFn, mask, average)


class LossNegativeLogLikelihood(LossMCXENT):
    JAVA_CLASS, NAME = _PKG + "LossNegativeLogLikelihood", "NEGATIVELOGLIKELIHOOD"


class LossSparseMCXENT(LossMCXENT):
    JAVA_CLASS, NAME = _PKG + "LossSparseMCXENT", "SPARSE_MCXENT"

    def _onehot(self, labels, n):
        idx = labels.reshape(-1).long()
        oh = torch.zeros(idx.shape[0], n, dtype=torch.float32, device=labels.device)
        oh.scatter_(1, idx.view(-1, 1), 1.0)
        return oh

    def computeScoreArray(self, labels, preOutput, activationFn, mask=None):
        return super().computeScoreArray(self._onehot(labels, preOutput.shape[1]), preOutput, activationFn, mask)

    def computeGradientAndScore(self, labels, preOutput, activationFn, mask=None, average=True):
        act = get_activation(activationFn)
        if isinstance(act, ActivationSoftmax) and self.weights is None and mask is None and preOutput.dim() == 2:
            from ..ops.softmax import softmax_xent
            score, grad, _ = softmax_xent(preOutput.contiguous(), None, labels.reshape(-1).long().contiguous())
            s = score[0]
            return (s / preOutput.shape[0] if average else s), grad
        return super().computeGradientAndScore(self._onehot(labels, preOutput.shape[1]), preOutput, activationFn,
                                               mask, average)


class LossBinaryXENT(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossBinaryXENT", "XENT"
    clipEps = 1e-5

    def _score_elems(self, labels, out):
        o = out.clamp(self.clipEps, 1.0 - self.clipEps)
        return -(labels * torch.log(o) + (1 - labels) * torch.log(1 - o))


class LossMSE(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossMSE", "MSE"

    def _score_elems(self, labels, out):
        return (out - labels) ** 2 / labels.shape[1]

    def computeGradientAndScore(self, labels, preOutput, activationFn, mask=None, average=True):
        act = get_activation(activationFn)
        if isinstance(act, ActivationIdentity) and self.weights is None and mask is None:
            d = _loss_dtype(preOutput) - labels.to(_loss_dtype(preOutput).dtype)
            s = (d ** 2).sum() / labels.shape[1]
            return (s / preOutput.shape[0] if average else s), (d * (2.0 / labels.shape[1])).to(preOutput.dtype)
        return super().computeGradientAndScore(labels, preOutput, activationFn, mask, average)


class LossL2(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossL2", "L2"

    def _score_elems(self, labels, out):
        return (out - labels) ** 2


class LossL1(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossL1", "L1"

    def _score_elems(self, labels, out):
        return (out - labels).abs()


class LossMAE(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossMAE", "MEAN_ABSOLUTE_ERROR"

    def _score_elems(self, labels, out):
        return (out - labels).abs() / labels.shape[1]


class LossMAPE(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossMAPE", "MEAN_ABSOLUTE_PERCENTAGE_ERROR"

    def _score_elems(self, labels, out):
        return 100.0 * ((labels - out) / labels).abs() / labels.shape[1]


class LossMSLE(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossMSLE", "MEAN_SQUARED_LOGARITHMIC_ERROR"

    def _score_elems(self, labels, out):
        return (torch.log((out + 1.0) / (labels + 1.0))) ** 2 / labels.shape[1]


class LossHinge(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossHinge", "HINGE"

    def _score_elems(self, labels, out):
        return torch.clamp_min(1.0 - labels * out, 0.0)


class LossSquaredHinge(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossSquaredHinge", "SQUARED_HINGE"

    def _score_elems(self, labels, out):
        return torch.clamp_min(1.0 - labels * out, 0.0) ** 2


class LossKLD(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossKLD", "KL_DIVERGENCE"

    def _score_elems(self, labels, out):
        o = out.clamp(1e-5, 1.0)
        l = labels.clamp(1e-5, 1.0)
        return l * torch.log(l / o)


class LossPoisson(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossPoisson", "POISSON"

    def _score_elems(self, labels, out):
        return out - labels * torch.log(out.clamp_min(1e-10))


class LossCosineProximity(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossCosineProximity", "COSINE_PROXIMITY"

    def _score_elems(self, labels, out):
        yn = labels.norm(dim=1, keepdim=True).clamp_min(1e-12)
        on = out.norm(dim=1, keepdim=True).clamp_min(1e-12)
        return -(labels * out) / (yn * on)


class LossWasserstein(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossWasserstein", "WASSERSTEIN"

    def _score_elems(self, labels, out):
        return labels * out / labels.shape[1]


class LossMultiLabel(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossMultiLabel", "MULTILABEL"

    def _score_elems(self, labels, out):
        # pairwise ranking loss: mean over (pos, neg) pairs of exp(out_neg - out_pos), spread over columns
        pos = labels
        neg = 1.0 - labels
        npos = pos.sum(1, keepdim=True).clamp_min(1.0)
        nneg = neg.sum(1, keepdim=True).clamp_min(1.0)
        e_neg = (torch.exp(out) * neg).sum(1, keepdim=True)
        contrib = pos * torch.exp(-out) * e_neg / (npos * nneg)
        return contrib


class LossFMeasure(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossFMeasure", "FMEASURE"
    beta = 1.0

    def _score_elems(self, labels, out):
        raise NotImplementedError

    def computeScoreArray(self, labels, preOutput, activationFn, mask=None):
        raise RuntimeError("Cannot calculate per-example score for F-measure loss")

    def _fscore(self, labels, out, mask):
        if out.shape[1] == 2:
            out, labels = out[:, 1:2], labels[:, 1:2]
        if mask is not None:
            out, labels = _apply_mask(out, mask), _apply_mask(labels, mask)
        tp = (labels * out).sum()
        fp = ((1 - labels) * out).sum() if mask is None else (_apply_mask(1 - labels, mask) * out).sum()
        fn = (labels * (1 - out)).sum()
        b2 = self.beta ** 2
        return (1 + b2) * tp / ((1 + b2) * tp + b2 * fn + fp).clamp_min(1e-12)

    def computeScore(self, labels, preOutput, activationFn, mask=None, average=True):
        out = get_activation(activationFn).forward(preOutput)
        return 1.0 - self._fscore(labels.to(out.dtype), out, mask)

    def computeGradientAndScore(self, labels, preOutput, activationFn, mask=None, average=True):
        act = get_activation(activationFn)
        z = preOutput.detach().requires_grad_(True)
        with torch.enable_grad():
            s = 1.0 - self._fscore(labels.to(z.dtype), act.forward(z, True), mask)
        return s.detach(), torch.autograd.grad(s, z)[0]


class LossMixtureDensity(ILossFunction):
    JAVA_CLASS, NAME = _PKG + "LossMixtureDensity", "MIXTURE_DENSITY"
    mixtures = 1
    labelWidth = 1

    def _nll(self, labels, out):
        k, w = self.mixtures, self.labelWidth
        alpha = torch.softmax(out[:, :k], dim=1)
        sigma = torch.exp(out[:, k:2 * k])
        mu = out[:, 2 * k:].reshape(-1, k, w)
        d2 = ((labels.reshape(-1, 1, w) - mu) ** 2).sum(2)
        logp = -0.5 * d2 / (sigma ** 2) - w * torch.log(sigma) - 0.5 * w * math.log(2 * math.pi)
        return -torch.logsumexp(torch.log(alpha.clamp_min(1e-30)) + logp, dim=1)

    def computeScoreArray(self, labels, preOutput, activationFn, mask=None):
        out = get_activation(activationFn).forward(preOutput)
        s = self._nll(labels.to(out.dtype), out)
        return s if mask is None else s * mask.reshape(-1).to(s.dtype)

    def computeGradientAndScore(self, labels, preOutput, activationFn, mask=None, average=True):
        act = get_activation(activationFn)
        z = preOutput.detach().requires_grad_(True)
        with torch.enable_grad():
            s = self._nll(labels.to(z.dtype), act.forward(z, True))
            if mask is not None:
                s = s * mask.reshape(-1).to(s.dtype)
            s = s.sum()
        g = torch.autograd.grad(s, z)[0]
        s = s.detach()
        return (s / preOutput.shape[0] if average else s), g

    def to_json(self):
        d = super().to_json()
        d.update({"mixtures": self.mixtures, "labelWidth": self.labelWidth})
        return d


_ALL = [LossMCXENT, LossNegativeLogLikelihood, LossSparseMCXENT, LossBinaryXENT, LossMSE, LossL2, LossL1, LossMAE,
        LossMAPE, LossMSLE, LossHinge, LossSquaredHinge, LossKLD, LossPoisson, LossCosineProximity, LossWasserstein,
        LossMultiLabel, LossFMeasure, LossMixtureDensity]
BY_NAME = {c.NAME: c for c in _ALL}
BY_NAME.update({"RECONSTRUCTION_CROSSENTROPY": LossBinaryXENT, "SQUARED_LOSS": LossL2, "MEAN_ABSOLUTE_ERROR": LossMAE})
BY_CLASS = {c.JAVA_CLASS: c for c in _ALL}


class LossFunctions:
    class LossFunction:
        MSE = "MSE"; L1 = "L1"; XENT = "XENT"; MCXENT = "MCXENT"; SPARSE_MCXENT = "SPARSE_MCXENT"
        SQUARED_LOSS = "SQUARED_LOSS"; RECONSTRUCTION_CROSSENTROPY = "RECONSTRUCTION_CROSSENTROPY"
        NEGATIVELOGLIKELIHOOD = "NEGATIVELOGLIKELIHOOD"; COSINE_PROXIMITY = "COSINE_PROXIMITY"; HINGE = "HINGE"
        SQUARED_HINGE = "SQUARED_HINGE"; KL_DIVERGENCE = "KL_DIVERGENCE"; MEAN_ABSOLUTE_ERROR = "MEAN_ABSOLUTE_ERROR"
        L2 = "L2"; MEAN_ABSOLUTE_PERCENTAGE_ERROR = "MEAN_ABSOLUTE_PERCENTAGE_ERROR"
        MEAN_SQUARED_LOGARITHMIC_ERROR = "MEAN_SQUARED_LOGARITHMIC_ERROR"; POISSON = "POISSON"
        WASSERSTEIN = "WASSERSTEIN"


def get_loss(l) -> ILossFunction:
    if isinstance(l, ILossFunction):
        return l
    if isinstance(l, dict):
        cls = BY_CLASS[l["@class"]]
        kw = {k: v for k, v in l.items() if k not in ("@class", "configProperties", "numerical")}
        return cls(**kw)
    return BY_NAME[str(l).upper()]()
