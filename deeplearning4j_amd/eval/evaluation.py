"""Evaluation family (reference: nd4j-api/src/main/java/org/nd4j/evaluation/** —
classification/Evaluation.java, EvaluationBinary.java, EvaluationCalibration.java, ROC.java, ROCBinary.java,
ROCMultiClass.java, regression/RegressionEvaluation.java; merge contract IEvaluation.merge).

All accumulators are small host-side tensors; per-batch statistics are computed on the device the
predictions live on (one bincount / sort per batch) and only the reduced counts cross to the host,
so evaluating on a MI355X does not serialise on per-example transfers.
"""
from __future__ import annotations

import json
import math
from typing import List, Optional

import torch


def _t(x):
    if x is None:
        return None
    if hasattr(x, "tensor"):
        return x.tensor()
    if isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(x)


def _to2d(labels, preds, mask):
    """Collapse time-series [mb, n, T] (NCW) / CNN [mb, n, H, W] outputs to [rows, n] and apply a
    per-example or per-timestep mask (reference: evaluation/BaseEvaluation.reshapeAndExtractNotMasked)."""
    labels, preds, mask = _t(labels), _t(preds), _t(mask)
    if preds.dim() == 1:
        preds = preds.unsqueeze(1)
    if labels.dim() == 1 and preds.dim() == 2 and preds.shape[1] > 1 and not labels.is_floating_point():
        labels = torch.nn.functional.one_hot(labels.long(), preds.shape[1])
    elif labels.dim() == 1:
        labels = labels.unsqueeze(1)
    if preds.dim() >= 3:
        n = preds.shape[1]
        perm = [0] + list(range(2, preds.dim())) + [1]
        preds = preds.permute(*perm).reshape(-1, n)
        labels = labels.permute(*perm).reshape(-1, labels.shape[1])
        if mask is not None:
            if mask.dim() == preds.dim() + 1 and mask.shape[1] == n:      # per-output mask
                mask = mask.permute(*perm).reshape(-1, n)
            else:
                mask = mask.reshape(-1)
    labels = labels.to(preds.device)
    if mask is not None:
        mask = mask.to(preds.device)
        if mask.dim() == 1 or (mask.dim() == 2 and mask.shape[1] == 1):
            keep = mask.reshape(-1) > 0
            labels, preds, mask = labels[keep], preds[keep], None
    return labels.float(), preds.float(), mask


class IEvaluation:
    def eval(self, labels, predictions, mask=None, recordMetaData=None):
        raise NotImplementedError

    def evalTimeSeries(self, labels, predictions, labelsMask=None):
        return self.eval(labels, predictions, labelsMask)

    def merge(self, other):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def stats(self):
        raise NotImplementedError

    def toJson(self):
        d = {"@class": "org.nd4j.evaluation." + self._JAVA}
        for k, v in self.__dict__.items():
            if isinstance(v, torch.Tensor):
                d[k] = v.tolist()
            elif isinstance(v, (int, float, str, bool, list, type(None))):
                d[k] = v
        return json.dumps(d)

    def __str__(self):
        return self.stats()


class ConfusionMatrix:
    def __init__(self, n):
        self.m = torch.zeros(n, n, dtype=torch.int64)

    def getCount(self, actual, predicted):
        return int(self.m[actual, predicted])

    def getActualTotal(self, c):
        return int(self.m[c].sum())

    def getPredictedTotal(self, c):
        return int(self.m[:, c].sum())

    def toCSV(self):
        return "\n".join(",".join(str(int(v)) for v in row) for row in self.m)


class Evaluation(IEvaluation):
    """Multi-class classification metrics (classification/Evaluation.java: eval :341-520, precision/recall/
    f1 macro-averaged excluding classes with no data :700-1000, topN :495)."""
    _JAVA = "classification.Evaluation"

    def __init__(self, numClasses=None, labels: Optional[List[str]] = None, topN=1, binaryDecisionThreshold=None,
                 costArray=None):
        if isinstance(numClasses, (list, tuple)):
            labels, numClasses = list(numClasses), None
        self.labelsList = list(labels) if labels else None
        self.numClasses = numClasses if numClasses is not None else (len(labels) if labels else None)
        self.topN = int(topN)
        self.binaryDecisionThreshold = binaryDecisionThreshold
        self.costArray = _t(costArray)
        self.reset()

    def reset(self):
        self.confusion = ConfusionMatrix(self.numClasses) if self.numClasses else None
        self.topNCorrectCount = 0
        self.topNTotalCount = 0
        self.numRowCounter = 0

    def _ensure(self, n):
        if self.confusion is None:
            self.numClasses = n
            self.confusion = ConfusionMatrix(n)

    def eval(self, labels, predictions, mask=None, recordMetaData=None):
        y, p, _ = _to2d(labels, predictions, mask)
        if y.shape[0] == 0:
            return
        n = p.shape[1]
        if n == 1:                                  # single sigmoid output: binary
            thr = 0.5 if self.binaryDecisionThreshold is None else self.binaryDecisionThreshold
            self._ensure(2)
            actual = (y[:, 0] > 0.5).long()
            pred = (p[:, 0] > thr).long()
        else:
            self._ensure(n)
            actual = y.argmax(1)
            if self.binaryDecisionThreshold is not None and n == 2:
                pred = (p[:, 1] >= self.binaryDecisionThreshold).long()
            elif self.costArray is not None:
                pred = (p * self.costArray.to(p.device).reshape(1, -1)).argmax(1)
            else:
                pred = p.argmax(1)
        nc = self.numClasses
        counts = torch.bincount(actual * nc + pred, minlength=nc * nc).reshape(nc, nc)
        self.confusion.m += counts.cpu()
        self.numRowCounter += int(y.shape[0])
        if self.topN > 1 and n > 1:
            topk = p.topk(min(self.topN, n), dim=1).indices
            self.topNCorrectCount += int((topk == actual.unsqueeze(1)).any(1).sum())
            self.topNTotalCount += int(y.shape[0])

    # --- counts
    def truePositives(self, c=None):
        d = self.confusion.m.diag()
        return int(d[c]) if c is not None else {i: int(v) for i, v in enumerate(d)}

    def falsePositives(self, c=None):
        v = self.confusion.m.sum(0) - self.confusion.m.diag()
        return int(v[c]) if c is not None else {i: int(x) for i, x in enumerate(v)}

    def falseNegatives(self, c=None):
        v = self.confusion.m.sum(1) - self.confusion.m.diag()
        return int(v[c]) if c is not None else {i: int(x) for i, x in enumerate(v)}

    def trueNegatives(self, c=None):
        m = self.confusion.m
        v = m.sum() - m.sum(0) - m.sum(1) + m.diag()
        return int(v[c]) if c is not None else {i: int(x) for i, x in enumerate(v)}

    def classCount(self, c):
        return self.confusion.getActualTotal(c)

    def getNumRowCounter(self):
        return self.numRowCounter

    # --- metrics
    def accuracy(self):
        tot = self.numRowCounter
        return float(self.confusion.m.diag().sum()) / tot if tot else 0.0

    def topNAccuracy(self):
        if self.topN <= 1:
            return self.accuracy()
        return self.topNCorrectCount / self.topNTotalCount if self.topNTotalCount else 0.0

    def _prf(self, c, which, edge=0.0):
        tp, fp, fn = self.truePositives(c), self.falsePositives(c), self.falseNegatives(c)
        if which == "p":
            return tp / (tp + fp) if tp + fp else edge
        if which == "r":
            return tp / (tp + fn) if tp + fn else edge
        p = tp / (tp + fp) if tp + fp else 0.0
        r = tp / (tp + fn) if tp + fn else 0.0
        return 2 * p * r / (p + r) if p + r else edge

    def _macro(self, which):
        if self.numClasses == 2 and which in "prf":
            return self._prf(1, which)               # binary case reports the positive class (Evaluation.java:713)
        vals = []
        for c in range(self.numClasses):
            tp, fp, fn = self.truePositives(c), self.falsePositives(c), self.falseNegatives(c)
            if which == "p" and tp + fp == 0:
                continue
            if which == "r" and tp + fn == 0:
                continue
            if which == "f" and (tp + fp == 0 or tp + fn == 0):
                continue
            vals.append(self._prf(c, which))
        return sum(vals) / len(vals) if vals else 0.0

    def precision(self, c=None, averaging="macro"):
        if c is not None:
            return self._prf(c, "p")
        if averaging == "micro":
            tp = self.confusion.m.diag().sum().item()
            fp = sum(self.falsePositives().values())
            return tp / (tp + fp) if tp + fp else 0.0
        return self._macro("p")

    def recall(self, c=None, averaging="macro"):
        if c is not None:
            return self._prf(c, "r")
        if averaging == "micro":
            tp = self.confusion.m.diag().sum().item()
            fn = sum(self.falseNegatives().values())
            return tp / (tp + fn) if tp + fn else 0.0
        return self._macro("r")

    def f1(self, c=None, averaging="macro"):
        if c is not None:
            return self._prf(c, "f")
        if averaging == "micro":
            p, r = self.precision(averaging="micro"), self.recall(averaging="micro")
            return 2 * p * r / (p + r) if p + r else 0.0
        return self._macro("f")

    def fBeta(self, beta, c=None):
        p, r = self.precision(c), self.recall(c)
        d = beta * beta * p + r
        return (1 + beta * beta) * p * r / d if d else 0.0

    def gMeasure(self, c=None):
        return math.sqrt(self.precision(c) * self.recall(c))

    def falsePositiveRate(self, c=None):
        if c is None:
            if self.numClasses == 2:
                return self.falsePositiveRate(1)
            return sum(self.falsePositiveRate(i) for i in range(self.numClasses)) / self.numClasses
        fp, tn = self.falsePositives(c), self.trueNegatives(c)
        return fp / (fp + tn) if fp + tn else 0.0

    def falseNegativeRate(self, c=None):
        if c is None:
            if self.numClasses == 2:
                return self.falseNegativeRate(1)
            return sum(self.falseNegativeRate(i) for i in range(self.numClasses)) / self.numClasses
        fn, tp = self.falseNegatives(c), self.truePositives(c)
        return fn / (fn + tp) if fn + tp else 0.0

    def falseAlarmRate(self):
        return (self.falsePositiveRate() + self.falseNegativeRate()) / 2.0

    def matthewsCorrelation(self, c=None):
        if c is None:
            if self.numClasses == 2:
                return self.matthewsCorrelation(1)
            return sum(self.matthewsCorrelation(i) for i in range(self.numClasses)) / self.numClasses
        tp, fp, fn, tn = self.truePositives(c), self.falsePositives(c), self.falseNegatives(c), self.trueNegatives(c)
        d = math.sqrt(float(tp + fp) * (tp + fn) * (tn + fp) * (tn + fn))
        return (tp * tn - fp * fn) / d if d else 0.0

    def getConfusionMatrix(self):
        return self.confusion

    def confusionMatrix(self):
        return self.confusionToString()

    def confusionToString(self):
        n = self.numClasses
        names = self.labelsList or [str(i) for i in range(n)]
        w = max(5, max(len(s) for s in names) + 1)
        lines = [" " * w + "".join(("%" + str(w) + "s") % i for i in range(n))]
        for i in range(n):
            lines.append(("%-" + str(w) + "s") % names[i] + "".join(("%" + str(w) + "d") % int(v) for v in self.confusion.m[i]))
        return "\n".join(lines)

    def merge(self, other: "Evaluation"):
        if other is None or other.confusion is None:
            return self
        if self.confusion is None:
            self._ensure(other.numClasses)
        self.confusion.m += other.confusion.m
        self.numRowCounter += other.numRowCounter
        self.topNCorrectCount += other.topNCorrectCount
        self.topNTotalCount += other.topNTotalCount
        return self

    def stats(self, suppressWarnings=False, includeConfusion=True):
        if self.confusion is None:
            return "Evaluation: no data"
        s = ["", "========================Evaluation Metrics========================",
             " # of classes:    %d" % self.numClasses,
             " Accuracy:        %.4f" % self.accuracy()]
        if self.topN > 1:
            s.append(" Top %d Accuracy:  %.4f" % (self.topN, self.topNAccuracy()))
        s += [" Precision:       %.4f" % self.precision(), " Recall:          %.4f" % self.recall(),
              " F1 Score:        %.4f" % self.f1()]
        if self.numClasses > 2:
            s.append("Precision, recall & F1: macro-averaged (equally weighted avg. of %d classes)" % self.numClasses)
        else:
            s.append("Precision, recall & F1: reported for positive class (class 1) only")
        if includeConfusion and self.numClasses <= 20:
            s += ["", "=========================Confusion Matrix=========================", self.confusionToString(),
                  "", "Confusion matrix format: Actual (rowClass) predicted as (columnClass) N times"]
        s.append("==================================================================")
        return "\n".join(s)


class EvaluationBinary(IEvaluation):
    """Independent binary metrics per output column (classification/EvaluationBinary.java)."""
    _JAVA = "classification.EvaluationBinary"

    def __init__(self, size=None, decisionThreshold=None, rocBinarySteps=None):
        self.decisionThreshold = _t(decisionThreshold)
        self.rocBinary = ROCBinary(rocBinarySteps) if rocBinarySteps is not None else None
        self.labelsList = None
        self.n = size
        self.reset()

    def reset(self):
        z = lambda: torch.zeros(self.n, dtype=torch.int64) if self.n else None
        self.countTP, self.countFP, self.countTN, self.countFN = z(), z(), z(), z()

    def eval(self, labels, predictions, mask=None, recordMetaData=None):
        y, p, m = _to2d(labels, predictions, mask)
        if self.countTP is None:
            self.n = p.shape[1]
            self.reset()
        thr = 0.5 if self.decisionThreshold is None else self.decisionThreshold.to(p.device).reshape(1, -1)
        pred = p > thr
        act = y > 0.5
        w = torch.ones_like(p, dtype=torch.bool) if m is None else (m > 0)
        self.countTP += (pred & act & w).sum(0).cpu()
        self.countFP += (pred & ~act & w).sum(0).cpu()
        self.countTN += (~pred & ~act & w).sum(0).cpu()
        self.countFN += (~pred & act & w).sum(0).cpu()
        if self.rocBinary is not None:
            self.rocBinary.eval(labels, predictions, mask)

    def numLabels(self):
        return self.n

    def totalCount(self, i):
        return int(self.countTP[i] + self.countFP[i] + self.countTN[i] + self.countFN[i])

    def truePositives(self, i):
        return int(self.countTP[i])

    def falsePositives(self, i):
        return int(self.countFP[i])

    def trueNegatives(self, i):
        return int(self.countTN[i])

    def falseNegatives(self, i):
        return int(self.countFN[i])

    def accuracy(self, i):
        t = self.totalCount(i)
        return (self.truePositives(i) + self.trueNegatives(i)) / t if t else 0.0

    def precision(self, i):
        d = self.truePositives(i) + self.falsePositives(i)
        return self.truePositives(i) / d if d else 0.0

    def recall(self, i):
        d = self.truePositives(i) + self.falseNegatives(i)
        return self.truePositives(i) / d if d else 0.0

    def f1(self, i):
        p, r = self.precision(i), self.recall(i)
        return 2 * p * r / (p + r) if p + r else 0.0

    def averageAccuracy(self):
        return sum(self.accuracy(i) for i in range(self.n)) / self.n

    def averagePrecision(self):
        return sum(self.precision(i) for i in range(self.n)) / self.n

    def averageRecall(self):
        return sum(self.recall(i) for i in range(self.n)) / self.n

    def averageF1(self):
        return sum(self.f1(i) for i in range(self.n)) / self.n

    def merge(self, o):
        if o.countTP is None:
            return self
        if self.countTP is None:
            self.n = o.n
            self.reset()
        for k in ("countTP", "countFP", "countTN", "countFN"):
            setattr(self, k, getattr(self, k) + getattr(o, k))
        if self.rocBinary is not None and o.rocBinary is not None:
            self.rocBinary.merge(o.rocBinary)
        return self

    def stats(self):
        rows = ["Label  Acc     Prec    Rec     F1      TP    FP    TN    FN"]
        for i in range(self.n or 0):
            rows.append("%-6d %.4f  %.4f  %.4f  %.4f  %-5d %-5d %-5d %-5d" % (
                i, self.accuracy(i), self.precision(i), self.recall(i), self.f1(i), self.truePositives(i),
                self.falsePositives(i), self.trueNegatives(i), self.falseNegatives(i)))
        return "\n".join(rows)


class RegressionEvaluation(IEvaluation):
    """Column-wise regression metrics (regression/RegressionEvaluation.java: running sums so that
    merge() is exact)."""
    _JAVA = "regression.RegressionEvaluation"

    def __init__(self, nColumns=None, columnNames=None, precision=5):
        if isinstance(nColumns, (list, tuple)):
            columnNames, nColumns = list(nColumns), len(nColumns)
        self.columnNames = columnNames
        self.n = nColumns
        self.precision = precision
        self.reset()

    def reset(self):
        z = lambda: torch.zeros(self.n, dtype=torch.float64) if self.n else None
        self.exampleCount = z()
        self.labelsSum, self.predSum = z(), z()
        self.sumSqErr, self.sumAbsErr = z(), z()
        self.sumSqLabels, self.sumSqPred, self.sumLabelPred = z(), z(), z()

    def eval(self, labels, predictions, mask=None, recordMetaData=None):
        y, p, m = _to2d(labels, predictions, mask)
        if self.exampleCount is None:
            self.n = p.shape[1]
            self.reset()
        y, p = y.double(), p.double()
        w = torch.ones_like(p) if m is None else (m > 0).double()
        y, p = y * w, p * w
        e = p - y
        acc = lambda name, v: setattr(self, name, getattr(self, name) + v.cpu())
        acc("exampleCount", w.sum(0))
        acc("labelsSum", y.sum(0))
        acc("predSum", p.sum(0))
        acc("sumSqErr", (e * e).sum(0))
        acc("sumAbsErr", e.abs().sum(0))
        acc("sumSqLabels", (y * y).sum(0))
        acc("sumSqPred", (p * p).sum(0))
        acc("sumLabelPred", (y * p).sum(0))

    def numColumns(self):
        return self.n

    def meanSquaredError(self, c):
        return float(self.sumSqErr[c] / self.exampleCount[c])

    def meanAbsoluteError(self, c):
        return float(self.sumAbsErr[c] / self.exampleCount[c])

    def rootMeanSquaredError(self, c):
        return math.sqrt(self.meanSquaredError(c))

    def pearsonCorrelation(self, c):
        n = self.exampleCount[c]
        num = n * self.sumLabelPred[c] - self.labelsSum[c] * self.predSum[c]
        den = torch.sqrt(n * self.sumSqLabels[c] - self.labelsSum[c] ** 2) * torch.sqrt(n * self.sumSqPred[c] - self.predSum[c] ** 2)
        return float(num / den) if float(den) != 0 else 0.0

    correlationR2 = pearsonCorrelation

    def rSquared(self, c):
        n = self.exampleCount[c]
        sstot = self.sumSqLabels[c] - self.labelsSum[c] ** 2 / n
        return float(1.0 - self.sumSqErr[c] / sstot) if float(sstot) != 0 else 0.0

    def relativeSquaredError(self, c):
        n = self.exampleCount[c]
        sstot = self.sumSqLabels[c] - self.labelsSum[c] ** 2 / n
        return float(self.sumSqErr[c] / sstot) if float(sstot) != 0 else float("inf")

    def _avg(self, f):
        return sum(f(c) for c in range(self.n)) / self.n

    def averageMeanSquaredError(self):
        return self._avg(self.meanSquaredError)

    def averageMeanAbsoluteError(self):
        return self._avg(self.meanAbsoluteError)

    def averagerootMeanSquaredError(self):
        return self._avg(self.rootMeanSquaredError)

    averageRootMeanSquaredError = averagerootMeanSquaredError

    def averagePearsonCorrelation(self):
        return self._avg(self.pearsonCorrelation)

    def averageRSquared(self):
        return self._avg(self.rSquared)

    def averagerelativeSquaredError(self):
        return self._avg(self.relativeSquaredError)

    def merge(self, o):
        if o.exampleCount is None:
            return self
        if self.exampleCount is None:
            self.n = o.n
            self.reset()
        for k in ("exampleCount", "labelsSum", "predSum", "sumSqErr", "sumAbsErr", "sumSqLabels", "sumSqPred", "sumLabelPred"):
            setattr(self, k, getattr(self, k) + getattr(o, k))
        return self

    def stats(self):
        if self.exampleCount is None:
            return "RegressionEvaluation: no data"
        names = self.columnNames or ["col_%d" % i for i in range(self.n)]
        f = "%." + str(self.precision) + "e"
        rows = ["%-12s %-14s %-14s %-14s %-14s %-14s %-14s" % ("Column", "MSE", "MAE", "RMSE", "RSE", "PC", "R^2")]
        for c in range(self.n):
            rows.append("%-12s " % names[c] + " ".join(("%-14s" % (f % v)) for v in (
                self.meanSquaredError(c), self.meanAbsoluteError(c), self.rootMeanSquaredError(c),
                self.relativeSquaredError(c), self.pearsonCorrelation(c), self.rSquared(c))))
        return "\n".join(rows)


class RocCurve:
    def __init__(self, thresholds, fpr, tpr):
        self.thresholds, self.fpr, self.tpr = thresholds, fpr, tpr

    def calculateAUC(self):
        return float(torch.trapz(self.tpr.flip(0), self.fpr.flip(0))) if self.fpr[0] > self.fpr[-1] else float(torch.trapz(self.tpr, self.fpr))


class PrecisionRecallCurve:
    def __init__(self, thresholds, precision, recall):
        self.thresholds, self.precision, self.recall = thresholds, precision, recall

    def calculateAUPRC(self):
        r, p = self.recall, self.precision
        if r[0] > r[-1]:
            r, p = r.flip(0), p.flip(0)
        return float(torch.trapz(p, r))


class ROC(IEvaluation):
    """Binary ROC / PR curves (classification/ROC.java). thresholdSteps == 0 → exact mode (stores all
    probabilities, sorts once at query time); > 0 → fixed histogram of thresholds so memory is O(steps)."""
    _JAVA = "classification.ROC"

    def __init__(self, thresholdSteps=0, rocRemoveRedundantPts=True):
        self.thresholdSteps = int(thresholdSteps)
        self.rocRemoveRedundantPts = rocRemoveRedundantPts
        self.reset()

    def reset(self):
        self.isExact = self.thresholdSteps == 0
        self.countActualPositive = 0
        self.countActualNegative = 0
        self._probs: List[torch.Tensor] = []
        self._labels: List[torch.Tensor] = []
        if not self.isExact:
            self._tp = torch.zeros(self.thresholdSteps + 1, dtype=torch.int64)
            self._fp = torch.zeros(self.thresholdSteps + 1, dtype=torch.int64)
        self._auc = self._auprc = None

    def eval(self, labels, predictions, mask=None, recordMetaData=None):
        y, p, _ = _to2d(labels, predictions, mask)
        if p.shape[1] == 2:
            y, p = y[:, 1], p[:, 1]
        elif p.shape[1] == 1:
            y, p = y[:, 0], p[:, 0]
        else:
            raise ValueError("ROC is for binary problems (1 or 2 output columns); use ROCMultiClass / ROCBinary")
        self._eval1d(y, p)

    def _eval1d(self, y, p):
        pos = y > 0.5
        self.countActualPositive += int(pos.sum())
        self.countActualNegative += int((~pos).sum())
        self._auc = self._auprc = None
        if self.isExact:
            self._probs.append(p.detach().cpu().double())
            self._labels.append(pos.cpu())
        else:
            steps = self.thresholdSteps
            b = torch.clamp((p * steps).floor().long(), 0, steps)          # p >= t_i  for i <= b
            hp = torch.bincount(b[pos], minlength=steps + 1)
            hn = torch.bincount(b[~pos], minlength=steps + 1)
            self._tp += hp.flip(0).cumsum(0).flip(0).cpu()
            self._fp += hn.flip(0).cumsum(0).flip(0).cpu()

    def _exact_curves(self):
        p = torch.cat(self._probs) if self._probs else torch.zeros(0, dtype=torch.float64)
        y = torch.cat(self._labels) if self._labels else torch.zeros(0, dtype=torch.bool)
        order = torch.argsort(p, descending=True, stable=True)
        p, y = p[order], y[order]
        tp = torch.cumsum(y.long(), 0)
        fp = torch.cumsum((~y).long(), 0)
        # collapse ties: keep the last entry for each distinct probability
        if p.numel():
            last = torch.ones_like(p, dtype=torch.bool)
            last[:-1] = p[:-1] != p[1:]
            p, tp, fp = p[last], tp[last], fp[last]
        thr = torch.cat([torch.ones(1, dtype=torch.float64), p])
        tp = torch.cat([torch.zeros(1, dtype=torch.int64), tp])
        fp = torch.cat([torch.zeros(1, dtype=torch.int64), fp])
        return thr, tp.double(), fp.double()

    def getRocCurve(self):
        P, N = max(self.countActualPositive, 1), max(self.countActualNegative, 1)
        if self.isExact:
            thr, tp, fp = self._exact_curves()
        else:
            thr = torch.arange(self.thresholdSteps + 1, dtype=torch.float64) / self.thresholdSteps
            tp, fp = self._tp.double(), self._fp.double()
        return RocCurve(thr, fp / N, tp / P)

    def getPrecisionRecallCurve(self):
        P = max(self.countActualPositive, 1)
        if self.isExact:
            thr, tp, fp = self._exact_curves()
        else:
            thr = torch.arange(self.thresholdSteps + 1, dtype=torch.float64) / self.thresholdSteps
            tp, fp = self._tp.double(), self._fp.double()
        denom = tp + fp
        prec = torch.where(denom > 0, tp / denom.clamp_min(1), torch.ones_like(tp))
        return PrecisionRecallCurve(thr, prec, tp / P)

    def calculateAUC(self):
        if self._auc is None:
            self._auc = self.getRocCurve().calculateAUC()
        return self._auc

    def calculateAUCPR(self):
        if self._auprc is None:
            self._auprc = self.getPrecisionRecallCurve().calculateAUPRC()
        return self._auprc

    def merge(self, o):
        self.countActualPositive += o.countActualPositive
        self.countActualNegative += o.countActualNegative
        if self.isExact:
            self._probs += o._probs
            self._labels += o._labels
        else:
            self._tp += o._tp
            self._fp += o._fp
        self._auc = self._auprc = None
        return self

    def stats(self):
        return "AUC (Area under ROC Curve):                %.6f\nAUPRC (Area under Precision/Recall Curve): %.6f" % (
            self.calculateAUC(), self.calculateAUCPR())


class _MultiROC(IEvaluation):
    def __init__(self, thresholdSteps=0, rocRemoveRedundantPts=True):
        self.thresholdSteps = int(thresholdSteps) if thresholdSteps is not None else 0
        self.rocRemoveRedundantPts = rocRemoveRedundantPts
        self.underlying: Optional[List[ROC]] = None

    def reset(self):
        self.underlying = None

    def _ensure(self, n):
        if self.underlying is None:
            self.underlying = [ROC(self.thresholdSteps, self.rocRemoveRedundantPts) for _ in range(n)]

    def getNumClasses(self):
        return len(self.underlying) if self.underlying else 0

    numLabels = getNumClasses

    def calculateAUC(self, c):
        return self.underlying[c].calculateAUC()

    def calculateAUCPR(self, c):
        return self.underlying[c].calculateAUCPR()

    def calculateAverageAUC(self):
        return sum(r.calculateAUC() for r in self.underlying) / len(self.underlying)

    def calculateAverageAUCPR(self):
        return sum(r.calculateAUCPR() for r in self.underlying) / len(self.underlying)

    calculateAverageAuc = calculateAverageAUC

    def getRocCurve(self, c):
        return self.underlying[c].getRocCurve()

    def getPrecisionRecallCurve(self, c):
        return self.underlying[c].getPrecisionRecallCurve()

    def merge(self, o):
        if o.underlying is None:
            return self
        self._ensure(len(o.underlying))
        for a, b in zip(self.underlying, o.underlying):
            a.merge(b)
        return self

    def stats(self):
        if not self.underlying:
            return "no data"
        rows = ["Label  AUC        AUPRC"]
        for i, r in enumerate(self.underlying):
            rows.append("%-6d %.6f   %.6f" % (i, r.calculateAUC(), r.calculateAUCPR()))
        rows.append("Average AUC: %.6f" % self.calculateAverageAUC())
        return "\n".join(rows)


class ROCMultiClass(_MultiROC):
    """One-vs-all ROC per class (classification/ROCMultiClass.java)."""
    _JAVA = "classification.ROCMultiClass"

    def eval(self, labels, predictions, mask=None, recordMetaData=None):
        y, p, _ = _to2d(labels, predictions, mask)
        self._ensure(p.shape[1])
        for c, r in enumerate(self.underlying):
            r._eval1d(y[:, c], p[:, c])


class ROCBinary(_MultiROC):
    """Independent binary ROC per output with optional per-output masks (classification/ROCBinary.java)."""
    _JAVA = "classification.ROCBinary"

    def eval(self, labels, predictions, mask=None, recordMetaData=None):
        y, p, m = _to2d(labels, predictions, mask)
        self._ensure(p.shape[1])
        for c, r in enumerate(self.underlying):
            if m is not None:
                k = m[:, c] > 0
                r._eval1d(y[k, c], p[k, c])
            else:
                r._eval1d(y[:, c], p[:, c])


class EvaluationCalibration(IEvaluation):
    """Reliability diagram, residual plot and probability histograms
    (classification/EvaluationCalibration.java)."""
    _JAVA = "classification.EvaluationCalibration"

    def __init__(self, reliabilityDiagNumBins=10, histogramNumBins=50, excludeEmptyBins=True):
        self.reliabilityDiagNumBins = reliabilityDiagNumBins
        self.histogramNumBins = histogramNumBins
        self.excludeEmptyBins = excludeEmptyBins
        self.n = None

    def reset(self):
        self.n = None

    def _init(self, n):
        self.n = n
        R, H = self.reliabilityDiagNumBins, self.histogramNumBins
        self.rDiagBinPosCount = torch.zeros(R, n, dtype=torch.int64)
        self.rDiagBinTotalCount = torch.zeros(R, n, dtype=torch.int64)
        self.rDiagBinSumPredictions = torch.zeros(R, n, dtype=torch.float64)
        self.labelCountsEachClass = torch.zeros(n, dtype=torch.int64)
        self.predictionCountsEachClass = torch.zeros(n, dtype=torch.int64)
        self.residualPlotOverall = torch.zeros(H, dtype=torch.int64)
        self.residualPlotByLabelClass = torch.zeros(H, n, dtype=torch.int64)
        self.probHistogramOverall = torch.zeros(H, dtype=torch.int64)
        self.probHistogramByLabelClass = torch.zeros(H, n, dtype=torch.int64)

    def eval(self, labels, predictions, mask=None, recordMetaData=None):
        y, p, m = _to2d(labels, predictions, mask)
        y, p = y.cpu(), p.cpu().double()
        if self.n is None:
            self._init(p.shape[1])
        n, R, H = self.n, self.reliabilityDiagNumBins, self.histogramNumBins
        pos = y > 0.5
        rb = torch.clamp((p * R).floor().long(), 0, R - 1)
        col = torch.arange(n).expand_as(rb)
        flat = (rb * n + col).reshape(-1)
        self.rDiagBinTotalCount += torch.bincount(flat, minlength=R * n).reshape(R, n)
        self.rDiagBinPosCount += torch.bincount(flat[pos.reshape(-1)], minlength=R * n).reshape(R, n)
        self.rDiagBinSumPredictions += torch.zeros(R * n, dtype=torch.float64).index_add_(0, flat, p.reshape(-1)).reshape(R, n)
        self.labelCountsEachClass += pos.sum(0)
        self.predictionCountsEachClass += torch.bincount(p.argmax(1), minlength=n)
        resid = (y.double() - p).abs()
        hb = torch.clamp((resid * H).floor().long(), 0, H - 1)
        self.residualPlotOverall += torch.bincount(hb.reshape(-1), minlength=H)
        pb = torch.clamp((p * H).floor().long(), 0, H - 1)
        self.probHistogramOverall += torch.bincount(pb.reshape(-1), minlength=H)
        for c in range(n):
            rows = pos[:, c]
            self.residualPlotByLabelClass[:, c] += torch.bincount(hb[rows, c], minlength=H)
            self.probHistogramByLabelClass[:, c] += torch.bincount(pb[rows, c], minlength=H)

    def getReliabilityDiagram(self, c):
        tot = self.rDiagBinTotalCount[:, c].double()
        keep = tot > 0 if self.excludeEmptyBins else torch.ones_like(tot, dtype=torch.bool)
        meanPred = self.rDiagBinSumPredictions[:, c] / tot.clamp_min(1)
        fracPos = self.rDiagBinPosCount[:, c].double() / tot.clamp_min(1)
        return meanPred[keep], fracPos[keep]

    def expectedCalibrationError(self, c):
        tot = self.rDiagBinTotalCount[:, c].double()
        meanPred = self.rDiagBinSumPredictions[:, c] / tot.clamp_min(1)
        fracPos = self.rDiagBinPosCount[:, c].double() / tot.clamp_min(1)
        return float(((meanPred - fracPos).abs() * tot).sum() / tot.sum().clamp_min(1))

    def getResidualPlotAllClasses(self):
        return self.residualPlotOverall

    def getResidualPlot(self, c):
        return self.residualPlotByLabelClass[:, c]

    def getProbabilityHistogramAllClasses(self):
        return self.probHistogramOverall

    def getProbabilityHistogram(self, c):
        return self.probHistogramByLabelClass[:, c]

    def merge(self, o):
        if o.n is None:
            return self
        if self.n is None:
            self._init(o.n)
        for k in ("rDiagBinPosCount", "rDiagBinTotalCount", "rDiagBinSumPredictions", "labelCountsEachClass",
                  "predictionCountsEachClass", "residualPlotOverall", "residualPlotByLabelClass",
                  "probHistogramOverall", "probHistogramByLabelClass"):
            setattr(self, k, getattr(self, k) + getattr(o, k))
        return self

    def stats(self):
        if self.n is None:
            return "EvaluationCalibration: no data"
        return "EvaluationCalibration(nClasses=%d, ECE=%s)" % (
            self.n, ["%.4f" % self.expectedCalibrationError(c) for c in range(self.n)])
