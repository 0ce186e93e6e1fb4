"""DataSet / MultiDataSet (reference: nd4j-api/.../linalg/dataset/{DataSet,MultiDataSet}.java)."""
from __future__ import annotations

import io
from typing import List, Optional

import numpy as np
import torch


def _t(x):
    if x is None:
        return None
    if hasattr(x, "tensor"):
        return x.tensor()
    if isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(np.asarray(x))


class DataSet:
    def __init__(self, features=None, labels=None, featuresMask=None, labelsMask=None):
        self.features, self.labels = _t(features), _t(labels)
        self.featuresMask, self.labelsMask = _t(featuresMask), _t(labelsMask)
        self.exampleMetaData = None

    def getFeatures(self):
        return self.features

    def getLabels(self):
        return self.labels

    def getFeaturesMaskArray(self):
        return self.featuresMask

    def getLabelsMaskArray(self):
        return self.labelsMask

    def setFeatures(self, f):
        self.features = _t(f)

    def setLabels(self, l):
        self.labels = _t(l)

    def setFeaturesMaskArray(self, m):
        self.featuresMask = _t(m)

    def setLabelsMaskArray(self, m):
        self.labelsMask = _t(m)

    def hasMaskArrays(self):
        return self.featuresMask is not None or self.labelsMask is not None

    def numExamples(self):
        return 0 if self.features is None else int(self.features.shape[0])

    def numInputs(self):
        return int(self.features[0].numel())

    def numOutcomes(self):
        return int(self.labels.shape[1])

    def copy(self):
        c = lambda t: None if t is None else t.clone()
        return DataSet(c(self.features), c(self.labels), c(self.featuresMask), c(self.labelsMask))

    def to(self, device, non_blocking=True):
        mv = lambda t: None if t is None else t.to(device, non_blocking=non_blocking)
        return DataSet(mv(self.features), mv(self.labels), mv(self.featuresMask), mv(self.labelsMask))

    def pin_memory(self):
        pm = lambda t: None if t is None or t.is_cuda else t.pin_memory()
        return DataSet(pm(self.features), pm(self.labels), pm(self.featuresMask), pm(self.labelsMask))

    def shuffle(self, seed=None):
        g = torch.Generator()
        if seed is not None:
            g.manual_seed(int(seed))
        p = torch.randperm(self.numExamples(), generator=g)
        for k in ("features", "labels", "featuresMask", "labelsMask"):
            t = getattr(self, k)
            if t is not None:
                setattr(self, k, t[p.to(t.device)])

    def get(self, idx):
        if isinstance(idx, int):
            idx = [idx]
        ix = torch.as_tensor(idx)
        s = lambda t: None if t is None else t[ix.to(t.device)]
        return DataSet(s(self.features), s(self.labels), s(self.featuresMask), s(self.labelsMask))

    def getRange(self, frm, to):
        s = lambda t: None if t is None else t[frm:to]
        return DataSet(s(self.features), s(self.labels), s(self.featuresMask), s(self.labelsMask))

    def asList(self):
        return [self.get(i) for i in range(self.numExamples())]

    def batchBy(self, n):
        return [self.getRange(i, min(i + n, self.numExamples())) for i in range(0, self.numExamples(), n)]

    def splitTestAndTrain(self, numOrFraction):
        n = self.numExamples()
        k = int(numOrFraction * n) if isinstance(numOrFraction, float) and numOrFraction < 1 else int(numOrFraction)
        return SplitTestAndTrain(self.getRange(0, k), self.getRange(k, n))

    def sample(self, n, seed=None, withReplacement=False):
        g = torch.Generator()
        if seed is not None:
            g.manual_seed(seed)
        idx = torch.randint(0, self.numExamples(), (n,), generator=g) if withReplacement else torch.randperm(self.numExamples(), generator=g)[:n]
        return self.get(idx.tolist())

    def normalizeZeroMeanZeroUnitVariance(self):
        f = self.features
        self.features = (f - f.mean(0, keepdim=True)) / f.std(0, keepdim=True, unbiased=True).clamp_min(1e-12)

    def scaleMinAndMax(self, mn, mx):
        f = self.features
        lo, hi = f.amin(0, keepdim=True), f.amax(0, keepdim=True)
        self.features = (f - lo) / (hi - lo).clamp_min(1e-12) * (mx - mn) + mn

    def save(self, path):
        torch.save({k: getattr(self, k) for k in ("features", "labels", "featuresMask", "labelsMask")}, path)

    @staticmethod
    def load(path):
        d = torch.load(path, weights_only=True)
        return DataSet(d["features"], d["labels"], d["featuresMask"], d["labelsMask"])

    @staticmethod
    def merge(datasets: List["DataSet"]):
        cat = lambda k: None if getattr(datasets[0], k) is None else torch.cat([getattr(d, k) for d in datasets], 0)
        return DataSet(cat("features"), cat("labels"), cat("featuresMask"), cat("labelsMask"))

    def __repr__(self):
        sh = lambda t: None if t is None else tuple(t.shape)
        return "DataSet(features=%s, labels=%s)" % (sh(self.features), sh(self.labels))


class SplitTestAndTrain:
    def __init__(self, train, test):
        self.train, self.test = train, test

    def getTrain(self):
        return self.train

    def getTest(self):
        return self.test


class MultiDataSet:
    def __init__(self, features=None, labels=None, featuresMasks=None, labelsMasks=None):
        L = lambda x: None if x is None else [_t(v) for v in (x if isinstance(x, (list, tuple)) else [x])]
        self.features, self.labels = L(features), L(labels)
        self.featuresMasks, self.labelsMasks = L(featuresMasks), L(labelsMasks)

    def getFeatures(self, i=None):
        return self.features if i is None else self.features[i]

    def getLabels(self, i=None):
        return self.labels if i is None else self.labels[i]

    def getFeaturesMaskArrays(self):
        return self.featuresMasks

    def getLabelsMaskArrays(self):
        return self.labelsMasks

    def getFeaturesMaskArray(self, i=0):
        return None if self.featuresMasks is None else self.featuresMasks[i]

    def getLabelsMaskArray(self, i=0):
        return None if self.labelsMasks is None else self.labelsMasks[i]

    def numFeatureArrays(self):
        return len(self.features)

    def numLabelsArrays(self):
        return len(self.labels)

    def numExamples(self):
        return int(self.features[0].shape[0])

    def to(self, device, non_blocking=True):
        mv = lambda l: None if l is None else [None if t is None else t.to(device, non_blocking=non_blocking) for t in l]
        return MultiDataSet(mv(self.features), mv(self.labels), mv(self.featuresMasks), mv(self.labelsMasks))

    def pin_memory(self):
        pm = lambda l: None if l is None else [None if (t is None or t.is_cuda) else t.pin_memory() for t in l]
        return MultiDataSet(pm(self.features), pm(self.labels), pm(self.featuresMasks), pm(self.labelsMasks))

    @staticmethod
    def merge(sets):
        cat = lambda k: None if getattr(sets[0], k) is None else [torch.cat([getattr(s, k)[i] for s in sets], 0) for i in range(len(getattr(sets[0], k)))]
        return MultiDataSet(cat("features"), cat("labels"), cat("featuresMasks"), cat("labelsMasks"))
