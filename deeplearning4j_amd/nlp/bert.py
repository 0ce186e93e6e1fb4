"""BertIterator + masked-LM masker + sentence providers (reference: deeplearning4j-nlp/.../iterator/BertIterator.java,
iterator/bert/{BertMaskedLMMasker,BertSequenceMasker}.java, iterator/provider/{CollectionLabeledSentenceProvider,
CollectionLabeledPairSentenceProvider}.java)."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ..datasets.dataset import MultiDataSet
from .tokenization import BertWordPieceTokenizerFactory


class CollectionLabeledSentenceProvider:
    def __init__(self, sentences: Sequence[str], labels: Optional[Sequence[str]] = None, rng=None):
        self.s, self.l = list(sentences), (list(labels) if labels is not None else None)
        self._labels = sorted(set(self.l)) if self.l is not None else []
        self._rng, self._order, self._i = rng, list(range(len(self.s))), 0
        self.reset()

    def hasNext(self):
        return self._i < len(self.s)

    def nextSentence(self) -> Tuple[str, Optional[str]]:
        k = self._order[self._i]
        self._i += 1
        return self.s[k], (self.l[k] if self.l is not None else None)

    def reset(self):
        self._i = 0
        if self._rng is not None:
            self._order = list(np.random.default_rng(self._rng).permutation(len(self.s)))

    def totalNumSentences(self):
        return len(self.s)

    def allLabels(self):
        return list(self._labels)

    def numLabelClasses(self):
        return len(self._labels)


class CollectionLabeledPairSentenceProvider(CollectionLabeledSentenceProvider):
    def __init__(self, left: Sequence[str], right: Sequence[str], labels=None, rng=None):
        super().__init__(left, labels, rng)
        self.r = list(right)

    def nextSentencePair(self):
        k = self._order[self._i]
        self._i += 1
        return self.s[k], self.r[k], (self.l[k] if self.l is not None else None)


class BertMaskedLMMasker:
    """15% of tokens selected; of those 80% -> [MASK], 10% -> random token, 10% unchanged (BertMaskedLMMasker.java)."""
    DEFAULT_MASK_PROB, DEFAULT_MASK_TOKEN_PROB, DEFAULT_RANDOM_WORD_PROB = 0.15, 0.8, 0.1

    def __init__(self, seed=12345, maskProb=0.15, maskTokenProb=0.8, randomTokenProb=0.1):
        self.r = np.random.default_rng(seed)
        self.maskProb, self.maskTokenProb, self.randomTokenProb = maskProb, maskTokenProb, randomTokenProb

    def maskSequence(self, tokens: List[str], maskToken: str, vocabWords: List[str]):
        out, masked = [], []
        for t in tokens:
            if self.r.random() < self.maskProb:
                masked.append(True)
                d = self.r.random()
                if d < self.maskTokenProb:
                    out.append(maskToken)
                elif d < self.maskTokenProb + self.randomTokenProb:
                    out.append(vocabWords[int(self.r.integers(0, len(vocabWords)))])
                else:
                    out.append(t)
            else:
                masked.append(False)
                out.append(t)
        return out, masked


class BertIterator:
    class Task:
        UNSUPERVISED, SEQ_CLASSIFICATION = "UNSUPERVISED", "SEQ_CLASSIFICATION"

    class LengthHandling:
        FIXED_LENGTH, ANY_LENGTH, CLIP_ONLY = "FIXED_LENGTH", "ANY_LENGTH", "CLIP_ONLY"

    class FeatureArrays:
        INDICES_MASK, INDICES_MASK_SEGMENTID = "INDICES_MASK", "INDICES_MASK_SEGMENTID"

    class UnsupervisedLabelFormat:
        RANK2_IDX, RANK3_NCL, RANK3_LNC = "RANK2_IDX", "RANK3_NCL", "RANK3_LNC"

    def __init__(self, tokenizerFactory: BertWordPieceTokenizerFactory, task="SEQ_CLASSIFICATION", lengthHandling="FIXED_LENGTH", maxTokens=128,
                 minibatchSize=32, padMinibatches=False, featureArrays="INDICES_MASK", vocabMap: Optional[Dict[str, int]] = None,
                 sentenceProvider=None, sentencePairProvider=None, unsupervisedLabelFormat="RANK2_IDX", masker=None, maskToken="[MASK]",
                 prependToken="[CLS]", appendToken=None, preProcessor=None):
        self.tf, self.task, self.lengthHandling, self.maxTokens = tokenizerFactory, task, lengthHandling, int(maxTokens)
        self.minibatchSize, self.padMinibatches, self.featureArrays = int(minibatchSize), padMinibatches, featureArrays
        self.vocabMap = vocabMap or tokenizerFactory.getVocab()
        self.sentenceProvider, self.sentencePairProvider = sentenceProvider, sentencePairProvider
        self.unsupervisedLabelFormat, self.masker, self.maskToken = unsupervisedLabelFormat, masker or BertMaskedLMMasker(), maskToken
        self.prependToken, self.appendToken, self.preProcessor = prependToken, appendToken, preProcessor
        self._vocabWords = [w for w, _ in sorted(self.vocabMap.items(), key=lambda kv: kv[1])]
        if sentenceProvider is None and sentencePairProvider is None:
            raise ValueError("No sentence provider set")
        if task == self.Task.UNSUPERVISED and maskToken not in self.vocabMap:
            raise ValueError("Mask token '%s' is not in the vocabulary" % maskToken)
        self._prov = sentencePairProvider or sentenceProvider
        self._labels = self._prov.allLabels() if task == self.Task.SEQ_CLASSIFICATION else []

    class Builder:
        def __init__(self):
            self._kw = {}

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)

            def setter(*v):
                if name == "lengthHandling":
                    self._kw["lengthHandling"], self._kw["maxTokens"] = v[0], v[1]
                elif name == "tokenizer":
                    self._kw["tokenizerFactory"] = v[0]
                else:
                    self._kw[name] = v[0]
                return self
            return setter

        def build(self):
            return BertIterator(**self._kw)

    @staticmethod
    def builder():
        return BertIterator.Builder()

    # -- iteration
    def hasNext(self):
        return self._prov.hasNext()

    def reset(self):
        self._prov.reset()

    def __iter__(self):
        self.reset()
        while self.hasNext():
            yield self.next()

    def setPreProcessor(self, p):
        self.preProcessor = p

    def _tokenize(self, text, first=True) -> List[str]:
        toks = self.tf.create(text).getTokens()
        if first and self.prependToken:
            toks = [self.prependToken] + toks
        if self.appendToken:
            toks = toks + [self.appendToken]
        return toks

    def _clip(self, toks):
        if self.lengthHandling != self.LengthHandling.ANY_LENGTH and len(toks) > self.maxTokens:
            toks = toks[:self.maxTokens]
            if self.appendToken:
                toks[-1] = self.appendToken
        return toks

    def _clip_pair(self, a, b):
        if self.lengthHandling == self.LengthHandling.ANY_LENGTH:
            return a, b
        while len(a) + len(b) > self.maxTokens:                      # trim the longer side, keeping its trailing separator
            side = a if len(a) >= len(b) else b
            keep_sep = bool(self.appendToken) and side[-1] == self.appendToken
            side.pop(-2 if keep_sep and len(side) > 1 else -1)
        return a, b

    def next(self, num=None) -> MultiDataSet:
        n = num or self.minibatchSize
        rows, segs, labs = [], [], []
        while len(rows) < n and self._prov.hasNext():
            if self.sentencePairProvider is not None:
                l, r, lab = self._prov.nextSentencePair()
                a, b = self._clip_pair(self._tokenize(l, True), self._tokenize(r, False))
                rows.append(a + b)
                segs.append([0] * len(a) + [1] * len(b))
            else:
                s, lab = self._prov.nextSentence()
                t = self._clip(self._tokenize(s))
                rows.append(t)
                segs.append([0] * len(t))
            labs.append(lab)
        return self._assemble(rows, segs, labs, n)

    def featurizeSentences(self, sentences: Sequence[str]):
        rows = [self._clip(self._tokenize(s)) for s in sentences]
        mds = self._assemble(rows, [[0] * len(r) for r in rows], [None] * len(rows), len(rows), labels=False)
        return mds.features, mds.featuresMasks

    def _assemble(self, rows, segs, labs, n, labels=True) -> MultiDataSet:
        mb = n if (self.padMinibatches and len(rows) < n) else len(rows)
        L = self.maxTokens if self.lengthHandling == self.LengthHandling.FIXED_LENGTH else max(len(r) for r in rows)
        unk = self.vocabMap.get("[UNK]", 0)
        masked_flags = None
        orig_rows = rows
        if labels and self.task == self.Task.UNSUPERVISED:
            masked_rows, masked_flags = [], []
            for r in rows:
                m, f = self.masker.maskSequence(r, self.maskToken, self._vocabWords)
                masked_rows.append(m)
                masked_flags.append(f)
            rows = masked_rows
        idx = np.zeros((mb, L), dtype=np.int64)
        mask = np.zeros((mb, L), dtype=np.float32)
        seg = np.zeros((mb, L), dtype=np.int64)
        for i, r in enumerate(rows):
            k = min(len(r), L)
            idx[i, :k] = [self.vocabMap.get(t, unk) for t in r[:k]]
            mask[i, :k] = 1
            seg[i, :k] = segs[i][:k]
        feats = [torch.from_numpy(idx)]
        fmasks = [torch.from_numpy(mask)]
        if self.featureArrays == self.FeatureArrays.INDICES_MASK_SEGMENTID:
            feats.append(torch.from_numpy(seg))
            fmasks.append(None)
        lab_arrs, lab_masks = None, None
        if labels and self.task == self.Task.SEQ_CLASSIFICATION:
            y = np.zeros((mb, len(self._labels)), dtype=np.float32)
            for i, l in enumerate(labs):
                if l is not None:
                    y[i, self._labels.index(l)] = 1
            lab_arrs, lab_masks = [torch.from_numpy(y)], None
        elif labels and self.task == self.Task.UNSUPERVISED:
            lm = np.zeros((mb, L), dtype=np.float32)
            true_idx = np.zeros((mb, L), dtype=np.int64)
            for i, r in enumerate(orig_rows):
                k = min(len(r), L)
                true_idx[i, :k] = [self.vocabMap.get(t, unk) for t in r[:k]]
                lm[i, :k] = np.asarray(masked_flags[i][:k], dtype=np.float32)
            V = len(self.vocabMap)
            if self.unsupervisedLabelFormat == self.UnsupervisedLabelFormat.RANK2_IDX:
                y = torch.from_numpy(true_idx)
            else:
                oh = torch.nn.functional.one_hot(torch.from_numpy(true_idx), V).float() * torch.from_numpy(mask).unsqueeze(2)
                y = oh.permute(0, 2, 1).contiguous() if self.unsupervisedLabelFormat == self.UnsupervisedLabelFormat.RANK3_NCL else \
                    oh.permute(1, 0, 2).contiguous()
            lab_arrs, lab_masks = [y], [torch.from_numpy(lm)]
        mds = MultiDataSet(feats, lab_arrs, fmasks, lab_masks)
        if self.preProcessor is not None:
            self.preProcessor.preProcess(mds)
        return mds
