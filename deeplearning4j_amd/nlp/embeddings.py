"""SequenceVectors / Word2Vec / ParagraphVectors / lookup table (reference: deeplearning4j-nlp/.../models/
{sequencevectors/SequenceVectors,word2vec/Word2Vec,paragraphvectors/ParagraphVectors}.java,
embeddings/{inmemory/InMemoryLookupTable,wordvectors/WordVectorsImpl,learning/impl/elements/{SkipGram,CBOW},
learning/impl/sequence/{DBOW,DM}}.java; native kernels libnd4j helpers/cuda/sg_cb.cu).

MI355X design: the reference trains one (word, context) sample per native call from N Hogwild threads.  Here the corpus is
indexed once, (input-row, target) pairs for a whole slab of text are generated vectorised on the host, and each device step
updates thousands of pairs at once: gather rows -> batched dot/sigmoid -> ``index_add_`` scatter (duplicates accumulate, the
batch-synchronous analogue of Hogwild).  Hierarchical softmax uses padded ``[V, L]`` code/point tables with a mask; negative
sampling draws from the unigram^0.75 CDF on device.  All tables stay resident in HBM for the whole fit.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from .iterators import (BasicLabelAwareIterator, CollectionSentenceIterator, LabelAwareIterator, LabelledDocument, LabelsSource,
                        SentenceIterator)
from .tokenization import DefaultTokenizerFactory, TokenizerFactory
from .vocab import AbstractCache, Huffman, VocabConstructor, VocabWord, MAX_CODE_LENGTH


def _default_device():
    from ..nn.network_base import default_device
    return default_device()


# ----------------------------------------------------------------------------- lookup table
class InMemoryLookupTable:
    def __init__(self, vocab: AbstractCache, vectorLength=100, useHS=True, negative=0.0, seed=12345, device=None, dtype=torch.float32):
        self.vocab, self.vectorLength, self.useHS, self.negative, self.seed = vocab, int(vectorLength), useHS, float(negative), seed
        self.device = torch.device(device) if device is not None else _default_device()
        self.dtype = dtype
        self.syn0 = self.syn1 = self.syn1neg = None
        self._neg_cdf = None

    def resetWeights(self, reset=True):
        if self.syn0 is not None and not reset:
            return
        V, D = self.vocab.numWords(), self.vectorLength
        g = torch.Generator().manual_seed(int(self.seed))
        self.syn0 = ((torch.rand(V, D, generator=g, dtype=torch.float32) - 0.5) / D).to(self.device, self.dtype)
        self.syn1 = torch.zeros(V, D, device=self.device, dtype=self.dtype) if self.useHS else None
        self.syn1neg = torch.zeros(V, D, device=self.device, dtype=self.dtype) if self.negative > 0 else None
        self._neg_cdf = None

    def neg_cdf(self):
        """Unigram^0.75 cumulative distribution (the reference materialises a 1e8-entry table; a CDF + searchsorted is exact)."""
        if self._neg_cdf is None:
            f = torch.tensor([0.0 if w.isLabel else w.elementFrequency for w in self.vocab.vocabWords()], dtype=torch.float64)
            p = f.pow(0.75)
            self._neg_cdf = (p.cumsum(0) / p.sum()).to(self.device, torch.float32)
        return self._neg_cdf

    def vector(self, word):
        i = self.vocab.indexOf(word)
        return None if i < 0 else self.syn0[i]

    def getWeights(self):
        return self.syn0

    def setSyn0(self, w):
        self.syn0 = torch.as_tensor(w).to(self.device, self.dtype)

    def layerSize(self):
        return self.vectorLength

    def putVector(self, word, vec):
        self.syn0[self.vocab.indexOf(word)] = torch.as_tensor(vec).to(self.device, self.dtype)

    def getVocabCache(self):
        return self.vocab


# ----------------------------------------------------------------------------- batched learning kernels
def _hs_tables(vocab: AbstractCache, device):
    ws = vocab.vocabWords()
    L = max([w.codeLength for w in ws] + [1])
    codes = torch.zeros(len(ws), L, dtype=torch.float32)
    points = torch.zeros(len(ws), L, dtype=torch.int64)
    mask = torch.zeros(len(ws), L, dtype=torch.float32)
    for i, w in enumerate(ws):
        n = w.codeLength
        if n:
            codes[i, :n] = torch.tensor(w.codes[:n], dtype=torch.float32)
            points[i, :n] = torch.tensor(w.points[:n], dtype=torch.int64)
            mask[i, :n] = 1
    return codes.to(device), points.to(device), mask.to(device)


def _scatter(mat, idx, upd):
    """``mat[idx] += upd`` with duplicate rows damped by 1/sqrt(multiplicity).

    Hogwild applies duplicate updates sequentially, each seeing the previous one's effect (the sigmoid saturates); a batched
    scatter sums gradients that were all computed from the same stale row, which overshoots by the multiplicity for frequent
    words / small vocabularies.  1/sqrt(n) keeps frequent rows learning faster than rare ones while staying stable."""
    cnt = torch.bincount(idx, minlength=mat.shape[0]).to(upd.dtype)
    mat.index_add_(0, idx, upd * cnt[idx].rsqrt().unsqueeze(1))


def _train_batch(table: InMemoryLookupTable, l1, targets, alpha, hs, n_neg, gen, train_outputs=True):
    """l1: [B, D] input vectors; targets: [B] predicted word rows.  Returns neu1e [B, D]; updates syn1 / syn1neg in place."""
    D = l1.shape[1]
    if l1.is_cuda and l1.dtype == torch.float32 and train_outputs:
        return _train_batch_hip(table, l1.contiguous(), targets, alpha, hs, n_neg, gen)
    neu1e = torch.zeros_like(l1)
    if hs is not None:
        codes, points, mask = hs
        pts, cds, msk = points[targets], codes[targets], mask[targets]               # [B, L]
        rows = table.syn1[pts]                                                         # [B, L, D]
        f = torch.sigmoid(torch.einsum("bld,bd->bl", rows, l1))
        g = (1.0 - cds - f) * alpha * msk
        neu1e += torch.einsum("bl,bld->bd", g, rows)
        if train_outputs:
            _scatter(table.syn1, pts.reshape(-1), (g.unsqueeze(2) * l1.unsqueeze(1)).reshape(-1, D))
    if n_neg > 0:
        B = targets.shape[0]
        r = torch.rand(B, n_neg, device=l1.device, generator=gen)
        negs = torch.searchsorted(table.neg_cdf(), r).clamp_(max=table.syn1neg.shape[0] - 1)
        tg = torch.cat([targets.unsqueeze(1), negs], 1)                               # [B, 1+K]
        lab = torch.zeros_like(tg, dtype=l1.dtype)
        lab[:, 0] = 1
        valid = torch.ones_like(lab)
        valid[:, 1:] = (negs != targets.unsqueeze(1)).to(l1.dtype)                    # word2vec skips a negative equal to the target
        rows = table.syn1neg[tg]
        f = torch.sigmoid(torch.einsum("bkd,bd->bk", rows, l1))
        g = (lab - f) * alpha * valid
        neu1e += torch.einsum("bk,bkd->bd", g, rows)
        if train_outputs:
            _scatter(table.syn1neg, tg.reshape(-1), (g.unsqueeze(2) * l1.unsqueeze(1)).reshape(-1, D))
    return neu1e


def _train_batch_hip(table, l1, targets, alpha, hs, n_neg, gen):
    """GPU path: the fused skip-gram/CBOW kernel (native/csrc/helpers_data.hip sg_pairs_kernel; reference sg_cb.cu
    hSoftmaxKernel / nSamplingKernel): one wave per (input, output row, label) triple does dot -> sigmoid -> both
    gradients; output rows are updated with fp32 atomics as the triples retire (hogwild, like the reference), so no
    duplicate damping is needed."""
    from ..ops import helpers as H
    B, D = l1.shape
    neu1e = torch.zeros_like(l1)

    def run(syn, tg, lab, saturation):
        K = tg.shape[1]
        hidden = l1.unsqueeze(1).expand(B, K, D).reshape(B * K, D).contiguous()
        dh = H.sg_pairs_(table.syn0, syn, tg.reshape(-1), tg.reshape(-1).contiguous(), lab.reshape(-1).contiguous(), float(alpha),
                         hidden=hidden, saturation=saturation)
        return dh.view(B, K, D).sum(1)

    if hs is not None:
        codes, points, mask = hs
        pts = torch.where(mask[targets] > 0, points[targets], torch.full_like(points[targets], -1))
        neu1e += run(table.syn1, pts.long(), (1.0 - codes[targets]).float(), 1)
    if n_neg > 0:
        r = torch.rand(B, n_neg, device=l1.device, generator=gen)
        negs = torch.searchsorted(table.neg_cdf(), r).clamp_(max=table.syn1neg.shape[0] - 1)
        negs = torch.where(negs == targets.unsqueeze(1), torch.full_like(negs, -1), negs)   # word2vec skips a negative equal to the target
        tg = torch.cat([targets.unsqueeze(1), negs], 1)
        lab = torch.zeros(tg.shape, dtype=torch.float32, device=l1.device)
        lab[:, 0] = 1
        neu1e += run(table.syn1neg, tg.long(), lab, 2)
    return neu1e


class _Pairs:
    """Vectorised (center, context) pair generation with word2vec's random reduced window + frequent-word subsampling."""

    def __init__(self, window, sampling, total_words, freqs, rng: np.random.Generator):
        self.window, self.sampling, self.total, self.freqs, self.rng = window, sampling, float(total_words), freqs, rng

    def subsample(self, ids: np.ndarray) -> np.ndarray:
        if self.sampling <= 0 or ids.size == 0:
            return ids
        f = self.freqs[ids]
        ran = (np.sqrt(f / (self.sampling * self.total)) + 1) * (self.sampling * self.total) / f
        return ids[ran >= self.rng.random(ids.size)]

    def skipgram(self, ids: np.ndarray):
        n = ids.size
        if n < 2:
            return np.empty(0, np.int64), np.empty(0, np.int64)
        b = self.rng.integers(0, self.window, n)
        span = self.window - b                                            # effective half-window per position
        offs = np.concatenate([np.arange(-self.window, 0), np.arange(1, self.window + 1)])
        pos = np.arange(n)[:, None] + offs[None, :]
        ok = (np.abs(offs)[None, :] <= span[:, None]) & (pos >= 0) & (pos < n)
        ci, cj = np.nonzero(ok)
        return ids[ci], ids[pos[ci, cj]]                                   # (center, context)

    def cbow(self, ids: np.ndarray):
        n = ids.size
        if n < 2:
            return np.empty(0, np.int64), np.empty((0, 2 * self.window), np.int64)
        b = self.rng.integers(0, self.window, n)
        span = self.window - b
        offs = np.concatenate([np.arange(-self.window, 0), np.arange(1, self.window + 1)])
        pos = np.arange(n)[:, None] + offs[None, :]
        ok = (np.abs(offs)[None, :] <= span[:, None]) & (pos >= 0) & (pos < n)
        ctx = np.where(ok, ids[np.clip(pos, 0, n - 1)], -1)
        keep = ok.any(1)
        return ids[keep], ctx[keep]


# ----------------------------------------------------------------------------- WordVectors API
class WordVectorsImpl:
    def __init__(self):
        self.vocab: Optional[AbstractCache] = None
        self.lookupTable: Optional[InMemoryLookupTable] = None
        self.layerSize = 100
        self._normed = None

    # -- accessors
    def getVocab(self):
        return self.vocab

    vocab_ = getVocab

    def getLookupTable(self):
        return self.lookupTable

    def lookupTable_(self):
        return self.lookupTable

    def hasWord(self, w):
        return self.vocab.containsWord(w)

    def indexOf(self, w):
        return self.vocab.indexOf(w)

    def getWordVector(self, w):
        v = self.lookupTable.vector(w)
        return None if v is None else v.detach().cpu().numpy().astype(np.float64)

    def getWordVectorMatrix(self, w):
        return self.lookupTable.vector(w)

    def getWordVectorMatrixNormalized(self, w):
        v = self.lookupTable.vector(w)
        return None if v is None else v / v.norm().clamp_min(1e-12)

    def getWordVectors(self, words: Iterable[str]):
        idx = [self.vocab.indexOf(w) for w in words]
        idx = [i for i in idx if i >= 0]
        return self.lookupTable.syn0[torch.tensor(idx, dtype=torch.int64, device=self.lookupTable.device)]

    def getWordVectorsMean(self, words):
        m = self.getWordVectors(words)
        return m.mean(0) if m.shape[0] else torch.zeros(self.lookupTable.vectorLength, device=self.lookupTable.device)

    def similarity(self, a, b):
        if a == b:
            return 1.0
        va, vb = self.lookupTable.vector(a), self.lookupTable.vector(b)
        if va is None or vb is None:
            return float("nan")
        return float(torch.dot(va, vb) / (va.norm() * vb.norm()).clamp_min(1e-12))

    def _unit(self):
        s = self.lookupTable.syn0
        if self._normed is None or self._normed[0] is not s or self._normed[1] != s._version:
            self._normed = (s, s._version, s / s.norm(dim=1, keepdim=True).clamp_min(1e-12))
        return self._normed[2]

    def wordsNearest(self, positive, negative=None, top=10):
        """wordsNearest(word, n) | wordsNearest(vector, n) | wordsNearest(positive[], negative[], n)."""
        if isinstance(negative, int) and not isinstance(negative, bool):
            top, negative = negative, None
        exclude = set()
        if isinstance(positive, str):
            positive = [positive]
        if isinstance(positive, (list, tuple)) and positive and isinstance(positive[0], str):
            neg = list(negative or [])
            exclude = set(positive) | set(neg)
            vs = [self.getWordVectorMatrixNormalized(w) for w in positive if self.hasWord(w)]
            ns = [self.getWordVectorMatrixNormalized(w) for w in neg if self.hasWord(w)]
            if not vs:
                return []
            q = torch.stack(vs).sum(0) - (torch.stack(ns).sum(0) if ns else 0)
        else:
            q = torch.as_tensor(positive, dtype=self.lookupTable.dtype).to(self.lookupTable.device).reshape(-1)
        q = q / q.norm().clamp_min(1e-12)
        sims = self._unit() @ q
        k = min(sims.numel(), top + len(exclude) + sum(1 for w in self.vocab.vocabWords() if w.isLabel))
        order = torch.topk(sims, k).indices.tolist()
        out = []
        for i in order:
            w = self.vocab.elementAtIndex(i)
            if w.word in exclude or w.isLabel:
                continue
            out.append(w.word)
            if len(out) == top:
                break
        return out

    def wordsNearestSum(self, positive, negative=None, top=10):
        return self.wordsNearest(positive, negative, top)

    def similarWordsInVocabTo(self, word, accuracy):
        import difflib
        return [w for w in self.vocab.words() if difflib.SequenceMatcher(None, word, w).ratio() >= accuracy]

    def accuracy(self, questions: Sequence[Sequence[str]]):
        """Analogy accuracy: each question is (a, b, c, expected): b - a + c ≈ expected."""
        ok = tot = 0
        for a, b, c, d in questions:
            if all(self.hasWord(w) for w in (a, b, c, d)):
                tot += 1
                ok += int(d in self.wordsNearest([b, c], [a], 1))
        return ok / tot if tot else 0.0


# ----------------------------------------------------------------------------- SequenceVectors
class SequenceVectors(WordVectorsImpl):
    """Generic embedding trainer over sequences of string elements (SequenceVectors.java)."""

    def __init__(self, **kw):
        super().__init__()
        self.minWordFrequency = kw.pop("minWordFrequency", 5)
        self.layerSize = kw.pop("layerSize", 100)
        self.window = kw.pop("windowSize", 5)
        self.iterations = kw.pop("iterations", 1)
        self.epochs = kw.pop("epochs", 1)
        self.learningRate = kw.pop("learningRate", 0.025)
        self.minLearningRate = kw.pop("minLearningRate", 1e-4)
        self.negative = float(kw.pop("negativeSample", 0.0))
        self.useHS = kw.pop("useHierarchicSoftmax", True)
        self.sampling = float(kw.pop("sampling", 0.0))
        self.seed = kw.pop("seed", 12345)
        self.batchSize = kw.pop("batchSize", 8192)
        self.workers = kw.pop("workers", 1)
        self.stopWords = list(kw.pop("stopWords", []))
        self.elementsLearningAlgorithm = str(kw.pop("elementsLearningAlgorithm", "SkipGram"))
        self.sequenceLearningAlgorithm = str(kw.pop("sequenceLearningAlgorithm", "DBOW"))
        self.trainElements = kw.pop("trainElementsRepresentation", True)
        self.trainSequences = kw.pop("trainSequencesRepresentation", False)
        self.resetModel = kw.pop("resetModel", True)
        self.device = kw.pop("device", None)
        self.vocabLimit = kw.pop("limitVocabularySize", 0)
        self.useUnknown = kw.pop("useUnknown", False)
        self.unknownElement = kw.pop("unknownElement", "UNK")
        self._sequences = kw.pop("sequences", None)          # iterable of token lists, or of (tokens, labels)
        self.vocab = kw.pop("vocabCache", None)
        self.lookupTable = kw.pop("lookupTable", None)
        if kw:
            raise TypeError("unknown options: %s" % sorted(kw))
        if not self.useHS and self.negative <= 0:
            self.useHS = True

    # -- overridable corpus access
    def _iter_sequences(self):
        for s in self._sequences:
            yield s

    def buildVocab(self):
        vc = VocabConstructor(self.minWordFrequency, self.stopWords, self.useHS, self.vocab if self.vocab is not None else AbstractCache(),
                              self.vocabLimit)
        self.vocab = vc.buildJointVocabulary(self._iter_sequences())
        if self.useUnknown and not self.vocab.containsWord(self.unknownElement):
            self.vocab.addToken(VocabWord(1.0, self.unknownElement))
            self.vocab.rebuild_index()
            if self.useHS:
                Huffman(self.vocab.vocabWords()).build()
        return self.vocab

    def _index_corpus(self):
        idx, unk = self.vocab.indexOf, (self.vocab.indexOf(self.unknownElement) if self.useUnknown else -1)
        out = []
        for s in self._iter_sequences():
            labels = ()
            if isinstance(s, tuple) and len(s) == 2 and not isinstance(s[0], str):
                s, labels = s
            ids = np.fromiter(((idx(t) if idx(t) >= 0 else unk) for t in s), dtype=np.int64, count=len(s))
            ids = ids[ids >= 0]
            # label rows must not appear as ordinary tokens
            out.append((ids, [idx(l) for l in labels if idx(l) >= 0]))
        return out

    def fit(self):
        if self.vocab is None or self.vocab.numWords() == 0 or self.resetModel:
            if self.vocab is None or self.vocab.numWords() == 0:
                self.buildVocab()
        if self.vocab.numWords() == 0:
            raise RuntimeError("Vocabulary is empty: check minWordFrequency / the corpus iterator")
        if self.lookupTable is None:
            self.lookupTable = InMemoryLookupTable(self.vocab, self.layerSize, self.useHS, self.negative, self.seed, self.device)
        self.lookupTable.resetWeights(self.resetModel or self.lookupTable.syn0 is None)
        t = self.lookupTable
        dev = t.device
        corpus = self._index_corpus()
        freqs = np.array([w.elementFrequency for w in self.vocab.vocabWords()], dtype=np.float64)
        total_words = float(sum(len(ids) for ids, _ in corpus)) or 1.0
        rng = np.random.default_rng(self.seed)
        gen = torch.Generator(device=dev).manual_seed(int(self.seed))
        pg = _Pairs(self.window, self.sampling, self.vocab.totalWordOccurrences() or total_words, freqs, rng)
        hs = _hs_tables(self.vocab, dev) if self.useHS else None
        n_neg = int(self.negative)
        total_steps = total_words * self.epochs * self.iterations
        seen = 0.0
        cbow = self.elementsLearningAlgorithm.lower().endswith("cbow")
        dm = self.sequenceLearningAlgorithm.lower().endswith("dm")
        sg_in, sg_tg, cb_ctx, cb_tg = [], [], [], []
        npairs = 0
        W2 = 2 * self.window + 1                       # cbow context width (+1 slot for a DM label row), -1 padded

        def flush():
            nonlocal sg_in, sg_tg, cb_ctx, cb_tg, npairs
            alpha = max(self.minLearningRate, self.learningRate * (1.0 - seen / (total_steps + 1)))
            if sg_in:
                inp = torch.from_numpy(np.concatenate(sg_in)).to(dev)
                tg = torch.from_numpy(np.concatenate(sg_tg)).to(dev)
                for s in range(0, inp.numel(), self.batchSize):
                    i_, t_ = inp[s:s + self.batchSize], tg[s:s + self.batchSize]
                    neu1e = _train_batch(t, t.syn0[i_], t_, alpha, hs, n_neg, gen)
                    _scatter(t.syn0, i_, neu1e)
            if cb_ctx:
                ctx = torch.from_numpy(np.concatenate(cb_ctx)).to(dev)
                tg = torch.from_numpy(np.concatenate(cb_tg)).to(dev)
                for s in range(0, ctx.shape[0], self.batchSize):
                    c_, t_ = ctx[s:s + self.batchSize], tg[s:s + self.batchSize]
                    m = (c_ >= 0).to(t.dtype)
                    cc = c_.clamp_min(0)
                    cnt = m.sum(1, keepdim=True).clamp_min(1)
                    neu1 = (t.syn0[cc] * m.unsqueeze(2)).sum(1) / cnt
                    neu1e = _train_batch(t, neu1, t_, alpha, hs, n_neg, gen)
                    upd = (neu1e.unsqueeze(1) * m.unsqueeze(2)).reshape(-1, neu1e.shape[1])
                    _scatter(t.syn0, cc.reshape(-1), upd)
            sg_in, sg_tg, cb_ctx, cb_tg, npairs = [], [], [], [], 0

        def pad(ctx, lab=-1):
            return np.concatenate([ctx, np.full((ctx.shape[0], W2 - ctx.shape[1]), lab, np.int64)], 1)

        slab = max(self.batchSize * 8, 1 << 16)
        for _ in range(self.epochs):
            order = rng.permutation(len(corpus)) if len(corpus) > 1 else np.arange(len(corpus))
            for si in order:
                ids, labels = corpus[si]
                for _it in range(self.iterations):
                    sub = pg.subsample(ids)
                    if self.trainElements and sub.size > 1:
                        if cbow:
                            c, ctx = pg.cbow(sub)
                            if c.size:
                                cb_ctx.append(pad(ctx)); cb_tg.append(c); npairs += c.size
                        else:
                            c, ctx = pg.skipgram(sub)
                            if c.size:
                                sg_in.append(ctx); sg_tg.append(c); npairs += c.size          # input row = context word, target = center
                    if self.trainSequences and labels and sub.size:
                        for lab in labels:
                            if dm:                                                             # DM: label row joins the CBOW context
                                if sub.size > 1:
                                    c, ctx = pg.cbow(sub)
                                else:
                                    c, ctx = sub, np.full((sub.size, 2 * self.window), -1, np.int64)
                                if c.size:
                                    cb_ctx.append(pad(ctx, lab)); cb_tg.append(c); npairs += c.size
                            else:                                                              # DBOW: label row predicts every word
                                sg_in.append(np.full(sub.size, lab, np.int64)); sg_tg.append(sub); npairs += sub.size
                    seen += ids.size
                    if npairs >= slab:
                        flush()
            flush()
        self._normed = None
        return self

    # -- config echo (DL4J getters)
    def getLayerSize(self):
        return self.layerSize

    def getWindow(self):
        return self.window


def _builder_for(cls, aliases=None):
    aliases = aliases or {}

    class Builder:
        def __init__(self):
            self._kw = {}

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)

            def setter(*v):
                self._kw[aliases.get(name, name)] = v[0] if len(v) == 1 else list(v)
                return self
            return setter

        def build(self):
            return cls(**self._kw)
    Builder.__qualname__ = cls.__name__ + ".Builder"
    return Builder


SequenceVectors.Builder = _builder_for(SequenceVectors)


# ----------------------------------------------------------------------------- Word2Vec
class Word2Vec(SequenceVectors):
    """Word2Vec.Builder().minWordFrequency(5).layerSize(100).windowSize(5).iterate(sentenceIter).tokenizerFactory(tf).build().fit()"""

    def __init__(self, **kw):
        self.sentenceIterator = kw.pop("iterate", None)
        self.tokenizerFactory: TokenizerFactory = kw.pop("tokenizerFactory", None) or DefaultTokenizerFactory()
        kw.pop("allowParallelTokenization", None)
        kw.pop("enableScavenger", None)
        kw.pop("usePreciseWeightInit", None)
        kw.pop("usePreciseMode", None)
        kw.pop("modelUtils", None)
        kw.pop("useAdaGrad", None)
        kw.pop("vocabCache_", None)
        super().__init__(**kw)
        self._tok_cache = None

    def _tokens(self, text):
        return self.tokenizerFactory.create(text).getTokens()

    def _iter_sequences(self):
        if self._sequences is not None:
            yield from self._sequences
            return
        if self._tok_cache is None:
            it = self.sentenceIterator
            if isinstance(it, (list, tuple)):
                it = CollectionSentenceIterator(it)
            self._tok_cache = [self._tokens(s) for s in it]
        yield from self._tok_cache

    def setSentenceIterator(self, it):
        self.sentenceIterator, self._tok_cache = it, None

    def setTokenizerFactory(self, tf):
        self.tokenizerFactory, self._tok_cache = tf, None


Word2Vec.Builder = _builder_for(Word2Vec)


# ----------------------------------------------------------------------------- ParagraphVectors
class ParagraphVectors(Word2Vec):
    """Doc2vec: DBOW (label row predicts words) / DM (label joins the CBOW context); ``inferVector`` for unseen text and
    ``predict`` / ``nearestLabels`` classification helpers (ParagraphVectors.java)."""

    def __init__(self, **kw):
        self.labelAwareIterator: Optional[LabelAwareIterator] = kw.pop("iterate", None)
        self.labelsSource: Optional[LabelsSource] = kw.pop("labelsSource", None)
        self.trainWordVectors = kw.pop("trainWordVectors", True)
        kw.setdefault("trainSequencesRepresentation", True)
        kw["trainElementsRepresentation"] = kw.pop("trainElementsRepresentation", self.trainWordVectors)
        kw.setdefault("minWordFrequency", 1)
        super().__init__(**kw)
        if isinstance(self.labelAwareIterator, SentenceIterator):
            self.labelAwareIterator = BasicLabelAwareIterator(self.labelAwareIterator, self.labelsSource)
        elif isinstance(self.labelAwareIterator, (list, tuple)) and self.labelAwareIterator and isinstance(self.labelAwareIterator[0], str):
            self.labelAwareIterator = BasicLabelAwareIterator(CollectionSentenceIterator(self.labelAwareIterator), self.labelsSource)
        elif isinstance(self.labelAwareIterator, (list, tuple)):
            from .iterators import SimpleLabelAwareIterator
            self.labelAwareIterator = SimpleLabelAwareIterator(self.labelAwareIterator)
        if self.labelAwareIterator is not None:
            self.labelsSource = self.labelAwareIterator.getLabelsSource()

    def _iter_sequences(self):
        if self._sequences is not None:
            yield from self._sequences
            return
        if self._tok_cache is None:
            self._tok_cache = [(self._tokens(d.content), list(d.labels)) for d in self.labelAwareIterator]
        yield from self._tok_cache

    # -- inference
    def inferVector(self, text, learningRate=None, minLearningRate=None, iterations=None):
        toks = text if isinstance(text, (list, tuple)) else self._tokens(text)
        t = self.lookupTable
        ids = np.array([i for i in (self.vocab.indexOf(w) for w in toks) if i >= 0 and not self.vocab.elementAtIndex(i).isLabel], dtype=np.int64)
        lr = self.learningRate if learningRate is None else learningRate
        mlr = self.minLearningRate if minLearningRate is None else minLearningRate
        iters = iterations or max(self.iterations * 5, 10)
        g = torch.Generator().manual_seed(int(self.seed) ^ (hash(tuple(ids.tolist())) & 0x7FFFFFFF))
        vec = ((torch.rand(1, t.vectorLength, generator=g) - 0.5) / t.vectorLength).to(t.device, t.dtype)
        if ids.size == 0:
            return vec[0]
        hs = _hs_tables(self.vocab, t.device) if self.useHS else None
        gen = torch.Generator(device=t.device).manual_seed(int(self.seed))
        tg = torch.from_numpy(ids).to(t.device)
        for k in range(iters):
            alpha = max(mlr, lr * (1 - k / iters))
            neu1e = _train_batch(t, vec.expand(tg.numel(), -1), tg, alpha, hs, int(self.negative), gen, train_outputs=False)
            vec = vec + neu1e.sum(0, keepdim=True)
        return vec[0]

    def _label_rows(self):
        labs = [w for w in self.vocab.vocabWords() if w.isLabel]
        idx = torch.tensor([w.index for w in labs], dtype=torch.int64, device=self.lookupTable.device)
        return labs, self.lookupTable.syn0[idx]

    def nearestLabels(self, text_or_vec, topN=1):
        v = text_or_vec if isinstance(text_or_vec, torch.Tensor) else self.inferVector(text_or_vec)
        labs, m = self._label_rows()
        sims = (m / m.norm(dim=1, keepdim=True).clamp_min(1e-12)) @ (v / v.norm().clamp_min(1e-12))
        order = torch.topk(sims, min(topN, len(labs))).indices.tolist()
        return [labs[i].word for i in order]

    def predict(self, text):
        r = self.nearestLabels(text, 1)
        return r[0] if r else None

    def predictSeveral(self, text, limit):
        return self.nearestLabels(text, limit)

    def similarityToLabel(self, text, label):
        v = self.inferVector(text)
        l = self.lookupTable.vector(label)
        return float(torch.dot(v, l) / (v.norm() * l.norm()).clamp_min(1e-12))

    def getLabelsSource(self):
        return self.labelsSource


ParagraphVectors.Builder = _builder_for(ParagraphVectors)
