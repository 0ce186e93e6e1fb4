"""FastText (reference: deeplearning4j-nlp/.../models/fasttext/FastText.java — a JNI wrapper over Facebook's fastText).

The wrapped C++ library is not available, so the algorithm itself is implemented here, device-resident: sub-word
(character n-gram, FNV-1a hashed into ``bucket`` rows) skip-gram / CBOW with negative sampling, and the supervised
classifier (averaged word + word-n-gram embeddings -> softmax).  Training is batched the same way as
``embeddings.SequenceVectors`` (gather -> batched dot -> index_add scatter)."""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .embeddings import InMemoryLookupTable, WordVectorsImpl, _Pairs, _default_device, _scatter
from .vocab import AbstractCache, VocabWord

BOW, EOW = "<", ">"


def _fnv1a(s: str) -> int:
    h = 2166136261
    for b in s.encode("utf-8"):
        # fastText casts each byte to int8 before the xor (dictionary.cc Dictionary::hash)
        h = (h ^ (b if b < 128 else (b - 256) & 0xFFFFFFFF)) & 0xFFFFFFFF
        h = (h * 16777619) & 0xFFFFFFFF
    return h


def char_ngrams(word: str, minn: int, maxn: int) -> List[str]:
    w = BOW + word + EOW
    out = []
    for i in range(len(w)):
        for n in range(minn, maxn + 1):
            if i + n <= len(w):
                g = w[i:i + n]
                if not (n == 1 and (i == 0 or i == len(w) - 1)):
                    out.append(g)
    return out


class FastText(WordVectorsImpl):
    def __init__(self, supervised=False, skipgram=False, cbow=False, inputFile=None, outputFile=None, bucket=2000000, minCount=1, minCountLabel=0,
                 wordNgrams=1, minNgramLength=3, maxNgramLength=6, samplingThreshold=1e-4, labelPrefix="__label__", learningRate=None,
                 learningRateUpdate=100, dim=100, contextWindowSize=5, epochs=5, negativeSamples=5, lossName=None, seed=12345, iterator=None,
                 batchSize=4096, device=None, quantize=False, **_ignored):
        super().__init__()
        if sum(map(bool, (supervised, skipgram, cbow))) > 1:
            raise ValueError("Only one of supervised / skipgram / cbow may be set")
        self.supervised, self.cbow = bool(supervised), bool(cbow)
        self.skipgram = bool(skipgram) or not (supervised or cbow)
        self.inputFile, self.outputFile, self.iterator = inputFile, outputFile, iterator
        self.bucket, self.minCount, self.minCountLabel, self.wordNgrams = int(bucket), minCount, minCountLabel, int(wordNgrams)
        self.minn, self.maxn = (0, 0) if supervised and minNgramLength == 3 and maxNgramLength == 6 else (int(minNgramLength), int(maxNgramLength))
        self.sampling, self.labelPrefix = samplingThreshold, labelPrefix
        self.learningRate = learningRate if learningRate is not None else (0.1 if supervised else 0.05)
        self.dim, self.window, self.epochs, self.negative = int(dim), int(contextWindowSize), int(epochs), int(negativeSamples)
        self.lossName = lossName or ("softmax" if supervised else "ns")
        self.seed, self.batchSize = seed, int(batchSize)
        self.device = torch.device(device) if device is not None else _default_device()
        self.layerSize = self.dim
        self.labels: List[str] = []
        self._input = None        # [V + bucket, D]
        self._output = None       # [V, D] (ns) or [nLabels, D] (supervised)

    class Builder:
        def __init__(self):
            self._kw = {}

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)

            def setter(*v):
                self._kw[name] = v[0] if v else True
                return self
            return setter

        def build(self):
            return FastText(**self._kw)

    @staticmethod
    def builder():
        return FastText.Builder()

    # -- corpus
    def _lines(self):
        if self.iterator is not None:
            src = self.iterator
            if hasattr(src, "reset"):
                src.reset()
            for s in src:
                yield s
        else:
            with open(os.fspath(self.inputFile), encoding="utf-8", errors="replace") as f:
                for line in f:
                    yield line.rstrip("\n")

    def _build_dict(self):
        wc: Dict[str, int] = {}
        lc: Dict[str, int] = {}
        docs = []
        for line in self._lines():
            toks = line.split()
            words = [t for t in toks if not t.startswith(self.labelPrefix)]
            labs = [t for t in toks if t.startswith(self.labelPrefix)]
            for w in words:
                wc[w] = wc.get(w, 0) + 1
            for l in labs:
                lc[l] = lc.get(l, 0) + 1
            docs.append((words, labs))
        self.vocab = AbstractCache()
        for w, n in wc.items():
            if n >= self.minCount:
                self.vocab.addToken(VocabWord(n, w))
        self.vocab.rebuild_index()
        self.labels = sorted((l for l, n in lc.items() if n >= self.minCountLabel), key=lambda l: -lc[l])
        self._label_idx = {l: i for i, l in enumerate(self.labels)}
        V = self.vocab.numWords()
        subs = []
        for i in range(V):
            w = self.vocab.wordAtIndex(i)
            ids = [i]
            if self.maxn > 0 and self.bucket > 0:
                ids += [V + _fnv1a(g) % self.bucket for g in char_ngrams(w, self.minn, self.maxn)]
            subs.append(ids)
        S = max(len(s) for s in subs) if subs else 1
        sub = np.full((V, S), -1, dtype=np.int64)
        for i, s in enumerate(subs):
            sub[i, :len(s)] = s
        self._sub = torch.from_numpy(sub).to(self.device)
        return docs

    def _subword_ids(self, word) -> List[int]:
        V = self.vocab.numWords()
        i = self.vocab.indexOf(word)
        ids = [i] if i >= 0 else []
        if self.maxn > 0 and self.bucket > 0:
            ids += [V + _fnv1a(g) % self.bucket for g in char_ngrams(word, self.minn, self.maxn)]
        return ids

    def _word_ngram_ids(self, wids: List[int]) -> List[int]:
        """Hashed word n-grams for the supervised model (Dictionary::addWordNgrams: h = h * 116049371 + next)."""
        out = []
        V = self.vocab.numWords()
        if self.wordNgrams <= 1 or self.bucket <= 0:
            return out
        hs = [_fnv1a(self.vocab.wordAtIndex(i)) for i in wids]
        for i in range(len(hs)):
            h = hs[i]
            for j in range(i + 1, min(len(hs), i + self.wordNgrams)):
                h = (h * 116049371 + hs[j]) & 0xFFFFFFFFFFFFFFFF
                out.append(V + h % self.bucket)
        return out

    # -- training
    def fit(self):
        docs = self._build_dict()
        V, D, dev = self.vocab.numWords(), self.dim, self.device
        if V == 0:
            raise RuntimeError("Empty vocabulary")
        g = torch.Generator().manual_seed(int(self.seed))
        rows = V + (self.bucket if (self.maxn > 0 or self.wordNgrams > 1) else 0)
        self._input = ((torch.rand(rows, D, generator=g) * 2 - 1) / D).to(dev)
        if self.supervised:
            self._fit_supervised(docs)
        else:
            self._fit_unsupervised(docs)
        self._finalize()
        if self.outputFile:
            self.saveModel(self.outputFile)
        return self

    def _finalize(self):
        """Materialise word vectors (mean of the word row and its n-gram rows) for the WordVectors API."""
        m = (self._sub >= 0).float()
        vec = (self._input[self._sub.clamp_min(0)] * m.unsqueeze(2)).sum(1) / m.sum(1, keepdim=True).clamp_min(1)
        self.lookupTable = InMemoryLookupTable(self.vocab, self.dim, useHS=False, negative=self.negative, device=self.device)
        self.lookupTable.syn0 = vec
        self._normed = None

    def _fit_unsupervised(self, docs):
        V, D, dev = self.vocab.numWords(), self.dim, self.device
        self._output = torch.zeros(V, D, device=dev)
        freqs = np.array([w.elementFrequency for w in self.vocab.vocabWords()], dtype=np.float64)
        p = torch.from_numpy(freqs).pow(0.5)                                   # fastText uses count^0.5 for its negative table
        cdf = (p.cumsum(0) / p.sum()).float().to(dev)
        rng = np.random.default_rng(self.seed)
        gen = torch.Generator(device=dev).manual_seed(int(self.seed))
        pg = _Pairs(self.window, self.sampling, freqs.sum(), freqs, rng)
        corpus = []
        for words, _ in docs:
            ids = np.array([i for i in (self.vocab.indexOf(w) for w in words) if i >= 0], dtype=np.int64)
            if ids.size > 1:
                corpus.append(ids)
        total = float(sum(c.size for c in corpus)) * self.epochs or 1.0
        seen = 0.0
        K = self.negative
        for ep in range(self.epochs):
            cs, ts = [], []
            for ids in corpus:
                sub = pg.subsample(ids)
                if sub.size > 1:
                    if self.cbow:
                        c, ctx = pg.cbow(sub)
                        cs.append(ctx); ts.append(c)
                    else:
                        c, ctx = pg.skipgram(sub)
                        cs.append(c); ts.append(ctx)                           # input = centre word's sub-words, target = context word
                seen += ids.size
            if not cs:
                continue
            inp = torch.from_numpy(np.concatenate(cs)).to(dev)
            tg = torch.from_numpy(np.concatenate(ts)).to(dev)
            nb = (tg.numel() + self.batchSize - 1) // self.batchSize
            for bi, s in enumerate(range(0, tg.numel(), self.batchSize)):
                alpha = self.learningRate * max(1e-4, 1.0 - (ep + bi / nb) / self.epochs)
                i_, t_ = inp[s:s + self.batchSize], tg[s:s + self.batchSize]
                if self.cbow:                                                   # [B, 2W] context words -> their sub-word rows
                    sw = self._sub[i_.clamp_min(0)]                              # [B, 2W, S]
                    sw = torch.where((i_ >= 0).unsqueeze(2), sw, torch.full_like(sw, -1)).reshape(i_.shape[0], -1)
                else:
                    sw = self._sub[i_]                                          # [B, S]
                m = (sw >= 0).float()
                swc = sw.clamp_min(0)
                h = (self._input[swc] * m.unsqueeze(2)).sum(1) / m.sum(1, keepdim=True).clamp_min(1)
                r = torch.rand(t_.numel(), K, device=dev, generator=gen)
                negs = torch.searchsorted(cdf, r).clamp_(max=V - 1)
                tgt = torch.cat([t_.unsqueeze(1), negs], 1)
                lab = torch.zeros_like(tgt, dtype=torch.float32)
                lab[:, 0] = 1
                valid = torch.ones_like(lab)
                valid[:, 1:] = (negs != t_.unsqueeze(1)).float()
                orow = self._output[tgt]
                f = torch.sigmoid(torch.einsum("bkd,bd->bk", orow, h))
                gk = (lab - f) * alpha * valid
                grad_h = torch.einsum("bk,bkd->bd", gk, orow)
                _scatter(self._output, tgt.reshape(-1), (gk.unsqueeze(2) * h.unsqueeze(1)).reshape(-1, D))
                _scatter(self._input, swc.reshape(-1), (grad_h.unsqueeze(1) * m.unsqueeze(2)).reshape(-1, D))

    def _featurize(self, words: Sequence[str]) -> List[int]:
        wids = [i for i in (self.vocab.indexOf(w) for w in words) if i >= 0]
        ids = []
        for w in words:
            ids += self._subword_ids(w) if self.maxn > 0 else ([self.vocab.indexOf(w)] if self.vocab.indexOf(w) >= 0 else [])
        return ids + self._word_ngram_ids(wids)

    def _fit_supervised(self, docs):
        D, dev = self.dim, self.device
        nL = len(self.labels)
        if nL == 0:
            raise RuntimeError("Supervised FastText needs labels (tokens starting with '%s')" % self.labelPrefix)
        self._output = torch.zeros(nL, D, device=dev)
        samples = []
        for words, labs in docs:
            ids = self._featurize(words)
            ls = [self._label_idx[l] for l in labs if l in self._label_idx]
            if ids and ls:
                samples.append((ids, ls))
        rng = np.random.default_rng(self.seed)
        bs = min(64, max(1, len(samples)))
        steps = self.epochs * math.ceil(len(samples) / bs)
        step = 0
        for _ in range(self.epochs):
            order = rng.permutation(len(samples))
            for s in range(0, len(order), bs):
                batch = [samples[k] for k in order[s:s + bs]]
                alpha = self.learningRate * max(1e-4, 1.0 - step / steps)
                step += 1
                L = max(len(b[0]) for b in batch)
                idx = torch.full((len(batch), L), -1, dtype=torch.int64)
                for i, (ids, _) in enumerate(batch):
                    idx[i, :len(ids)] = torch.tensor(ids)
                idx = idx.to(dev)
                m = (idx >= 0).float()
                ic = idx.clamp_min(0)
                cnt = m.sum(1, keepdim=True).clamp_min(1)
                h = (self._input[ic] * m.unsqueeze(2)).sum(1) / cnt
                y = torch.zeros(len(batch), nL, device=dev)
                for i, (_, ls) in enumerate(batch):
                    y[i, ls[int(rng.integers(0, len(ls)))]] = 1              # fastText samples one of the labels per step
                p = torch.softmax(h @ self._output.t(), 1)
                go = (y - p) * alpha                                          # [B, nL]
                grad_h = go @ self._output
                self._output += go.t() @ h
                self._input.index_add_(0, ic.reshape(-1), ((grad_h / cnt).unsqueeze(1) * m.unsqueeze(2)).reshape(-1, D))

    # -- inference
    def _sentence_vector(self, text) -> torch.Tensor:
        words = [t for t in text.split() if not t.startswith(self.labelPrefix)]
        ids = self._featurize(words)
        if not ids:
            return torch.zeros(self.dim, device=self.device)
        return self._input[torch.tensor(ids, device=self.device)].mean(0)

    def predictProbabilities(self, text, k=1) -> List[Tuple[str, float]]:
        if not self.supervised:
            raise RuntimeError("predict is only available for supervised models")
        p = torch.softmax(self._output @ self._sentence_vector(text), 0)
        top = torch.topk(p, min(k, p.numel()))
        return [(self.labels[i], float(v)) for v, i in zip(top.values.tolist(), top.indices.tolist())]

    def predict(self, text):
        r = self.predictProbabilities(text, 1)
        return r[0][0] if r else None

    def predictProbability(self, text):
        return self.predictProbabilities(text, 1)[0]

    def getWordVector(self, w):
        """Works for out-of-vocabulary words too, through their character n-grams."""
        ids = self._subword_ids(w)
        if not ids:
            return None
        return self._input[torch.tensor(ids, device=self.device)].mean(0).detach().cpu().numpy().astype(np.float64)

    def getWordVectorMatrix(self, w):
        ids = self._subword_ids(w)
        return None if not ids else self._input[torch.tensor(ids, device=self.device)].mean(0)

    def getLabels(self):
        return list(self.labels)

    def getDimension(self):
        return self.dim

    # -- persistence (own container: the Facebook .bin layout is tied to its C++ Args/Dictionary structs)
    def saveModel(self, path):
        torch.save({"args": dict(supervised=self.supervised, skipgram=self.skipgram, cbow=self.cbow, bucket=self.bucket, minCount=self.minCount,
                                 wordNgrams=self.wordNgrams, minNgramLength=self.minn, maxNgramLength=self.maxn, labelPrefix=self.labelPrefix,
                                 dim=self.dim, contextWindowSize=self.window, negativeSamples=self.negative, seed=self.seed),
                    "words": [(w.word, w.elementFrequency) for w in self.vocab.vocabWords()], "labels": self.labels,
                    "input": self._input.cpu(), "output": self._output.cpu()}, os.fspath(path))

    @staticmethod
    def loadModel(path, device=None) -> "FastText":
        d = torch.load(os.fspath(path), weights_only=True)
        m = FastText(device=device, **d["args"])
        m.vocab = AbstractCache()
        for i, (w, f) in enumerate(d["words"]):
            m.vocab.addToken(VocabWord(f, w, i))
        m.vocab.rebuild_index()
        m.labels = list(d["labels"])
        m._label_idx = {l: i for i, l in enumerate(m.labels)}
        m._input, m._output = d["input"].to(m.device), d["output"].to(m.device)
        V = m.vocab.numWords()
        subs = [m._subword_ids(m.vocab.wordAtIndex(i)) for i in range(V)]
        S = max(len(s) for s in subs) if subs else 1
        sub = np.full((V, S), -1, dtype=np.int64)
        for i, s in enumerate(subs):
            sub[i, :len(s)] = s
        m._sub = torch.from_numpy(sub).to(m.device)
        m._finalize()
        return m

    loadBinaryModel = loadModel
