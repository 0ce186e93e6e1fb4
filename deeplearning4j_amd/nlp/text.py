"""Text vectorizers, sentence -> CNN input iterator, moving windows, inverted index and read-only word vectors (reference:
deeplearning4j-nlp/src/main/java/org/deeplearning4j/{bagofwords/vectorizer/{BaseTextVectorizer,BagOfWordsVectorizer,TfidfVectorizer}.java,
iterator/CnnSentenceDataSetIterator.java, text/movingwindow/{Window,Windows,WindowConverter,ContextLabelRetriever}.java,
text/invertedindex/InvertedIndex.java, models/word2vec/StaticWord2Vec.java}).

The vectorizers build their vocabulary (with per-word document frequencies) in one pass over a LabelAwareIterator and produce dense
[1, vocab] rows; TF-IDF uses the reference's definition tf = count / documentLength, idf = log10(totalDocs / docsContainingWord).
"""
from __future__ import annotations

import math
import re
from collections import Counter, OrderedDict
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ..datasets.dataset import DataSet
from .iterators import LabelAwareIterator, LabelledDocument, LabelsSource
from .tokenization import DefaultTokenizerFactory, TokenizerFactory
from .vocab import AbstractCache, VocabWord


# ----------------------------------------------------------------------------- bag of words / tf-idf
class BaseTextVectorizer:
    """Vocabulary construction shared by the vectorizers: tokens below ``minWordFrequency`` and stop words are dropped, every kept
    word remembers in how many documents it appeared."""

    def __init__(self, iterator: LabelAwareIterator = None, tokenizerFactory: TokenizerFactory = None, minWordFrequency=1, stopWords=(),
                 labels: Optional[Sequence[str]] = None, vocabCache: Optional[AbstractCache] = None):
        self.iterator, self.tokenizerFactory = iterator, tokenizerFactory or DefaultTokenizerFactory()
        self.minWordFrequency, self.stopWords = int(minWordFrequency), set(stopWords)
        self.vocabCache = vocabCache or AbstractCache()
        self.labelsSource = LabelsSource(list(labels)) if labels is not None else LabelsSource([])
        self._doc_freq: Dict[str, int] = {}
        self._docs = 0
        self._fitted = vocabCache is not None

    class Builder:
        _cls = None

        def __init__(self):
            self._kw = {}

        def setIterator(self, it):
            self._kw["iterator"] = it
            return self

        def setTokenizerFactory(self, tf):
            self._kw["tokenizerFactory"] = tf
            return self

        def setMinWordFrequency(self, n):
            self._kw["minWordFrequency"] = n
            return self

        def setStopWords(self, words):
            self._kw["stopWords"] = words
            return self

        def setVocab(self, cache):
            self._kw["vocabCache"] = cache
            return self

        def setLabels(self, labels):
            self._kw["labels"] = labels
            return self

        def build(self):
            return self._cls(**self._kw)

    def _tokens(self, text: str) -> List[str]:
        return [t for t in self.tokenizerFactory.create(text).getTokens() if t and t not in self.stopWords]

    def buildVocab(self):
        if self.iterator is None:
            raise ValueError("%s: no document iterator set" % type(self).__name__)
        counts, doc_freq, labels = Counter(), Counter(), []
        self.iterator.reset()
        self._docs = 0
        while self.iterator.hasNextDocument():
            doc = self.iterator.nextDocument()
            toks = self._tokens(doc.getContent())
            counts.update(toks)
            doc_freq.update(set(toks))
            self._docs += 1
            for l in doc.getLabels():
                if l not in labels:
                    labels.append(l)
        for w in sorted(counts, key=lambda k: (-counts[k], k)):
            if counts[w] >= self.minWordFrequency:
                vw = VocabWord(float(counts[w]), w)
                self.vocabCache.addToken(vw)
                self.vocabCache.addWordToIndex(self.vocabCache.numWords() - 1, w)
        if hasattr(self.vocabCache, "rebuild_index"):
            self.vocabCache.rebuild_index()
        self.vocabCache._docs = self._docs
        self._doc_freq = {w: int(doc_freq[w]) for w in self.vocabCache.words()}
        if not self.labelsSource.getLabels():
            self.labelsSource = LabelsSource(labels)
        self._fitted = True

    fit = buildVocab

    def getVocabCache(self):
        return self.vocabCache

    def getLabelsSource(self):
        return self.labelsSource

    def docAppearedIn(self, word) -> int:
        return self._doc_freq.get(word, 0)

    def totalNumberOfDocs(self) -> int:
        return self._docs

    def numWords(self):
        return self.vocabCache.numWords()

    # -- to implement
    def _weight(self, word: str, count: int, doc_len: int) -> float:
        raise NotImplementedError

    def transform(self, text_or_tokens) -> torch.Tensor:
        """One document -> [1, vocab] row."""
        if not self._fitted:
            raise RuntimeError("%s: call buildVocab() / fit() first" % type(self).__name__)
        toks = self._tokens(text_or_tokens) if isinstance(text_or_tokens, str) else [t for t in text_or_tokens if t not in self.stopWords]
        row = torch.zeros(1, self.vocabCache.numWords())
        cnt = Counter(t for t in toks if self.vocabCache.containsWord(t))
        for w, c in cnt.items():
            row[0, self.vocabCache.indexOf(w)] = self._weight(w, c, len(toks))
        return row

    def vectorize(self, text: str, label: Optional[str] = None) -> DataSet:
        labels = self.labelsSource.getLabels()
        y = torch.zeros(1, max(len(labels), 1))
        if label is not None:
            if label not in labels:
                raise ValueError("unknown label %r (have %s)" % (label, labels))
            y[0, labels.index(label)] = 1.0
        return DataSet(self.transform(text), y)

    def vectorizeAll(self) -> DataSet:
        """Every document of the iterator as one DataSet (features [docs, vocab], one-hot labels)."""
        xs, ys = [], []
        self.iterator.reset()
        while self.iterator.hasNextDocument():
            d = self.iterator.nextDocument()
            ds = self.vectorize(d.getContent(), d.getLabel())
            xs.append(ds.features)
            ys.append(ds.labels)
        return DataSet(torch.cat(xs), torch.cat(ys))


class BagOfWordsVectorizer(BaseTextVectorizer):
    """Word counts."""

    def _weight(self, word, count, doc_len):
        return float(count)

    class Builder(BaseTextVectorizer.Builder):
        pass


BagOfWordsVectorizer.Builder._cls = BagOfWordsVectorizer


class TfidfVectorizer(BaseTextVectorizer):
    def tfForWord(self, count, doc_len):
        return count / doc_len if doc_len else 0.0

    def idfForWord(self, word):
        df = self.docAppearedIn(word)
        return math.log10(self._docs / df) if (self._docs > 0 and df > 0) else 0.0

    def tfidfWord(self, word, count, doc_len):
        return self.tfForWord(count, doc_len) * self.idfForWord(word)

    def _weight(self, word, count, doc_len):
        return self.tfidfWord(word, count, doc_len)

    class Builder(BaseTextVectorizer.Builder):
        pass


TfidfVectorizer.Builder._cls = TfidfVectorizer


# ----------------------------------------------------------------------------- CnnSentenceDataSetIterator
class UnknownWordHandling:
    RemoveWord, UseUnknownVector = "RemoveWord", "UseUnknownVector"


class Format:
    CNN1D, CNN2D = "CNN1D", "CNN2D"


class CnnSentenceDataSetIterator:
    """Sentences -> stacked word vectors for convolutional text classifiers. CNN2D: features [mb, 1, maxLen, vecSize] (sentencesAlongHeight)
    or [mb, 1, vecSize, maxLen]; CNN1D: [mb, vecSize, maxLen]; features mask [mb, maxLen] marks real tokens; labels one-hot."""

    def __init__(self, sentenceProvider, wordVectors, tokenizerFactory=None, unknownWordHandling=UnknownWordHandling.RemoveWord, minibatchSize=32,
                 maxSentenceLength=-1, sentencesAlongHeight=True, useNormalizedWordVectors=True, format=Format.CNN2D, dataSetPreProcessor=None):
        if sentenceProvider is None or wordVectors is None:
            raise ValueError("CnnSentenceDataSetIterator needs a sentence provider and word vectors")
        self.provider, self.wv = sentenceProvider, wordVectors
        self.tf = tokenizerFactory or DefaultTokenizerFactory()
        self.unknown, self.mb, self.maxLen = unknownWordHandling, int(minibatchSize), int(maxSentenceLength)
        self.alongHeight, self.normalized, self.format, self.pre = sentencesAlongHeight, useNormalizedWordVectors, format, dataSetPreProcessor
        self.labels = sorted(self.provider.allLabels())
        self.vecSize = int(self.wv.lookupTable_().layerSize())
        self._unk = None

    class Builder:
        def __init__(self, format=Format.CNN2D):
            self._kw = {"format": format}

        def __getattr__(self, name):
            def setter(v, *more):
                self._kw[name] = v
                return self
            return setter

        def build(self):
            return CnnSentenceDataSetIterator(**self._kw)

    def getLabels(self):
        return list(self.labels)

    def _vector(self, w):
        m = self.wv.getWordVectorMatrixNormalized(w) if self.normalized else self.wv.getWordVectorMatrix(w)
        return torch.as_tensor(getattr(m, "tensor", lambda: m)() if callable(getattr(m, "tensor", None)) else m).float().reshape(-1).cpu()

    def _unknown_vector(self):
        if self._unk is None:
            w = self.wv.getWeights() if hasattr(self.wv, "getWeights") else self.wv.lookupTable_().getWeights()
            w = torch.as_tensor(getattr(w, "tensor", lambda: w)() if callable(getattr(w, "tensor", None)) else w).float().cpu()
            self._unk = w.mean(0)
        return self._unk

    def _tokens(self, sentence):
        out = []
        for t in self.tf.create(sentence).getTokens():
            if self.wv.hasWord(t):
                out.append(t)
            elif self.unknown == UnknownWordHandling.UseUnknownVector:
                out.append(None)
        return out[:self.maxLen] if self.maxLen > 0 else out

    def loadSingleSentence(self, sentence) -> torch.Tensor:
        toks = self._tokens(sentence)
        if not toks:
            raise ValueError("no known words in sentence %r" % sentence)
        return self._shape(torch.stack([self._unknown_vector() if t is None else self._vector(t) for t in toks]).unsqueeze(0))

    def _shape(self, x):          # x [mb, len, vec]
        if self.format == Format.CNN1D:
            return x.permute(0, 2, 1).contiguous()
        return (x if self.alongHeight else x.permute(0, 2, 1)).unsqueeze(1).contiguous()

    def hasNext(self):
        return self.provider.hasNext()

    def reset(self):
        self.provider.reset()

    def next(self, num=None) -> DataSet:
        if not self.hasNext():
            raise StopIteration
        rows, labs = [], []
        while self.provider.hasNext() and len(rows) < (num or self.mb):
            s, l = self.provider.nextSentence()
            toks = self._tokens(s)
            if not toks:
                continue                        # nothing usable in this sentence (CnnSentenceDataSetIterator skips them too)
            rows.append(toks)
            labs.append(l)
        if not rows:
            raise StopIteration
        L_ = max(len(r) for r in rows)
        x = torch.zeros(len(rows), L_, self.vecSize)
        mask = torch.zeros(len(rows), L_)
        y = torch.zeros(len(rows), len(self.labels))
        for i, (toks, l) in enumerate(zip(rows, labs)):
            for j, t in enumerate(toks):
                x[i, j] = self._unknown_vector() if t is None else self._vector(t)
            mask[i, :len(toks)] = 1.0
            y[i, self.labels.index(l)] = 1.0
        ds = DataSet(self._shape(x), y, featuresMask=mask)
        if self.pre is not None:
            self.pre.preProcess(ds)
        return ds

    def __iter__(self):
        self.reset()
        while self.hasNext():
            try:
                yield self.next()
            except StopIteration:
                return

    def inputColumns(self):
        return self.vecSize

    def totalOutcomes(self):
        return len(self.labels)

    def batch(self):
        return self.mb


# ----------------------------------------------------------------------------- moving windows
class Window:
    """A window of tokens around a focus word, padded with <s> / </s> at the sentence borders (Window.java)."""

    def __init__(self, words: Sequence[str], windowSize: int, begin: int, end: int, label: str = "NONE"):
        self.words, self.windowSize, self.begin, self.end, self.label = list(words), windowSize, begin, end, label
        self.median = len(self.words) // 2

    def getWords(self):
        return self.words

    def getFocusWord(self):
        return self.words[self.median]

    def getLabel(self):
        return self.label

    def setLabel(self, l):
        self.label = l

    def isBeginLabel(self):
        return self.words[0] == "<s>"

    def isEndLabel(self):
        return self.words[-1] == "</s>"

    def asTokens(self):
        return " ".join(self.words)

    def __repr__(self):
        return "Window(%s, focus=%s, label=%s)" % (self.words, self.getFocusWord(), self.label)


class Windows:
    @staticmethod
    def windowForWordInPosition(windowSize, wordPos, sentence: Sequence[str]) -> Window:
        half = windowSize // 2
        ctx = []
        for i in range(wordPos - half, wordPos + half + 1):
            ctx.append("<s>" if i < 0 else ("</s>" if i >= len(sentence) else sentence[i]))
        return Window(ctx, windowSize, max(wordPos - half, 0), min(wordPos + half, len(sentence) - 1))

    @staticmethod
    def windows(words, windowSize=5, tokenizerFactory: Optional[TokenizerFactory] = None) -> List[Window]:
        toks = (tokenizerFactory or DefaultTokenizerFactory()).create(words).getTokens() if isinstance(words, str) else list(words)
        return [Windows.windowForWordInPosition(windowSize, i, toks) for i in range(len(toks))]


class WindowConverter:
    """Window -> concatenated word vectors (WindowConverter.asExampleArray / asExampleMatrix)."""

    @staticmethod
    def asExampleArray(window: Window, vec, normalize=False) -> torch.Tensor:
        size = int(vec.lookupTable_().layerSize())
        parts = []
        for w in window.getWords():
            if vec.hasWord(w):
                m = vec.getWordVectorMatrixNormalized(w) if normalize else vec.getWordVectorMatrix(w)
                t = getattr(m, "tensor", None)
                parts.append(torch.as_tensor(t() if callable(t) else m).float().reshape(-1).cpu())
            else:
                parts.append(torch.zeros(size))
        return torch.cat(parts).unsqueeze(0)

    @staticmethod
    def asExampleMatrix(window: Window, vec) -> torch.Tensor:
        return WindowConverter.asExampleArray(window, vec).reshape(len(window.getWords()), -1)


class ContextLabelRetriever:
    """``"some <LABEL> tagged words </LABEL> here"`` -> (plain sentence, {span text: label}) (ContextLabelRetriever.stringWithLabels)."""
    _BEGIN = re.compile(r"<([A-Za-z]+|\d+)>$")
    _END = re.compile(r"</([A-Za-z]+|\d+)>$")

    @staticmethod
    def stringWithLabels(sentence: str, tokenizerFactory: Optional[TokenizerFactory] = None) -> Tuple[str, "OrderedDict[str, str]"]:
        toks = sentence.split() if tokenizerFactory is None else tokenizerFactory.create(sentence).getTokens()
        plain, spans, cur, label = [], OrderedDict(), [], None
        for t in toks:
            b, e = ContextLabelRetriever._BEGIN.match(t), ContextLabelRetriever._END.match(t)
            if b:
                if label is not None:
                    raise ValueError("nested label <%s> inside <%s>" % (b.group(1), label))
                label, cur = b.group(1), []
            elif e:
                if label is None or e.group(1) != label:
                    raise ValueError("closing label </%s> does not match <%s>" % (e.group(1), label))
                spans[" ".join(cur)] = label
                label = None
            else:
                plain.append(t)
                if label is not None:
                    cur.append(t)
        if label is not None:
            raise ValueError("unclosed label <%s>" % label)
        return " ".join(plain), spans


# ----------------------------------------------------------------------------- inverted index
class InvertedIndex:
    """In-memory inverted index over tokenised documents (InvertedIndex.java API: addWordsToDoc, documents(word), document(i), ...)."""

    def __init__(self):
        self._docs: "OrderedDict[int, Tuple[List[str], List[str]]]" = OrderedDict()
        self._postings: Dict[str, List[int]] = {}

    def addWordToDoc(self, doc: int, word):
        w = word.word if isinstance(word, VocabWord) else word
        self._docs.setdefault(doc, ([], []))[0].append(w)
        p = self._postings.setdefault(w, [])
        if not p or p[-1] != doc:
            p.append(doc)

    def addWordsToDoc(self, doc: int, words: Iterable, label=None):
        for w in words:
            self.addWordToDoc(doc, w)
        if label is not None:
            self.addLabelForDoc(doc, label)

    def addLabelForDoc(self, doc: int, label):
        self._docs.setdefault(doc, ([], []))[1].append(label.word if isinstance(label, VocabWord) else label)

    def document(self, i) -> List[str]:
        return list(self._docs[i][0])

    def documentWithLabels(self, i):
        return list(self._docs[i][0]), list(self._docs[i][1])

    def documents(self, word) -> List[int]:
        return list(self._postings.get(word.word if isinstance(word, VocabWord) else word, []))

    def numDocuments(self):
        return len(self._docs)

    def allDocs(self):
        return list(self._docs)

    def totalWords(self):
        return sum(len(d[0]) for d in self._docs.values())

    def docs(self):
        for i, (w, _) in self._docs.items():
            yield list(w)

    def eachDoc(self, fn):
        for i in self._docs:
            fn(self.document(i))

    def batchIter(self, batchSize):
        cur = []
        for i in self._docs:
            cur.append(self.document(i))
            if len(cur) == batchSize:
                yield cur
                cur = []
        if cur:
            yield cur


# ----------------------------------------------------------------------------- StaticWord2Vec
class StaticWord2Vec:
    """Read-only word vectors for inference-only deployments (StaticWord2Vec.java): vectors stay in a (memory-mapped) host matrix and
    only the rows that are asked for travel to the device, through a small LRU cache per device."""

    def __init__(self, vocab: AbstractCache, matrix, cachePerDevice=1000, device=None):
        self.vocab = vocab
        self._m = matrix if isinstance(matrix, np.ndarray) else np.asarray(matrix)
        self.cachePerDevice, self.device = int(cachePerDevice), device
        self._cache: "OrderedDict[int, torch.Tensor]" = OrderedDict()

    @staticmethod
    def fromModel(wv, cachePerDevice=1000, device=None) -> "StaticWord2Vec":
        w = wv.lookupTable_().getWeights()
        t = getattr(w, "tensor", None)
        return StaticWord2Vec(wv.getVocab(), torch.as_tensor(t() if callable(t) else w).detach().float().cpu().numpy(), cachePerDevice, device)

    @staticmethod
    def load(vocab: AbstractCache, npy_path, cachePerDevice=1000, device=None) -> "StaticWord2Vec":
        return StaticWord2Vec(vocab, np.load(npy_path, mmap_mode="r"), cachePerDevice, device)

    def save(self, npy_path):
        np.save(npy_path, np.asarray(self._m))

    def hasWord(self, w):
        return self.vocab.containsWord(w)

    def indexOf(self, w):
        return self.vocab.indexOf(w)

    def getWordVectorMatrix(self, w) -> Optional[torch.Tensor]:
        i = self.vocab.indexOf(w)
        if i < 0:
            return None
        v = self._cache.get(i)
        if v is None:
            v = torch.from_numpy(np.array(self._m[i], dtype=np.float32))
            if self.device is not None:
                v = v.to(self.device)
            if self.cachePerDevice > 0:
                self._cache[i] = v
                if len(self._cache) > self.cachePerDevice:
                    self._cache.popitem(last=False)
        else:
            self._cache.move_to_end(i)
        return v

    getWordVector = getWordVectorMatrix

    def getWordVectors(self, words):
        vs = [self.getWordVectorMatrix(w) for w in words if self.hasWord(w)]
        return torch.stack(vs) if vs else torch.zeros(0, self._m.shape[1])

    def getWordVectorsMean(self, words):
        vs = self.getWordVectors(words)
        return vs.mean(0) if vs.numel() else torch.zeros(self._m.shape[1])

    def similarity(self, a, b) -> float:
        va, vb = self.getWordVectorMatrix(a), self.getWordVectorMatrix(b)
        if va is None or vb is None:
            return float("nan")
        return float(torch.dot(va, vb) / (va.norm() * vb.norm()).clamp_min(1e-12))

    def wordsNearest(self, *a, **k):
        raise NotImplementedError("StaticWord2Vec is a lookup-only model (as in the reference): use the full model for nearest-word queries")
