"""Vocabulary (reference: deeplearning4j-nlp/.../models/word2vec/{VocabWord,Huffman}.java,
models/word2vec/wordstore/{VocabCache,VocabConstructor}.java, wordstore/inmemory/AbstractCache.java)."""
from __future__ import annotations

import heapq
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

MAX_CODE_LENGTH = 40


class VocabWord:
    __slots__ = ("word", "index", "elementFrequency", "codes", "points", "codeLength", "isLabel", "special", "sequencesCount")

    def __init__(self, elementFrequency=1.0, word="", index=-1):
        self.word, self.index, self.elementFrequency = word, index, float(elementFrequency)
        self.codes: List[int] = []
        self.points: List[int] = []
        self.codeLength = 0
        self.isLabel = False
        self.special = False
        self.sequencesCount = 0

    def getWord(self):
        return self.word

    getLabel = getWord

    def getIndex(self):
        return self.index

    def getElementFrequency(self):
        return self.elementFrequency

    def increaseElementFrequency(self, by=1):
        self.elementFrequency += by

    def markAsLabel(self, b=True):
        self.isLabel = b

    def __repr__(self):
        return "VocabWord(%r, idx=%d, freq=%g)" % (self.word, self.index, self.elementFrequency)


class AbstractCache:
    """In-memory VocabCache."""

    def __init__(self):
        self._by_word: Dict[str, VocabWord] = {}
        self._by_index: List[VocabWord] = []
        self._total = 0.0
        self._docs = 0
        self.stopWords: set = set()

    # -- mutation
    def addToken(self, w: VocabWord):
        cur = self._by_word.get(w.word)
        if cur is not None:
            cur.elementFrequency += w.elementFrequency
            self._total += w.elementFrequency
            return cur
        self._by_word[w.word] = w
        self._total += w.elementFrequency
        return w

    def addWordToIndex(self, index, word):
        w = self._by_word[word]
        w.index = index
        while len(self._by_index) <= index:
            self._by_index.append(None)
        self._by_index[index] = w

    def putVocabWord(self, word):
        pass

    def incrementWordCount(self, word, by=1):
        w = self._by_word.get(word)
        if w is None:
            w = VocabWord(0.0, word)
            self._by_word[word] = w
        w.elementFrequency += by
        self._total += by

    def removeElement(self, word):
        w = self._by_word.pop(word if isinstance(word, str) else word.word, None)
        if w is not None:
            self._total -= w.elementFrequency

    def updateWordsOccurrences(self):
        self._total = sum(w.elementFrequency for w in self._by_word.values() if not w.isLabel)

    def rebuild_index(self):
        """Index by descending frequency (labels last), as VocabConstructor does after filtering."""
        ws = sorted(self._by_word.values(), key=lambda w: (w.isLabel, -w.elementFrequency, w.word))
        self._by_index = ws
        for i, w in enumerate(ws):
            w.index = i

    # -- queries
    def containsWord(self, word):
        return word in self._by_word

    hasToken = containsWord

    def wordFor(self, word) -> Optional[VocabWord]:
        return self._by_word.get(word)

    tokenFor = wordFor

    def wordAtIndex(self, i) -> Optional[str]:
        return self._by_index[i].word if 0 <= i < len(self._by_index) else None

    def elementAtIndex(self, i) -> Optional[VocabWord]:
        return self._by_index[i] if 0 <= i < len(self._by_index) else None

    def indexOf(self, word) -> int:
        w = self._by_word.get(word)
        return w.index if w is not None else -1

    def wordFrequency(self, word) -> int:
        w = self._by_word.get(word)
        return int(w.elementFrequency) if w else 0

    def numWords(self):
        return len(self._by_word)

    def words(self) -> List[str]:
        return [w.word for w in self._by_index] if self._by_index else list(self._by_word)

    def vocabWords(self) -> List[VocabWord]:
        return list(self._by_index) if self._by_index else list(self._by_word.values())

    tokens = vocabWords

    def totalWordOccurrences(self):
        return int(self._total)

    def totalNumberOfDocs(self):
        return self._docs

    def incrementTotalDocCount(self, by=1):
        self._docs += by

    def docAppearedIn(self, word):
        w = self._by_word.get(word)
        return w.sequencesCount if w else 0

    def incrementDocCount(self, word, by=1):
        w = self._by_word.get(word)
        if w:
            w.sequencesCount += by

    def importVocabulary(self, other: "AbstractCache"):
        for w in other.vocabWords():
            self.addToken(w)


InMemoryLookupCache = AbstractCache


class Huffman:
    """word2vec binary Huffman tree: assigns ``codes`` (0/1 branch bits) and ``points`` (inner-node rows of syn1), root first
    (Huffman.java:57-150)."""

    def __init__(self, words: Sequence[VocabWord]):
        self.words = list(words)

    def build(self):
        n = len(self.words)
        if n == 0:
            return
        if n == 1:
            w = self.words[0]
            w.codes, w.points, w.codeLength = [0], [0], 1
            return
        order = sorted(range(n), key=lambda i: -self.words[i].elementFrequency)
        count = np.empty(2 * n + 1, dtype=np.float64)
        count[:n] = [self.words[i].elementFrequency for i in order]
        count[n:] = 1e15
        binary = np.zeros(2 * n + 1, dtype=np.int8)
        parent = np.zeros(2 * n + 1, dtype=np.int64)
        pos1, pos2 = n - 1, n
        for a in range(n - 1):
            mins = []
            for _ in range(2):
                if pos1 >= 0 and count[pos1] < count[pos2]:
                    mins.append(pos1); pos1 -= 1
                else:
                    mins.append(pos2); pos2 += 1
            count[n + a] = count[mins[0]] + count[mins[1]]
            parent[mins[0]] = parent[mins[1]] = n + a
            binary[mins[1]] = 1
        for a in range(n):
            code, point = [], []
            b = a
            while b != 2 * n - 2:
                code.append(int(binary[b]))
                point.append(b)
                b = int(parent[b])
            w = self.words[order[a]]
            w.codeLength = min(len(code), MAX_CODE_LENGTH)
            w.codes = code[::-1][:MAX_CODE_LENGTH]
            pts = [n - 2] + [p - n for p in point[::-1][:-1]]
            w.points = pts[:MAX_CODE_LENGTH]

    def applyIndexes(self, cache: AbstractCache):
        for i, w in enumerate(sorted(self.words, key=lambda w: (w.isLabel, -w.elementFrequency, w.word))):
            w.index = i
        cache.rebuild_index()


class VocabConstructor:
    """Builds a vocabulary from tokenised sequences (VocabConstructor.java: counting, minWordFrequency filter, optional
    Huffman tree)."""

    def __init__(self, minWordFrequency=1, stopWords: Iterable[str] = (), useHuffman=True, cache: Optional[AbstractCache] = None,
                 limit=0):
        self.minWordFrequency, self.stopWords, self.useHuffman = minWordFrequency, set(stopWords), useHuffman
        self.cache = cache or AbstractCache()
        self.limit = limit

    def buildJointVocabulary(self, sequences: Iterable[Sequence[str]], labels_of=None) -> AbstractCache:
        counts: Dict[str, int] = {}
        docs: Dict[str, int] = {}
        label_set: Dict[str, int] = {}
        ndocs = 0
        for seq in sequences:
            labels = ()
            if isinstance(seq, tuple) and len(seq) == 2 and not isinstance(seq[0], str):
                seq, labels = seq
            ndocs += 1
            for t in seq:
                if t and t not in self.stopWords:
                    counts[t] = counts.get(t, 0) + 1
            for t in set(seq):
                docs[t] = docs.get(t, 0) + 1
            for l in labels:
                label_set[l] = label_set.get(l, 0) + 1
        c = self.cache
        items = [(w, n) for w, n in counts.items() if n >= self.minWordFrequency]
        if self.limit and len(items) > self.limit:
            items = heapq.nlargest(self.limit, items, key=lambda kv: kv[1])
        for w, n in items:
            vw = c.addToken(VocabWord(n, w))
            vw.sequencesCount = docs.get(w, 0)
        for l, n in label_set.items():
            vw = c.wordFor(l)
            if vw is None:
                vw = c.addToken(VocabWord(n, l))
                c._total -= n
            vw.isLabel = True
            vw.special = True
        c.incrementTotalDocCount(ndocs)
        c.rebuild_index()
        if self.useHuffman:
            Huffman(c.vocabWords()).build()
        return c
