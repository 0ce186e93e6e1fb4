"""Sentence / document iterators (reference: deeplearning4j-nlp/.../text/sentenceiterator/** and text/documentiterator/**)."""
from __future__ import annotations

import os
from typing import Callable, Iterable, List, Optional, Sequence


class SentencePreProcessor:
    def preProcess(self, s: str) -> str:
        return s


class SentenceIterator:
    def __init__(self):
        self._pre = None

    def setPreProcessor(self, p):
        self._pre = p

    def getPreProcessor(self):
        return self._pre

    def _pp(self, s):
        if self._pre is None:
            return s
        return self._pre.preProcess(s) if hasattr(self._pre, "preProcess") else self._pre(s)

    def hasNext(self) -> bool:
        raise NotImplementedError

    def nextSentence(self) -> str:
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def finish(self):
        pass

    def __iter__(self):
        self.reset()
        while self.hasNext():
            yield self.nextSentence()


class CollectionSentenceIterator(SentenceIterator):
    def __init__(self, coll: Iterable[str], preProcessor=None):
        super().__init__()
        self._c, self._i, self._pre = list(coll), 0, preProcessor

    def hasNext(self):
        return self._i < len(self._c)

    def nextSentence(self):
        s = self._c[self._i]
        self._i += 1
        return self._pp(s)

    def reset(self):
        self._i = 0


class BasicLineIterator(SentenceIterator):
    """One sentence per line of a text file (BasicLineIterator.java)."""

    def __init__(self, path):
        super().__init__()
        self.path = os.fspath(path)
        self._f, self._next = None, None
        self.reset()

    def _advance(self):
        line = self._f.readline()
        self._next = line.rstrip("\r\n") if line else None

    def hasNext(self):
        return self._next is not None

    def nextSentence(self):
        s = self._next
        self._advance()
        return self._pp(s)

    def reset(self):
        if self._f:
            self._f.close()
        self._f = open(self.path, encoding="utf-8", errors="replace")
        self._advance()

    def finish(self):
        if self._f:
            self._f.close()
            self._f = None


LineSentenceIterator = BasicLineIterator


class FileSentenceIterator(SentenceIterator):
    """Every line of every file under a directory (FileSentenceIterator.java)."""

    def __init__(self, root, preProcessor=None):
        super().__init__()
        self._pre = preProcessor
        root = os.fspath(root)
        if os.path.isdir(root):
            self.files = sorted(os.path.join(d, f) for d, _, fs in os.walk(root) for f in fs)
        else:
            self.files = [root]
        self.reset()

    def _load(self):
        self._lines = []
        while self._fi < len(self.files) and not self._lines:
            with open(self.files[self._fi], encoding="utf-8", errors="replace") as f:
                self._lines = [l.rstrip("\r\n") for l in f if l.strip()]
            self._fi += 1
        self._li = 0

    def hasNext(self):
        return self._li < len(self._lines)

    def nextSentence(self):
        s = self._lines[self._li]
        self._li += 1
        if self._li >= len(self._lines):
            self._load()
        return self._pp(s)

    def reset(self):
        self._fi = 0
        self._load()


class AggregatingSentenceIterator(SentenceIterator):
    def __init__(self, *its):
        super().__init__()
        self.its = list(its)
        self.reset()

    def hasNext(self):
        while self._k < len(self.its) and not self.its[self._k].hasNext():
            self._k += 1
        return self._k < len(self.its)

    def nextSentence(self):
        self.hasNext()
        return self._pp(self.its[self._k].nextSentence())

    def reset(self):
        for i in self.its:
            i.reset()
        self._k = 0


# ----------------------------------------------------------------------------- labelled documents
class LabelledDocument:
    def __init__(self, content="", labels: Optional[Sequence[str]] = None, id=None):
        self.content, self.labels, self.id = content, list(labels or []), id
        self.referencedContent = None

    def getContent(self):
        return self.content

    def getLabels(self):
        return self.labels

    def getLabel(self):
        return self.labels[0] if self.labels else None

    def addLabel(self, l):
        self.labels.append(l)


class LabelsSource:
    """Generates / collects document labels (LabelsSource.java): template ``DOC_%d`` or an explicit list."""

    def __init__(self, template_or_labels="DOC_%d"):
        if isinstance(template_or_labels, str):
            self.template, self.labels, self._explicit = template_or_labels, [], False
        else:
            self.template, self.labels, self._explicit = None, list(template_or_labels), True
        self._counter = 0

    def nextLabel(self):
        if self._explicit:
            l = self.labels[self._counter]
        else:
            l = self.template % self._counter if "%d" in self.template else self.template + str(self._counter)
            self.labels.append(l)
        self._counter += 1
        return l

    def storeLabel(self, l):
        if l not in self.labels:
            self.labels.append(l)

    def getLabels(self):
        return list(self.labels)

    def indexOf(self, l):
        return self.labels.index(l)

    def size(self):
        return len(self.labels)

    def reset(self):
        self._counter = 0


class LabelAwareIterator:
    def hasNextDocument(self) -> bool:
        raise NotImplementedError

    def nextDocument(self) -> LabelledDocument:
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def getLabelsSource(self) -> LabelsSource:
        raise NotImplementedError

    hasNext = lambda self: self.hasNextDocument()
    next = lambda self: self.nextDocument()

    def __iter__(self):
        self.reset()
        while self.hasNextDocument():
            yield self.nextDocument()


class SimpleLabelAwareIterator(LabelAwareIterator):
    def __init__(self, docs: Iterable[LabelledDocument]):
        self.docs, self._i = list(docs), 0
        self.source = LabelsSource([])
        for d in self.docs:
            for l in d.labels:
                self.source.storeLabel(l)

    def hasNextDocument(self):
        return self._i < len(self.docs)

    def nextDocument(self):
        d = self.docs[self._i]
        self._i += 1
        return d

    def reset(self):
        self._i = 0

    def getLabelsSource(self):
        return self.source


class BasicLabelAwareIterator(LabelAwareIterator):
    """Wraps a SentenceIterator, labelling each sentence ``DOC_n`` (BasicLabelAwareIterator.java)."""

    def __init__(self, sentenceIterator: SentenceIterator, labelsSource: Optional[LabelsSource] = None):
        self.it, self.source = sentenceIterator, labelsSource or LabelsSource("DOC_%d")

    def hasNextDocument(self):
        return self.it.hasNext()

    def nextDocument(self):
        return LabelledDocument(self.it.nextSentence(), [self.source.nextLabel()])

    def reset(self):
        self.it.reset()
        self.source.reset()

    def getLabelsSource(self):
        return self.source


class FileLabelAwareIterator(LabelAwareIterator):
    """One document per file; the parent directory name is the label (FileLabelAwareIterator.java)."""

    def __init__(self, *roots):
        self.files = []
        self.source = LabelsSource([])
        for root in roots:
            root = os.fspath(root)
            for lab in sorted(os.listdir(root)):
                d = os.path.join(root, lab)
                if os.path.isdir(d):
                    self.source.storeLabel(lab)
                    self.files += [(os.path.join(d, f), lab) for f in sorted(os.listdir(d)) if os.path.isfile(os.path.join(d, f))]
        self._i = 0

    def hasNextDocument(self):
        return self._i < len(self.files)

    def nextDocument(self):
        p, lab = self.files[self._i]
        self._i += 1
        with open(p, encoding="utf-8", errors="replace") as f:
            return LabelledDocument(f.read(), [lab])

    def reset(self):
        self._i = 0

    def getLabelsSource(self):
        return self.source


class LabelAwareListSentenceIterator(SimpleLabelAwareIterator):
    def __init__(self, sentences: Sequence[str], labels: Sequence[str]):
        super().__init__([LabelledDocument(s, [l]) for s, l in zip(sentences, labels)])
