"""WordVectorSerializer (reference: deeplearning4j-nlp/.../models/embeddings/loader/WordVectorSerializer.java).

Formats: DL4J zip (``syn0.txt`` with ``NUM_WORDS VECTOR_SIZE NUM_DOCS`` header and ``B64:``-encoded labels, ``syn1.txt``,
``syn1Neg.txt``, ``codes.txt``, ``huffman.txt``, ``frequencies.txt``, ``config.json`` [+ ``labels.txt`` for ParagraphVectors],
WordVectorSerializer.java:716-843, 857-965), plain text (``word v1 v2 ...``), and the Google word2vec binary format
(header ``V D\\n`` then ``word<space>`` + D little-endian float32, :133-230)."""
from __future__ import annotations

import base64
import gzip
import io
import json
import os
import struct
import zipfile
from typing import Optional

import numpy as np
import torch

from .embeddings import InMemoryLookupTable, ParagraphVectors, SequenceVectors, Word2Vec, WordVectorsImpl
from .vocab import AbstractCache, VocabWord

B64 = "B64:"


def encodeB64(word: str) -> str:
    return B64 + base64.b64encode(word.encode("utf-8")).decode("ascii")


def decodeB64(word: str) -> str:
    return base64.b64decode(word[len(B64):]).decode("utf-8") if word.startswith(B64) else word


def _rows_txt(m: Optional[torch.Tensor]) -> str:
    if m is None:
        return ""
    a = m.detach().cpu().double().numpy()
    return "\n".join(" ".join(repr(float(v)) for v in row) for row in a) + "\n"


def _open(path, mode="rb"):
    path = os.fspath(path)
    return gzip.open(path, mode) if path.endswith(".gz") else open(path, mode)


class WordVectorSerializer:
    # ------------------------------------------------------------------ text
    @staticmethod
    def writeWordVectors(model_or_table, path, header=True):
        table = model_or_table.lookupTable if hasattr(model_or_table, "lookupTable") else model_or_table
        with _open(path, "wb") as f:
            f.write(WordVectorSerializer._syn0_txt(table, header).encode("utf-8"))

    @staticmethod
    def _syn0_txt(table, header=True) -> str:
        vocab = table.vocab
        syn0 = table.syn0.detach().cpu().double().numpy()
        lines = []
        if header:
            lines.append("%d %d %d" % (vocab.numWords(), table.vectorLength, vocab.totalNumberOfDocs()))
        for i in range(vocab.numWords()):
            w = vocab.elementAtIndex(i)
            lines.append(encodeB64(w.word) + " " + " ".join(repr(float(v)) for v in syn0[i]))
        return "\n".join(lines) + "\n"

    @staticmethod
    def _parse_text(text: str, device=None) -> WordVectorsImpl:
        lines = text.splitlines()
        start = 0
        first = lines[0].split(" ") if lines else []
        ndocs = 0
        if 2 <= len(first) <= 3 and all(t.lstrip("-").isdigit() for t in first):
            start = 1
            ndocs = int(first[2]) if len(first) == 3 else 0
        vocab = AbstractCache()
        vecs = []
        for i, line in enumerate(l for l in lines[start:] if l.strip()):
            parts = line.rstrip().split(" ")
            w = VocabWord(1.0, decodeB64(parts[0]), i)
            vocab.addToken(w)
            vocab.addWordToIndex(i, w.word)
            vecs.append(np.asarray(parts[1:], dtype=np.float64))
        vocab.incrementTotalDocCount(ndocs)
        D = len(vecs[0]) if vecs else 0
        model = WordVectorsImpl()
        model.vocab = vocab
        model.layerSize = D
        model.lookupTable = InMemoryLookupTable(vocab, D, useHS=False, negative=0, device=device)
        model.lookupTable.syn0 = torch.from_numpy(np.stack(vecs)).to(model.lookupTable.device, torch.float32) if vecs else torch.zeros(0, 0)
        return model

    @staticmethod
    def loadTxtVectors(path, device=None) -> WordVectorsImpl:
        with _open(path, "rb") as f:
            return WordVectorSerializer._parse_text(f.read().decode("utf-8"), device)

    loadTxt = loadTxtVectors

    # ------------------------------------------------------------------ google binary
    @staticmethod
    def writeGoogleBinary(model, path):
        t = model.lookupTable
        a = t.syn0.detach().cpu().float().numpy()
        with _open(path, "wb") as f:
            f.write(("%d %d\n" % a.shape).encode())
            for i in range(a.shape[0]):
                f.write(model.vocab.wordAtIndex(i).encode("utf-8") + b" ")
                f.write(a[i].astype("<f4").tobytes())
                f.write(b"\n")

    @staticmethod
    def readBinaryModel(path, linebreaks=None, normalize=False, device=None) -> WordVectorsImpl:
        with _open(path, "rb") as f:
            buf = f.read()
        nl = buf.index(b"\n")
        V, D = (int(x) for x in buf[:nl].split())
        pos = nl + 1
        vocab = AbstractCache()
        mat = np.empty((V, D), dtype=np.float32)
        for i in range(V):
            while buf[pos:pos + 1] in (b"\n", b"\r"):           # optional line break after each vector
                pos += 1
            sp = buf.index(b" ", pos)
            word = buf[pos:sp].decode("utf-8", "replace")
            pos = sp + 1
            mat[i] = np.frombuffer(buf, dtype="<f4", count=D, offset=pos)
            pos += 4 * D
            w = VocabWord(1.0, word, i)
            vocab.addToken(w)
            vocab.addWordToIndex(i, word)
        if normalize:
            mat /= np.maximum(np.linalg.norm(mat, axis=1, keepdims=True), 1e-12)
        model = WordVectorsImpl()
        model.vocab, model.layerSize = vocab, D
        model.lookupTable = InMemoryLookupTable(vocab, D, useHS=False, negative=0, device=device)
        model.lookupTable.syn0 = torch.from_numpy(mat).to(model.lookupTable.device)
        return model

    # ------------------------------------------------------------------ DL4J zip
    @staticmethod
    def _config(model) -> dict:
        return {"minWordFrequency": model.minWordFrequency, "layersSize": model.layerSize, "window": model.window,
                "iterations": model.iterations, "epochs": model.epochs, "learningRate": model.learningRate,
                "minLearningRate": model.minLearningRate, "negative": model.negative, "useHierarchicSoftmax": model.useHS,
                "sampling": model.sampling, "seed": model.seed, "batchSize": model.batchSize,
                "elementsLearningAlgorithm": model.elementsLearningAlgorithm, "sequenceLearningAlgorithm": model.sequenceLearningAlgorithm,
                "trainElementsVectors": model.trainElements, "trainSequenceVectors": model.trainSequences, "useUnknown": model.useUnknown,
                "UNK": model.unknownElement, "stopList": list(model.stopWords), "vocabSize": model.vocab.numWords()}

    @staticmethod
    def writeWord2VecModel(model: SequenceVectors, path):
        t, vocab = model.lookupTable, model.vocab
        ws = [vocab.elementAtIndex(i) for i in range(vocab.numWords())]
        with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
            z.writestr("syn0.txt", WordVectorSerializer._syn0_txt(t))
            z.writestr("syn1.txt", _rows_txt(t.syn1))
            z.writestr("syn1Neg.txt", _rows_txt(t.syn1neg))
            z.writestr("codes.txt", "".join((encodeB64(w.word) + " " + " ".join(str(c) for c in w.codes)).strip() + "\n" for w in ws))
            z.writestr("huffman.txt", "".join((encodeB64(w.word) + " " + " ".join(str(p) for p in w.points)).strip() + "\n" for w in ws))
            z.writestr("frequencies.txt", "".join("%s %s %d\n" % (encodeB64(w.word), repr(float(w.elementFrequency)), w.sequencesCount) for w in ws))
            z.writestr("config.json", json.dumps(WordVectorSerializer._config(model)))
            if isinstance(model, ParagraphVectors) or any(w.isLabel for w in ws):
                z.writestr("labels.txt", "".join(encodeB64(w.word) + "\n" for w in ws if w.isLabel))

    writeParagraphVectors = writeWord2VecModel
    writeSequenceVectors = writeWord2VecModel

    @staticmethod
    def _read_zip(path, cls, device=None):
        with zipfile.ZipFile(path) as z:
            names = set(z.namelist())
            base = WordVectorSerializer._parse_text(z.read("syn0.txt").decode("utf-8"), device)
            cfg = json.loads(z.read("config.json").decode("utf-8")) if "config.json" in names else {}
            kw = dict(minWordFrequency=cfg.get("minWordFrequency", 5), layerSize=cfg.get("layersSize", base.layerSize), windowSize=cfg.get("window", 5),
                      iterations=cfg.get("iterations", 1), epochs=cfg.get("epochs", 1), learningRate=cfg.get("learningRate", 0.025),
                      minLearningRate=cfg.get("minLearningRate", 1e-4), negativeSample=cfg.get("negative", 0.0),
                      useHierarchicSoftmax=cfg.get("useHierarchicSoftmax", True), sampling=cfg.get("sampling", 0.0), seed=cfg.get("seed", 12345),
                      elementsLearningAlgorithm=cfg.get("elementsLearningAlgorithm", "SkipGram"),
                      sequenceLearningAlgorithm=cfg.get("sequenceLearningAlgorithm", "DBOW"), device=device)
            model = cls(**kw)
            model.vocab, model.lookupTable = base.vocab, base.lookupTable
            t = model.lookupTable
            t.useHS, t.negative = model.useHS, model.negative
            model.resetModel = False

            def mat(name):
                if name not in names:
                    return None
                txt = z.read(name).decode("utf-8").strip()
                if not txt:
                    return None
                return torch.from_numpy(np.loadtxt(io.StringIO(txt), dtype=np.float64, ndmin=2)).to(t.device, torch.float32)
            t.syn1, t.syn1neg = mat("syn1.txt"), mat("syn1Neg.txt")
            for name, attr in (("codes.txt", "codes"), ("huffman.txt", "points")):
                if name in names:
                    for line in z.read(name).decode("utf-8").splitlines():
                        parts = line.split(" ")
                        w = model.vocab.wordFor(decodeB64(parts[0]))
                        if w is not None:
                            setattr(w, attr, [int(p) for p in parts[1:] if p])
                            w.codeLength = len(w.codes)
            if "frequencies.txt" in names:
                tot = 0.0
                for line in z.read("frequencies.txt").decode("utf-8").splitlines():
                    parts = line.split(" ")
                    w = model.vocab.wordFor(decodeB64(parts[0]))
                    if w is not None:
                        w.elementFrequency = float(parts[1])
                        w.sequencesCount = int(float(parts[2])) if len(parts) > 2 else 0
                        tot += w.elementFrequency
                model.vocab._total = tot
            if "labels.txt" in names:
                for line in z.read("labels.txt").decode("utf-8").splitlines():
                    w = model.vocab.wordFor(decodeB64(line.strip()))
                    if w is not None:
                        w.isLabel = w.special = True
                model.vocab.updateWordsOccurrences()
            return model

    @staticmethod
    def readWord2VecModel(path, extendedModel=True, device=None):
        """Auto-detects DL4J zip / google binary / text (WordVectorSerializer.readWord2VecModel)."""
        path = os.fspath(path)
        if zipfile.is_zipfile(path):
            return WordVectorSerializer._read_zip(path, Word2Vec, device)
        with _open(path, "rb") as f:
            head = f.read(4096)
        try:
            first = head.split(b"\n", 1)[0].decode("ascii").split()
            rest = head.split(b"\n", 1)[1][:256]
            if len(first) == 2 and all(t.isdigit() for t in first) and any(b > 127 or b < 9 for b in rest):
                return WordVectorSerializer.readBinaryModel(path, device=device)
        except (UnicodeDecodeError, IndexError):
            return WordVectorSerializer.readBinaryModel(path, device=device)
        return WordVectorSerializer.loadTxtVectors(path, device)

    @staticmethod
    def readParagraphVectors(path, device=None) -> ParagraphVectors:
        return WordVectorSerializer._read_zip(path, ParagraphVectors, device)

    @staticmethod
    def readSequenceVectors(path, device=None) -> SequenceVectors:
        return WordVectorSerializer._read_zip(path, SequenceVectors, device)

    @staticmethod
    def loadStaticModel(path, device=None) -> WordVectorsImpl:
        return WordVectorSerializer.readWord2VecModel(path, False, device)

    # ------------------------------------------------------------------ vocab
    @staticmethod
    def writeVocabCache(vocab: AbstractCache, path):
        with open(path, "w", encoding="utf-8") as f:
            for w in vocab.vocabWords():
                f.write(json.dumps({"word": w.word, "index": w.index, "frequency": w.elementFrequency, "codes": w.codes, "points": w.points,
                                    "isLabel": w.isLabel, "sequencesCount": w.sequencesCount}) + "\n")

    @staticmethod
    def readVocabCache(path) -> AbstractCache:
        c = AbstractCache()
        with open(path, encoding="utf-8") as f:
            for line in f:
                d = json.loads(line)
                w = VocabWord(d["frequency"], d["word"], d["index"])
                w.codes, w.points, w.codeLength, w.isLabel, w.sequencesCount = d["codes"], d["points"], len(d["codes"]), d["isLabel"], d["sequencesCount"]
                c.addToken(w)
                c.addWordToIndex(w.index, w.word)
        return c
