"""Tokenizers and token pre-processors (reference: deeplearning4j-nlp/.../text/tokenization/**:
tokenizer/{DefaultTokenizer,NGramTokenizer,BertWordPieceTokenizer}.java, tokenizerfactory/*, tokenizer/preprocessor/*)."""
from __future__ import annotations

import re
import unicodedata
from typing import Dict, Iterable, List, Optional


# ----------------------------------------------------------------------------- pre-processors
class TokenPreProcess:
    def preProcess(self, token: str) -> str:
        raise NotImplementedError


class CommonPreprocessor(TokenPreProcess):
    """Lower-cases and strips punctuation / digits (CommonPreprocessor.java: ``[\\d\\.:,"'\\(\\)\\[\\]|/?!;]+``)."""
    _P = re.compile(r"[\d\.:,\"'\(\)\[\]|/?!;]+")

    def preProcess(self, token):
        return self._P.sub("", token).lower()


class LowCasePreProcessor(TokenPreProcess):
    def preProcess(self, token):
        return token.lower()


class EndingPreProcessor(TokenPreProcess):
    """Crude stemmer: strips s / ed / ing / ly / . endings (EndingPreProcessor.java)."""

    def preProcess(self, token):
        for suf in ("s", "."):
            if token.endswith(suf) and not token.endswith("ss"):
                token = token[:-1]
        for suf in ("ed", "ing", "ly"):
            if token.endswith(suf):
                token = token[:-len(suf)]
        return token


class StringCleaning:
    @staticmethod
    def stripPunct(s):
        return re.sub(r"[^\w\s]", "", s)


class CompositePreProcessor(TokenPreProcess):
    def __init__(self, *ps):
        self.ps = ps

    def preProcess(self, token):
        for p in self.ps:
            token = p.preProcess(token)
        return token


class BertWordPiecePreProcessor(TokenPreProcess):
    """Lower-case + accent stripping + control-char cleanup (BertWordPiecePreProcessor.java)."""

    def __init__(self, lowerCase=False, stripAccents=False, vocab=None):
        self.lowerCase, self.stripAccents, self.vocab = lowerCase, stripAccents, vocab

    def preProcess(self, token):
        out = []
        if self.stripAccents:
            token = unicodedata.normalize("NFD", token)
        for ch in token:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or (unicodedata.category(ch) in ("Cc", "Cf") and ch not in "\t\n\r"):
                continue
            if self.stripAccents and unicodedata.category(ch) == "Mn":
                continue
            if ch in "\t\n\r" or unicodedata.category(ch) == "Zs":
                out.append(" ")
            else:
                out.append(ch.lower() if self.lowerCase else ch)
        return "".join(out)


# ----------------------------------------------------------------------------- tokenizers
class Tokenizer:
    def __init__(self, tokens: List[str], pre: Optional[TokenPreProcess] = None):
        self._tokens, self._pre, self._pos = tokens, pre, 0

    def setTokenPreProcessor(self, p):
        self._pre = p

    def hasMoreTokens(self):
        return self._pos < len(self._tokens)

    def countTokens(self):
        return len(self._tokens)

    def nextToken(self):
        t = self._tokens[self._pos]
        self._pos += 1
        return self._pre.preProcess(t) if self._pre else t

    def getTokens(self) -> List[str]:
        out = []
        for t in self._tokens:
            t = self._pre.preProcess(t) if self._pre else t
            if t:
                out.append(t)
        return out

    def __iter__(self):
        return iter(self.getTokens())


class TokenizerFactory:
    def __init__(self):
        self._pre = None

    def setTokenPreProcessor(self, p):
        self._pre = p
        return self

    def getTokenPreProcessor(self):
        return self._pre

    def create(self, text: str) -> Tokenizer:
        raise NotImplementedError


class DefaultTokenizerFactory(TokenizerFactory):
    """Whitespace tokenizer (java.util.StringTokenizer semantics: split on space/tab/newline/CR/FF)."""

    def create(self, text):
        return Tokenizer(text.split(), self._pre)


class NGramTokenizerFactory(TokenizerFactory):
    def __init__(self, base: TokenizerFactory, minN: int, maxN: int):
        super().__init__()
        self.base, self.minN, self.maxN = base, minN, maxN

    def create(self, text):
        toks = self.base.create(text).getTokens()
        out = list(toks) if self.minN <= 1 else []
        for n in range(max(2, self.minN), self.maxN + 1):
            out += [" ".join(toks[i:i + n]) for i in range(len(toks) - n + 1)]
        return Tokenizer(out, self._pre)


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or
            0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class BertWordPieceTokenizer(Tokenizer):
    """Greedy longest-match-first WordPiece over a basic (whitespace + punctuation split) tokenisation
    (BertWordPieceTokenizer.java:73-140)."""

    def __init__(self, text, vocab: Dict[str, int], pre=None, unk="[UNK]", maxCharsPerWord=100):
        if pre is not None:
            text = pre.preProcess(text)
        toks = []
        for basic in self._basic(text):
            toks += self._wordpiece(basic, vocab, unk, maxCharsPerWord)
        super().__init__(toks, None)

    @staticmethod
    def _basic(text):
        out, cur = [], []
        for ch in text:
            if ch.isspace():
                if cur:
                    out.append("".join(cur)); cur = []
            elif _is_punct(ch) or _is_cjk(ord(ch)):
                if cur:
                    out.append("".join(cur)); cur = []
                out.append(ch)
            else:
                cur.append(ch)
        if cur:
            out.append("".join(cur))
        return out

    @staticmethod
    def _wordpiece(word, vocab, unk, maxChars):
        if len(word) > maxChars:
            return [unk]
        out, start = [], 0
        while start < len(word):
            end, piece = len(word), None
            while start < end:
                sub = word[start:end]
                if start > 0:
                    sub = "##" + sub
                if sub in vocab:
                    piece = sub
                    break
                end -= 1
            if piece is None:
                return [unk]
            out.append(piece)
            start = end
        return out


class BertWordPieceTokenizerFactory(TokenizerFactory):
    def __init__(self, vocab, lowerCaseOnly=False, stripAccents=False, encoding="utf-8"):
        super().__init__()
        if isinstance(vocab, (str, bytes)) or hasattr(vocab, "__fspath__"):
            vocab = self.loadVocab(vocab, encoding)
        elif not isinstance(vocab, dict):
            vocab = {t: i for i, t in enumerate(vocab)}
        self.vocab: Dict[str, int] = vocab
        self._pre = BertWordPiecePreProcessor(lowerCaseOnly, stripAccents, vocab)

    @staticmethod
    def loadVocab(path, encoding="utf-8") -> Dict[str, int]:
        vocab = {}
        with open(path, encoding=encoding) as f:
            for i, line in enumerate(f):
                vocab[line.rstrip("\n")] = i
        return vocab

    def getVocab(self):
        return self.vocab

    def create(self, text):
        return BertWordPieceTokenizer(text, self.vocab, self._pre)


# ----------------------------------------------------------------------------- stop words
class StopWords:
    _WORDS = ("a about above after again against all am an and any are aren't as at be because been before being below between both but by "
              "can't cannot could couldn't did didn't do does doesn't doing don't down during each few for from further had hadn't has hasn't "
              "have haven't having he he'd he'll he's her here here's hers herself him himself his how how's i i'd i'll i'm i've if in into is "
              "isn't it it's its itself let's me more most mustn't my myself no nor not of off on once only or other ought our ours ourselves "
              "out over own same shan't she she'd she'll she's should shouldn't so some such than that that's the their theirs them themselves "
              "then there there's these they they'd they'll they're they've this those through to too under until up very was wasn't we we'd "
              "we'll we're we've were weren't what what's when when's where where's which while who who's whom why why's with won't would "
              "wouldn't you you'd you'll you're you've your yours yourself yourselves").split()

    @staticmethod
    def getStopWords():
        return list(StopWords._WORDS)
