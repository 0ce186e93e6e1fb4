"""InputPreProcessors (reference: deeplearning4j-nn/.../nn/conf/preprocessor/*.java).
Shape adapters inserted automatically between layer families by setInputType()."""
from __future__ import annotations

import torch

from .base import InputType, CNN2DFormat

_P = "org.deeplearning4j.nn.conf.preprocessor."


class InputPreProcessor:
    JAVA_CLASS = None

    def preProcess(self, x, miniBatchSize=None, train=False):
        raise NotImplementedError

    def backprop(self, eps, miniBatchSize=None):
        raise NotImplementedError

    def getOutputType(self, inputType):
        raise NotImplementedError

    def feedForwardMaskArray(self, mask, minibatch):
        return mask

    def to_json(self):
        d = {"@class": self.JAVA_CLASS}
        d.update({k: v for k, v in self.__dict__.items() if not k.startswith("_")})
        return d

    def __eq__(self, o):
        return type(self) is type(o) and self.to_json() == o.to_json()

    def __hash__(self):
        return hash(type(self).__name__)


class CnnToFeedForwardPreProcessor(InputPreProcessor):
    JAVA_CLASS = _P + "CnnToFeedForwardPreProcessor"

    def __init__(self, inputHeight=0, inputWidth=0, numChannels=0, format=CNN2DFormat.NCHW):
        self.inputHeight, self.inputWidth, self.numChannels, self.format = inputHeight, inputWidth, numChannels, format

    def preProcess(self, x, miniBatchSize=None, train=False):
        if x.dim() == 2:
            return x
        self._shape = x.shape
        return x.reshape(x.shape[0], -1)

    def backprop(self, eps, miniBatchSize=None):
        out = eps.reshape(eps.shape[0], self.numChannels, self.inputHeight, self.inputWidth)
        if eps.is_cuda:
            out = out.contiguous(memory_format=torch.channels_last)
        return out

    def getOutputType(self, inputType):
        return InputType.feedForward(inputType.arrayElementsPerExample())


class FeedForwardToCnnPreProcessor(InputPreProcessor):
    JAVA_CLASS = _P + "FeedForwardToCnnPreProcessor"

    def __init__(self, inputHeight=0, inputWidth=0, numChannels=1):
        self.inputHeight, self.inputWidth, self.numChannels = inputHeight, inputWidth, numChannels

    def preProcess(self, x, miniBatchSize=None, train=False):
        if x.dim() == 4:
            return x
        return x.reshape(x.shape[0], self.numChannels, self.inputHeight, self.inputWidth)

    def backprop(self, eps, miniBatchSize=None):
        return eps.reshape(eps.shape[0], -1)

    def getOutputType(self, inputType):
        return InputType.convolutional(self.inputHeight, self.inputWidth, self.numChannels)


class Cnn3DToFeedForwardPreProcessor(InputPreProcessor):
    JAVA_CLASS = _P + "Cnn3DToFeedForwardPreProcessor"

    def __init__(self, inputDepth=0, inputHeight=0, inputWidth=0, numChannels=0, isNCDHW=True):
        self.inputDepth, self.inputHeight, self.inputWidth, self.numChannels, self.isNCDHW = inputDepth, inputHeight, inputWidth, numChannels, isNCDHW

    def preProcess(self, x, miniBatchSize=None, train=False):
        return x.reshape(x.shape[0], -1)

    def backprop(self, eps, miniBatchSize=None):
        return eps.reshape(eps.shape[0], self.numChannels, self.inputDepth, self.inputHeight, self.inputWidth)

    def getOutputType(self, inputType):
        return InputType.feedForward(inputType.arrayElementsPerExample())


class RnnToFeedForwardPreProcessor(InputPreProcessor):
    JAVA_CLASS = _P + "RnnToFeedForwardPreProcessor"

    def __init__(self, rnnDataFormat="NCW"):
        self.rnnDataFormat = rnnDataFormat

    def preProcess(self, x, miniBatchSize=None, train=False):
        # [mb, size, T] -> [mb*T, size]
        self._mb = x.shape[0]
        return x.permute(0, 2, 1).reshape(-1, x.shape[1])

    def backprop(self, eps, miniBatchSize=None):
        mb = miniBatchSize or self._mb
        return eps.reshape(mb, -1, eps.shape[1]).permute(0, 2, 1)

    def getOutputType(self, inputType):
        return InputType.feedForward(inputType.size)

    def feedForwardMaskArray(self, mask, minibatch):
        return None if mask is None else mask.reshape(-1, 1)


class FeedForwardToRnnPreProcessor(InputPreProcessor):
    JAVA_CLASS = _P + "FeedForwardToRnnPreProcessor"

    def __init__(self, rnnDataFormat="NCW"):
        self.rnnDataFormat = rnnDataFormat

    def preProcess(self, x, miniBatchSize=None, train=False):
        mb = miniBatchSize or x.shape[0]
        return x.reshape(mb, -1, x.shape[1]).permute(0, 2, 1)

    def backprop(self, eps, miniBatchSize=None):
        return eps.permute(0, 2, 1).reshape(-1, eps.shape[1])

    def getOutputType(self, inputType):
        return InputType.recurrent(inputType.size)

    def feedForwardMaskArray(self, mask, minibatch):
        return None if mask is None else mask.reshape(minibatch, -1)


class CnnToRnnPreProcessor(InputPreProcessor):
    JAVA_CLASS = _P + "CnnToRnnPreProcessor"

    def __init__(self, inputHeight=0, inputWidth=0, numChannels=0, rnnDataFormat="NCW"):
        self.inputHeight, self.inputWidth, self.numChannels, self.rnnDataFormat = inputHeight, inputWidth, numChannels, rnnDataFormat

    def preProcess(self, x, miniBatchSize=None, train=False):
        # [mb*T, c, h, w] -> [mb, c*h*w, T]
        mb = miniBatchSize
        self._mb = mb
        return x.reshape(mb, -1, self.numChannels * self.inputHeight * self.inputWidth).permute(0, 2, 1)

    def backprop(self, eps, miniBatchSize=None):
        return eps.permute(0, 2, 1).reshape(-1, self.numChannels, self.inputHeight, self.inputWidth)

    def getOutputType(self, inputType):
        return InputType.recurrent(inputType.arrayElementsPerExample())


class RnnToCnnPreProcessor(InputPreProcessor):
    JAVA_CLASS = _P + "RnnToCnnPreProcessor"

    def __init__(self, inputHeight=0, inputWidth=0, numChannels=0, rnnDataFormat="NCW"):
        self.inputHeight, self.inputWidth, self.numChannels, self.rnnDataFormat = inputHeight, inputWidth, numChannels, rnnDataFormat

    def preProcess(self, x, miniBatchSize=None, train=False):
        self._mb = x.shape[0]
        return x.permute(0, 2, 1).reshape(-1, self.numChannels, self.inputHeight, self.inputWidth)

    def backprop(self, eps, miniBatchSize=None):
        mb = miniBatchSize or self._mb
        return eps.reshape(mb, -1, self.numChannels * self.inputHeight * self.inputWidth).permute(0, 2, 1)

    def getOutputType(self, inputType):
        return InputType.convolutional(self.inputHeight, self.inputWidth, self.numChannels)


class ComposableInputPreProcessor(InputPreProcessor):
    JAVA_CLASS = _P + "ComposableInputPreProcessor"

    def __init__(self, *inputPreProcessors):
        if len(inputPreProcessors) == 1 and isinstance(inputPreProcessors[0], (list, tuple)):
            inputPreProcessors = inputPreProcessors[0]
        self.inputPreProcessors = [preprocessor_from_json(p) if isinstance(p, dict) else p for p in inputPreProcessors]

    def preProcess(self, x, miniBatchSize=None, train=False):
        for p in self.inputPreProcessors:
            x = p.preProcess(x, miniBatchSize, train)
        return x

    def backprop(self, eps, miniBatchSize=None):
        for p in reversed(self.inputPreProcessors):
            eps = p.backprop(eps, miniBatchSize)
        return eps

    def getOutputType(self, inputType):
        for p in self.inputPreProcessors:
            inputType = p.getOutputType(inputType)
        return inputType

    def to_json(self):
        return {"@class": self.JAVA_CLASS, "inputPreProcessors": [p.to_json() for p in self.inputPreProcessors]}


_ALL = {c.JAVA_CLASS: c for c in (CnnToFeedForwardPreProcessor, FeedForwardToCnnPreProcessor, Cnn3DToFeedForwardPreProcessor,
                                  RnnToFeedForwardPreProcessor, FeedForwardToRnnPreProcessor, CnnToRnnPreProcessor,
                                  RnnToCnnPreProcessor, ComposableInputPreProcessor)}


def preprocessor_from_json(d):
    if d is None:
        return None
    cls = _ALL[d["@class"]]
    kw = {k: v for k, v in d.items() if k != "@class"}
    if cls is ComposableInputPreProcessor:
        return cls(kw.get("inputPreProcessors", []))
    return cls(**kw)
