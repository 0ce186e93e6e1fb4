"""Writable value types (reference: datavec-api/.../writable/*.java)."""
from __future__ import annotations

import math

import numpy as np


class Writable:
    __slots__ = ("value",)
    TYPE = "Writable"

    def __init__(self, value=None):
        self.value = value

    def get(self):
        return self.value

    def toDouble(self):
        return float(self.value)

    def toFloat(self):
        return float(self.value)

    def toInt(self):
        return int(float(self.value))

    def toLong(self):
        return int(float(self.value))

    def toString(self):
        return str(self.value)

    def getType(self):
        return self.TYPE

    def __str__(self):
        return self.toString()

    def __repr__(self):
        return "%s(%r)" % (type(self).__name__, self.value)

    def __eq__(self, o):
        return isinstance(o, Writable) and type(o) is type(self) and o.value == self.value

    def __hash__(self):
        return hash((type(self).__name__, self.value))


class IntWritable(Writable):
    TYPE = "Int"

    def __init__(self, v=0):
        super().__init__(int(v))


class LongWritable(IntWritable):
    TYPE = "Long"


class FloatWritable(Writable):
    TYPE = "Float"

    def __init__(self, v=0.0):
        super().__init__(float(np.float32(v)))


class DoubleWritable(Writable):
    TYPE = "Double"

    def __init__(self, v=0.0):
        super().__init__(float(v))


class BooleanWritable(Writable):
    TYPE = "Boolean"

    def __init__(self, v=False):
        super().__init__(bool(v))

    def toDouble(self):
        return 1.0 if self.value else 0.0

    def toInt(self):
        return 1 if self.value else 0


class Text(Writable):
    TYPE = "Text"

    def __init__(self, v=""):
        super().__init__(str(v))

    def toDouble(self):
        return float(self.value)

    def toInt(self):
        return int(float(self.value))


class NullWritable(Writable):
    TYPE = "Null"

    def toDouble(self):
        return float("nan")

    def toString(self):
        return ""


class BytesWritable(Writable):
    TYPE = "Bytes"


class NDArrayWritable(Writable):
    TYPE = "NDArray"

    def __eq__(self, o):
        import torch
        return isinstance(o, NDArrayWritable) and torch.equal(torch.as_tensor(self.value), torch.as_tensor(o.value))

    __hash__ = None

    def length(self):
        import torch
        return int(torch.as_tensor(self.value).numel())


def wrap(v) -> Writable:
    """Python value -> Writable."""
    if isinstance(v, Writable):
        return v
    if isinstance(v, bool):
        return BooleanWritable(v)
    if isinstance(v, (int, np.integer)):
        return IntWritable(v) if -2 ** 31 <= int(v) < 2 ** 31 else LongWritable(v)
    if isinstance(v, (float, np.floating)):
        return DoubleWritable(v)
    if v is None:
        return NullWritable()
    if isinstance(v, (bytes, bytearray)):
        return BytesWritable(bytes(v))
    if isinstance(v, str):
        return Text(v)
    return NDArrayWritable(v)


def parse(s: str) -> Writable:
    """CSV token -> Text (DataVec keeps CSV fields as Text; numeric conversion happens on use)."""
    return Text(s)
