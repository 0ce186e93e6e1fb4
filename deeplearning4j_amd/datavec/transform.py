"""Schema + TransformProcess + local executor + data analysis (reference: datavec-api/.../transform/**:
schema/Schema.java, TransformProcess.java, transform/{categorical,column,doubletransform,integer,string,time,condition}/*,
filter/*, condition/*, reduce/Reducer.java, sequence/*, join/Join.java, analysis/*; datavec-local LocalTransformExecutor,
AnalyzeLocal)."""
from __future__ import annotations

import copy
import json
import math
import re
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .writable import (BooleanWritable, DoubleWritable, FloatWritable, IntWritable, LongWritable, NullWritable, Text, Writable, wrap)


class ColumnType:
    String, Integer, Long, Double, Float, Categorical, Time, Bytes, Boolean, NDArray = \
        "String", "Integer", "Long", "Double", "Float", "Categorical", "Time", "Bytes", "Boolean", "NDArray"


class ColumnMetaData:
    def __init__(self, name, type_, stateNames=None, minAllowed=None, maxAllowed=None):
        self.name, self.type, self.stateNames, self.min, self.max = name, type_, (list(stateNames) if stateNames else None), minAllowed, maxAllowed

    def getName(self):
        return self.name

    def getColumnType(self):
        return self.type

    def getStateNames(self):
        return self.stateNames

    def isValid(self, w: Writable) -> bool:
        try:
            if self.type in (ColumnType.Integer, ColumnType.Long, ColumnType.Time):
                v = float(w.toString())
                if v != int(v):
                    return False
            elif self.type in (ColumnType.Double, ColumnType.Float):
                v = w.toDouble()
                if math.isnan(v) or math.isinf(v):
                    return False
            elif self.type == ColumnType.Categorical:
                return w.toString() in self.stateNames
            else:
                return True
            return (self.min is None or v >= self.min) and (self.max is None or v <= self.max)
        except (ValueError, TypeError):
            return False

    def to_json(self):
        return {"name": self.name, "type": self.type, "stateNames": self.stateNames, "min": self.min, "max": self.max}

    def __eq__(self, o):
        return isinstance(o, ColumnMetaData) and self.to_json() == o.to_json()

    def __repr__(self):
        return "%s:%s" % (self.name, self.type)


class Schema:
    def __init__(self, columns: Sequence[ColumnMetaData]):
        self.columns = list(columns)
        names = [c.name for c in self.columns]
        if len(set(names)) != len(names):
            raise ValueError("Duplicate column names in schema: %s" % names)

    class Builder:
        def __init__(self):
            self._cols: List[ColumnMetaData] = []

        def _add(self, type_, names, **kw):
            for n in names:
                self._cols.append(ColumnMetaData(n, type_, **kw))
            return self

        def addColumnString(self, *n): return self._add(ColumnType.String, n)
        addColumnsString = addColumnString
        def addColumnInteger(self, name, minAllowed=None, maxAllowed=None): return self._add(ColumnType.Integer, [name], minAllowed=minAllowed, maxAllowed=maxAllowed)
        def addColumnsInteger(self, *n): return self._add(ColumnType.Integer, n)
        def addColumnLong(self, name, minAllowed=None, maxAllowed=None): return self._add(ColumnType.Long, [name], minAllowed=minAllowed, maxAllowed=maxAllowed)
        def addColumnsLong(self, *n): return self._add(ColumnType.Long, n)
        def addColumnDouble(self, name, minAllowed=None, maxAllowed=None): return self._add(ColumnType.Double, [name], minAllowed=minAllowed, maxAllowed=maxAllowed)
        def addColumnsDouble(self, *n): return self._add(ColumnType.Double, n)
        def addColumnFloat(self, name, minAllowed=None, maxAllowed=None): return self._add(ColumnType.Float, [name], minAllowed=minAllowed, maxAllowed=maxAllowed)
        def addColumnsFloat(self, *n): return self._add(ColumnType.Float, n)
        def addColumnBoolean(self, *n): return self._add(ColumnType.Boolean, n)
        def addColumnTime(self, name, tz="UTC"): return self._add(ColumnType.Time, [name])
        def addColumnNDArray(self, name, shape=None): return self._add(ColumnType.NDArray, [name])
        def addColumnBytes(self, *n): return self._add(ColumnType.Bytes, n)

        def addColumnCategorical(self, name, *states):
            st = list(states[0]) if len(states) == 1 and isinstance(states[0], (list, tuple)) else list(states)
            return self._add(ColumnType.Categorical, [name], stateNames=st)

        def build(self):
            return Schema(self._cols)

    # -- queries
    def numColumns(self):
        return len(self.columns)

    def getColumnNames(self):
        return [c.name for c in self.columns]

    def getColumnTypes(self):
        return [c.type for c in self.columns]

    def getName(self, i):
        return self.columns[i].name

    def getType(self, i_or_name):
        return self.getMetaData(i_or_name).type

    def getMetaData(self, i_or_name) -> ColumnMetaData:
        return self.columns[i_or_name if isinstance(i_or_name, int) else self.getIndexOfColumn(i_or_name)]

    def getIndexOfColumn(self, name):
        for i, c in enumerate(self.columns):
            if c.name == name:
                return i
        raise KeyError("Unknown column: \"%s\" (columns: %s)" % (name, self.getColumnNames()))

    def hasColumn(self, name):
        return any(c.name == name for c in self.columns)

    def toJson(self):
        return json.dumps({"columns": [c.to_json() for c in self.columns]})

    @staticmethod
    def fromJson(s):
        return Schema([ColumnMetaData(c["name"], c["type"], c.get("stateNames"), c.get("min"), c.get("max")) for c in json.loads(s)["columns"]])

    def __eq__(self, o):
        return isinstance(o, Schema) and self.columns == o.columns

    def __repr__(self):
        return "Schema(%s)" % ", ".join(map(repr, self.columns))


SequenceSchema = Schema


# ----------------------------------------------------------------------------- conditions
class ConditionOp:
    LessThan, LessOrEqual, GreaterThan, GreaterOrEqual, Equal, NotEqual, InSet, NotInSet = \
        "LessThan", "LessOrEqual", "GreaterThan", "GreaterOrEqual", "Equal", "NotEqual", "InSet", "NotInSet"


def _cmp(op, a, b):
    if op == ConditionOp.LessThan: return a < b
    if op == ConditionOp.LessOrEqual: return a <= b
    if op == ConditionOp.GreaterThan: return a > b
    if op == ConditionOp.GreaterOrEqual: return a >= b
    if op == ConditionOp.Equal: return a == b
    if op == ConditionOp.NotEqual: return a != b
    if op == ConditionOp.InSet: return a in b
    if op == ConditionOp.NotInSet: return a not in b
    raise ValueError(op)


class Condition:
    def bind(self, schema: Schema):
        self.schema = schema
        return self

    def condition(self, record: List[Writable]) -> bool:
        raise NotImplementedError

    def conditionSequence(self, seq):
        return any(self.condition(r) for r in seq)


class ColumnCondition(Condition):
    CAST = staticmethod(lambda w: w.toString())

    def __init__(self, column, op, value):
        self.column, self.op = column, op
        self.value = set(value) if op in (ConditionOp.InSet, ConditionOp.NotInSet) else value

    def bind(self, schema):
        super().bind(schema)
        self._idx = schema.getIndexOfColumn(self.column)
        return self

    def condition(self, record):
        return _cmp(self.op, self.CAST(record[self._idx]), self.value)


class DoubleColumnCondition(ColumnCondition):
    CAST = staticmethod(lambda w: w.toDouble())


class IntegerColumnCondition(ColumnCondition):
    CAST = staticmethod(lambda w: w.toInt())


LongColumnCondition = IntegerColumnCondition


class StringColumnCondition(ColumnCondition):
    pass


class CategoricalColumnCondition(ColumnCondition):
    pass


class NullWritableColumnCondition(Condition):
    def __init__(self, column):
        self.column = column

    def bind(self, schema):
        super().bind(schema)
        self._idx = schema.getIndexOfColumn(self.column)
        return self

    def condition(self, record):
        return isinstance(record[self._idx], NullWritable)


class NaNColumnCondition(NullWritableColumnCondition):
    def condition(self, record):
        try:
            return math.isnan(record[self._idx].toDouble())
        except (ValueError, TypeError):
            return False


class InvalidValueColumnCondition(NullWritableColumnCondition):
    def condition(self, record):
        return not self.schema.getMetaData(self._idx).isValid(record[self._idx])


class BooleanCondition(Condition):
    def __init__(self, kind, *conds):
        self.kind, self.conds = kind, conds

    @staticmethod
    def AND(*c): return BooleanCondition("AND", *c)
    @staticmethod
    def OR(*c): return BooleanCondition("OR", *c)
    @staticmethod
    def NOT(c): return BooleanCondition("NOT", c)
    @staticmethod
    def XOR(a, b): return BooleanCondition("XOR", a, b)

    def bind(self, schema):
        super().bind(schema)
        for c in self.conds:
            c.bind(schema)
        return self

    def condition(self, record):
        r = [c.condition(record) for c in self.conds]
        return {"AND": all(r), "OR": any(r), "NOT": not r[0], "XOR": (r[0] != r[-1])}[self.kind]


# ----------------------------------------------------------------------------- steps
class _Step:
    def outputSchema(self, s: Schema) -> Schema:
        return s

    def bind(self, s: Schema):
        self.inputSchema = s
        return self

    def map(self, r: List[Writable]) -> Optional[List[Writable]]:
        return r

    def mapSequence(self, seq):
        out = [self.map(r) for r in seq]
        return [r for r in out if r is not None]


class ConditionFilter(_Step):
    """Removes records for which the condition holds (ConditionFilter.java: removeExample)."""

    def __init__(self, condition: Condition):
        self.cond = condition

    def bind(self, s):
        self.cond.bind(s)
        return super().bind(s)

    def map(self, r):
        return None if self.cond.condition(r) else r

    def mapSequence(self, seq):
        return None if self.cond.conditionSequence(seq) else seq


class FilterInvalidValues(_Step):
    def __init__(self, *columns):
        self.columns = list(columns)

    def bind(self, s):
        self._idx = [s.getIndexOfColumn(c) for c in (self.columns or s.getColumnNames())]
        return super().bind(s)

    def map(self, r):
        return None if any(not self.inputSchema.getMetaData(i).isValid(r[i]) for i in self._idx) else r


class _ColumnMap(_Step):
    """Transform of a single column in place; subclasses set ``newType`` / ``fn``."""
    newType = None

    def __init__(self, column):
        self.column = column

    def bind(self, s):
        self._idx = s.getIndexOfColumn(self.column)
        return super().bind(s)

    def newMeta(self, old: ColumnMetaData) -> ColumnMetaData:
        return ColumnMetaData(old.name, self.newType or old.type, old.stateNames)

    def outputSchema(self, s):
        cols = list(s.columns)
        i = s.getIndexOfColumn(self.column)
        cols[i] = self.newMeta(cols[i])
        return Schema(cols)

    def fn(self, w: Writable) -> Writable:
        raise NotImplementedError

    def map(self, r):
        r = list(r)
        r[self._idx] = self.fn(r[self._idx])
        return r


class LambdaColumn(_ColumnMap):
    def __init__(self, column, fn, newType=None):
        super().__init__(column)
        self._fn, self.newType = fn, newType

    def fn(self, w):
        return wrap(self._fn(w))


class MathOp:
    Add, Subtract, Multiply, Divide, Modulus, ReverseSubtract, ReverseDivide, ScalarMin, ScalarMax = \
        "Add", "Subtract", "Multiply", "Divide", "Modulus", "ReverseSubtract", "ReverseDivide", "ScalarMin", "ScalarMax"


def _math(op, a, b):
    if op == MathOp.Add: return a + b
    if op == MathOp.Subtract: return a - b
    if op == MathOp.Multiply: return a * b
    if op == MathOp.Divide: return a / b
    if op == MathOp.Modulus: return a % b
    if op == MathOp.ReverseSubtract: return b - a
    if op == MathOp.ReverseDivide: return b / a
    if op == MathOp.ScalarMin: return min(a, b)
    if op == MathOp.ScalarMax: return max(a, b)
    raise ValueError(op)


class MathFunction:
    ABS, ACOS, ASIN, ATAN, CEIL, COS, COSH, EXP, FLOOR, LOG, LOG10, SIGNUM, SIN, SINH, SQRT, TAN, TANH = \
        "ABS", "ACOS", "ASIN", "ATAN", "CEIL", "COS", "COSH", "EXP", "FLOOR", "LOG", "LOG10", "SIGNUM", "SIN", "SINH", "SQRT", "TAN", "TANH"
    _F = {"ABS": abs, "ACOS": math.acos, "ASIN": math.asin, "ATAN": math.atan, "CEIL": math.ceil, "COS": math.cos, "COSH": math.cosh,
          "EXP": math.exp, "FLOOR": math.floor, "LOG": math.log, "LOG10": math.log10, "SIGNUM": lambda v: (v > 0) - (v < 0), "SIN": math.sin,
          "SINH": math.sinh, "SQRT": math.sqrt, "TAN": math.tan, "TANH": math.tanh}


class ReduceOp:
    Min, Max, Range, Sum, Mean, Stdev, Count, CountUnique, TakeFirst, TakeLast, Prod, Variance, Append = \
        "Min", "Max", "Range", "Sum", "Mean", "Stdev", "Count", "CountUnique", "TakeFirst", "TakeLast", "Prod", "Variance", "Append"


def _reduce(op, ws: List[Writable]) -> Writable:
    if op == ReduceOp.TakeFirst: return ws[0]
    if op == ReduceOp.TakeLast: return ws[-1]
    if op == ReduceOp.Count: return LongWritable(len(ws))
    if op == ReduceOp.CountUnique: return LongWritable(len({w.toString() for w in ws}))
    if op == ReduceOp.Append: return Text("".join(w.toString() for w in ws))
    v = np.array([w.toDouble() for w in ws], dtype=np.float64)
    if op == ReduceOp.Min: return DoubleWritable(v.min())
    if op == ReduceOp.Max: return DoubleWritable(v.max())
    if op == ReduceOp.Range: return DoubleWritable(v.max() - v.min())
    if op == ReduceOp.Sum: return DoubleWritable(v.sum())
    if op == ReduceOp.Prod: return DoubleWritable(v.prod())
    if op == ReduceOp.Mean: return DoubleWritable(v.mean())
    if op == ReduceOp.Stdev: return DoubleWritable(v.std(ddof=1) if v.size > 1 else 0.0)
    if op == ReduceOp.Variance: return DoubleWritable(v.var(ddof=1) if v.size > 1 else 0.0)
    raise ValueError(op)


class Reducer:
    """Group-by reduction (reduce/Reducer.java)."""

    def __init__(self, keyColumns, defaultOp=ReduceOp.TakeFirst, ops: Optional[Dict[str, List[str]]] = None):
        self.keys, self.defaultOp, self.ops = list(keyColumns), defaultOp, dict(ops or {})

    class Builder:
        def __init__(self, defaultOp=ReduceOp.TakeFirst):
            self._default, self._keys, self._ops = defaultOp, [], OrderedDict()

        def keyColumns(self, *k):
            self._keys = list(k)
            return self

        def _op(self, op, cols):
            for c in cols:
                self._ops.setdefault(c, []).append(op)
            return self

        def minColumns(self, *c): return self._op(ReduceOp.Min, c)
        def maxColumns(self, *c): return self._op(ReduceOp.Max, c)
        def sumColumns(self, *c): return self._op(ReduceOp.Sum, c)
        def meanColumns(self, *c): return self._op(ReduceOp.Mean, c)
        def stdevColumns(self, *c): return self._op(ReduceOp.Stdev, c)
        def countColumns(self, *c): return self._op(ReduceOp.Count, c)
        def countUniqueColumns(self, *c): return self._op(ReduceOp.CountUnique, c)
        def rangeColumns(self, *c): return self._op(ReduceOp.Range, c)
        def takeFirstColumns(self, *c): return self._op(ReduceOp.TakeFirst, c)
        def takeLastColumns(self, *c): return self._op(ReduceOp.TakeLast, c)
        def prodColumns(self, *c): return self._op(ReduceOp.Prod, c)
        def varianceColumns(self, *c): return self._op(ReduceOp.Variance, c)
        def appendColumns(self, *c): return self._op(ReduceOp.Append, c)
        def multipleOpColmumns(self, ops, *c):
            for op in ops:
                self._op(op, c)
            return self

        def build(self):
            return Reducer(self._keys, self._default, self._ops)

    def _plan(self, s: Schema):
        plan = []
        for i, c in enumerate(s.columns):
            if c.name in self.keys:
                plan.append((i, None, c.name, c.type, c.stateNames))
            else:
                for op in self.ops.get(c.name, [self.defaultOp]):
                    name = c.name if (op in (ReduceOp.TakeFirst, ReduceOp.TakeLast) and c.name not in self.ops) else "%s(%s)" % (op.lower(), c.name)
                    if op in (ReduceOp.Count, ReduceOp.CountUnique):
                        t = ColumnType.Long
                    elif op in (ReduceOp.TakeFirst, ReduceOp.TakeLast):
                        t = c.type
                    elif op == ReduceOp.Append:
                        t = ColumnType.String
                    else:
                        t = ColumnType.Double
                    plan.append((i, op, name, t, c.stateNames if t == c.type else None))
        return plan

    def outputSchema(self, s):
        return Schema([ColumnMetaData(n, t, st) for _, _, n, t, st in self._plan(s)])

    def reduce(self, s: Schema, records: List[List[Writable]]) -> List[List[Writable]]:
        kidx = [s.getIndexOfColumn(k) for k in self.keys]
        groups: "OrderedDict[tuple, list]" = OrderedDict()
        for r in records:
            groups.setdefault(tuple(r[i].toString() for i in kidx), []).append(r)
        plan = self._plan(s)
        out = []
        for g in groups.values():
            out.append([g[0][i] if op is None else _reduce(op, [r[i] for r in g]) for i, op, _, _, _ in plan])
        return out


class Join:
    Inner, LeftOuter, RightOuter, FullOuter = "Inner", "LeftOuter", "RightOuter", "FullOuter"

    def __init__(self, joinType, leftSchema: Schema, rightSchema: Schema, leftKeys, rightKeys=None):
        self.type, self.left, self.right = joinType, leftSchema, rightSchema
        self.lk, self.rk = list(leftKeys), list(rightKeys or leftKeys)

    class Builder:
        def __init__(self, joinType):
            self._t, self._kw = joinType, {}

        def setSchemas(self, l, r):
            self._kw["leftSchema"], self._kw["rightSchema"] = l, r
            return self

        def setJoinColumns(self, *k):
            self._kw["leftKeys"] = list(k)
            return self

        setKeyColumns = setJoinColumns

        def setJoinColumnsLeft(self, *k):
            self._kw["leftKeys"] = list(k)
            return self

        def setJoinColumnsRight(self, *k):
            self._kw["rightKeys"] = list(k)
            return self

        def build(self):
            return Join(self._t, **self._kw)

    def outputSchema(self):
        cols = list(self.left.columns) + [c for c in self.right.columns if c.name not in self.rk]
        return Schema(cols)

    def execute(self, left: List[List[Writable]], right: List[List[Writable]]):
        li = [self.left.getIndexOfColumn(k) for k in self.lk]
        ri = [self.right.getIndexOfColumn(k) for k in self.rk]
        rrest = [i for i in range(self.right.numColumns()) if i not in ri]
        lrest_n = self.left.numColumns()
        idx: Dict[tuple, list] = {}
        for r in right:
            idx.setdefault(tuple(r[i].toString() for i in ri), []).append(r)
        out, used = [], set()
        for l in left:
            k = tuple(l[i].toString() for i in li)
            ms = idx.get(k)
            if ms:
                used.add(k)
                out += [list(l) + [m[i] for i in rrest] for m in ms]
            elif self.type in (Join.LeftOuter, Join.FullOuter):
                out.append(list(l) + [NullWritable() for _ in rrest])
        if self.type in (Join.RightOuter, Join.FullOuter):
            for k, ms in idx.items():
                if k not in used:
                    for m in ms:
                        row = [NullWritable()] * lrest_n
                        for a, b in zip(li, ri):
                            row[a] = m[b]
                        out.append(row + [m[i] for i in rrest])
        return out


# ----------------------------------------------------------------------------- TransformProcess
class TransformProcess:
    def __init__(self, initialSchema: Schema, steps: List[object]):
        self.initialSchema, self.steps = initialSchema, steps
        s = initialSchema
        self._schemas = [s]
        for st in steps:
            if isinstance(st, _Step):
                st.bind(s)
                s = st.outputSchema(s)
            elif isinstance(st, _ToSequence):
                st.bind(s)
            elif isinstance(st, Reducer):
                st._in = s
                s = st.outputSchema(s)
            elif isinstance(st, _SortedRank):
                s.getIndexOfColumn(st.sortCol)
                if hasattr(st, "setInputSchema"):
                    st.setInputSchema(s)
                s = Schema(list(s.columns) + [ColumnMetaData(st.newCol, ColumnType.Long)])
            self._schemas.append(s)
        self.finalSchema = s

    def getFinalSchema(self):
        return self.finalSchema

    def getInitialSchema(self):
        return self.initialSchema

    def getSchemaAfterStep(self, i):
        return self._schemas[i + 1]

    def getActionList(self):
        return list(self.steps)

    # -- single-record execution (no reductions / sequence conversion)
    def execute(self, record: List[Writable]) -> Optional[List[Writable]]:
        r = list(record)
        for st in self.steps:
            if not isinstance(st, _Step):
                raise ValueError("Cannot execute %s on a single record; use LocalTransformExecutor" % type(st).__name__)
            r = st.map(r)
            if r is None:
                return None
        return r

    def executeSequence(self, seq):
        for st in self.steps:
            seq = st.mapSequence(seq)
            if seq is None:
                return None
        return seq

    def toJson(self):
        def enc(o):
            if hasattr(o, "to_json") and isinstance(o, _SortedRank):
                return o.to_json()
            d = {"@class": type(o).__name__}
            for k, v in vars(o).items():
                if k.startswith("_") or k in ("inputSchema", "schema"):
                    continue
                d[k] = enc(v) if isinstance(v, (_Step, Condition, Reducer)) else ([enc(x) if isinstance(x, Condition) else x for x in v]
                                                                                   if isinstance(v, (list, tuple)) else
                                                                                   (sorted(v) if isinstance(v, set) else (None if callable(v) else v)))
            return d
        return json.dumps({"initialSchema": json.loads(self.initialSchema.toJson()), "steps": [enc(s) for s in self.steps]}, default=str)

    class Builder:
        def __init__(self, schema: Schema):
            self._schema, self._steps = schema, []

        def _add(self, s):
            self._steps.append(s)
            return self

        def build(self):
            return TransformProcess(self._schema, self._steps)

        # -- column structure
        def removeColumns(self, *c): return self._add(RemoveColumns(*c))
        def removeAllColumnsExceptFor(self, *c): return self._add(RemoveColumns(*c, keep=True))
        def renameColumn(self, old, new): return self._add(RenameColumns([old], [new]))
        def renameColumns(self, olds, news): return self._add(RenameColumns(olds, news))
        def reorderColumns(self, *c): return self._add(ReorderColumns(*c))
        def duplicateColumn(self, col, new): return self._add(DuplicateColumns([col], [new]))
        def duplicateColumns(self, cols, news): return self._add(DuplicateColumns(cols, news))
        def addConstantColumn(self, name, type_, value): return self._add(AddConstantColumn(name, type_, value))
        def addConstantDoubleColumn(self, name, v): return self._add(AddConstantColumn(name, ColumnType.Double, DoubleWritable(v)))
        def addConstantIntegerColumn(self, name, v): return self._add(AddConstantColumn(name, ColumnType.Integer, IntWritable(v)))
        # -- filters
        def filter(self, f): return self._add(f if isinstance(f, _Step) else ConditionFilter(f))
        # -- categorical
        def categoricalToInteger(self, *c):
            for x in c:
                self._add(CategoricalToInteger(x))
            return self

        def categoricalToOneHot(self, *c):
            for x in c:
                self._add(CategoricalToOneHot(x))
            return self

        def integerToCategorical(self, col, states): return self._add(IntegerToCategorical(col, states))
        def stringToCategorical(self, col, states): return self._add(StringToCategorical(col, states))
        def integerToOneHot(self, col, minV, maxV): return self._add(IntegerToOneHot(col, minV, maxV))
        # -- math
        def doubleMathOp(self, col, op, scalar): return self._add(ScalarMathOp(col, op, float(scalar), ColumnType.Double))
        def integerMathOp(self, col, op, scalar): return self._add(ScalarMathOp(col, op, int(scalar), ColumnType.Integer))
        longMathOp = integerMathOp
        floatMathOp = doubleMathOp
        def doubleMathFunction(self, col, fn): return self._add(MathFunctionOp(col, fn))
        floatMathFunction = doubleMathFunction
        def doubleColumnsMathOp(self, new, op, *cols): return self._add(ColumnsMathOp(new, op, cols, ColumnType.Double))
        def integerColumnsMathOp(self, new, op, *cols): return self._add(ColumnsMathOp(new, op, cols, ColumnType.Integer))
        longColumnsMathOp = integerColumnsMathOp
        floatColumnsMathOp = doubleColumnsMathOp
        # -- normalisation (needs a DataAnalysis)
        def normalize(self, col, kind, analysis): return self._add(Normalize(col, kind, analysis))
        # -- conversion
        def convertToDouble(self, col): return self._add(LambdaColumn(col, lambda w: w.toDouble(), ColumnType.Double))
        def convertToInteger(self, col): return self._add(LambdaColumn(col, lambda w: w.toInt(), ColumnType.Integer))
        def convertToString(self, col): return self._add(LambdaColumn(col, lambda w: w.toString(), ColumnType.String))
        # -- conditional replace
        def conditionalReplaceValueTransform(self, col, newValue, cond): return self._add(ConditionalReplace(col, wrap(newValue), cond))
        def conditionalReplaceValueTransformWithDefault(self, col, yes, no, cond): return self._add(ConditionalReplace(col, wrap(yes), cond, wrap(no)))
        def conditionalCopyValueTransform(self, col, src, cond): return self._add(ConditionalCopy(col, src, cond))
        def replaceInvalidWithInteger(self, col, v): return self._add(ConditionalReplace(col, IntWritable(v), InvalidValueColumnCondition(col)))
        def replaceEmptyStringWithValue(self, col, v): return self._add(ConditionalReplace(col, Text(v), StringColumnCondition(col, ConditionOp.Equal, "")))
        # -- strings
        def stringMapTransform(self, col, mapping): return self._add(LambdaColumn(col, lambda w: mapping.get(w.toString(), w.toString())))
        def replaceStringTransform(self, col, mapping):
            def f(w):
                s = w.toString()
                for k, v in mapping.items():
                    s = re.sub(k, v, s)
                return s
            return self._add(LambdaColumn(col, f))
        def appendStringColumnTransform(self, col, s): return self._add(LambdaColumn(col, lambda w: w.toString() + s))
        def changeCaseStringTransform(self, col, upper=False): return self._add(LambdaColumn(col, lambda w: w.toString().upper() if upper else w.toString().lower()))
        def stringRemoveWhitespaceTransform(self, col): return self._add(LambdaColumn(col, lambda w: re.sub(r"\s", "", w.toString())))
        def concatenateStringColumns(self, new, delim, *cols): return self._add(ConcatStrings(new, delim, cols))
        def stringToTimeTransform(self, col, fmt, tz="UTC"): return self._add(StringToTime(col, fmt))
        def timeMathOp(self, col, op, amount, unit="MILLISECONDS"):
            mult = {"MILLISECONDS": 1, "SECONDS": 1000, "MINUTES": 60000, "HOURS": 3600000, "DAYS": 86400000}[unit]
            return self._add(ScalarMathOp(col, op, int(amount) * mult, ColumnType.Time))
        def transform(self, t): return self._add(t)
        # -- sequence
        def convertToSequence(self, keyColumn=None, comparator=None): return self._add(_ToSequence(keyColumn, comparator))
        def convertFromSequence(self): return self._add(_FromSequence())
        def offsetSequence(self, cols, offset, op="InPlace"): return self._add(SequenceOffset(cols, offset, op))
        def trimSequence(self, n, fromFirst=True): return self._add(SequenceTrim(n, fromFirst))
        def sequenceMovingWindowReduce(self, col, lookback, op): return self._add(SequenceMovingWindowReduce(col, lookback, op))
        def splitSequence(self, maxLen): return self._add(SplitMaxLength(maxLen))
        # -- reduce
        def reduce(self, reducer): return self._add(reducer)
        def calculateSortedRank(self, newCol, sortCol, comparator=None, ascending=True):
            """CalculateSortedRank (TransformProcess.java:1211-1246): ``comparator`` is a datavec.analysis.WritableComparator
            (default: by the column's type); a bool in its place is taken as ``ascending`` (earlier signature)."""
            from .analysis import CalculateSortedRank
            if isinstance(comparator, bool):
                comparator, ascending = None, comparator
            return self._add(CalculateSortedRank(newCol, sortCol, comparator, ascending))


# -- concrete steps
class RemoveColumns(_Step):
    def __init__(self, *cols, keep=False):
        self.cols, self.keep = list(cols), keep

    def bind(self, s):
        for c in self.cols:
            s.getIndexOfColumn(c)
        self._keep = [i for i, c in enumerate(s.columns) if (c.name in self.cols) == self.keep]
        return super().bind(s)

    def outputSchema(self, s):
        return Schema([c for c in s.columns if (c.name in self.cols) == self.keep])

    def map(self, r):
        return [r[i] for i in self._keep]


class RenameColumns(_Step):
    def __init__(self, olds, news):
        self.olds, self.news = list(olds), list(news)

    def outputSchema(self, s):
        m = dict(zip(self.olds, self.news))
        for o in self.olds:
            s.getIndexOfColumn(o)
        return Schema([ColumnMetaData(m.get(c.name, c.name), c.type, c.stateNames, c.min, c.max) for c in s.columns])


class ReorderColumns(_Step):
    def __init__(self, *cols):
        self.cols = list(cols)

    def bind(self, s):
        first = [s.getIndexOfColumn(c) for c in self.cols]
        self._order = first + [i for i in range(s.numColumns()) if i not in first]
        return super().bind(s)

    def outputSchema(self, s):
        first = [s.getIndexOfColumn(c) for c in self.cols]
        order = first + [i for i in range(s.numColumns()) if i not in first]
        return Schema([s.columns[i] for i in order])

    def map(self, r):
        return [r[i] for i in self._order]


class DuplicateColumns(_Step):
    def __init__(self, cols, news):
        self.cols, self.news = list(cols), list(news)

    def bind(self, s):
        self._idx = [s.getIndexOfColumn(c) for c in self.cols]
        return super().bind(s)

    def outputSchema(self, s):
        out = []
        m = dict(zip(self.cols, self.news))
        for c in s.columns:
            out.append(c)
            if c.name in m:
                out.append(ColumnMetaData(m[c.name], c.type, c.stateNames, c.min, c.max))
        return Schema(out)

    def map(self, r):
        out = []
        for i, w in enumerate(r):
            out.append(w)
            if i in self._idx:
                out.append(w)
        return out


class AddConstantColumn(_Step):
    def __init__(self, name, type_, value):
        self.name, self.type, self.value = name, type_, wrap(value)

    def outputSchema(self, s):
        return Schema(list(s.columns) + [ColumnMetaData(self.name, self.type)])

    def map(self, r):
        return list(r) + [self.value]


class CategoricalToInteger(_ColumnMap):
    newType = ColumnType.Integer

    def bind(self, s):
        self._states = s.getMetaData(self.column).stateNames
        return super().bind(s)

    def newMeta(self, old):
        return ColumnMetaData(old.name, ColumnType.Integer, None, 0, len(old.stateNames) - 1)

    def fn(self, w):
        return IntWritable(self._states.index(w.toString()))


class CategoricalToOneHot(_Step):
    def __init__(self, column):
        self.column = column

    def bind(self, s):
        self._idx = s.getIndexOfColumn(self.column)
        self._states = s.getMetaData(self.column).stateNames
        return super().bind(s)

    def outputSchema(self, s):
        out = []
        for c in s.columns:
            if c.name == self.column:
                out += [ColumnMetaData("%s[%s]" % (c.name, st), ColumnType.Integer, None, 0, 1) for st in c.stateNames]
            else:
                out.append(c)
        return Schema(out)

    def map(self, r):
        k = self._states.index(r[self._idx].toString())
        return list(r[:self._idx]) + [IntWritable(1 if j == k else 0) for j in range(len(self._states))] + list(r[self._idx + 1:])


class IntegerToOneHot(CategoricalToOneHot):
    def __init__(self, column, minV, maxV):
        super().__init__(column)
        self.minV, self.maxV = int(minV), int(maxV)

    def bind(self, s):
        _Step.bind(self, s)
        self._idx = s.getIndexOfColumn(self.column)
        self._states = [str(i) for i in range(self.minV, self.maxV + 1)]
        return self

    def outputSchema(self, s):
        out = []
        for c in s.columns:
            if c.name == self.column:
                out += [ColumnMetaData("%s[%d]" % (c.name, i), ColumnType.Integer, None, 0, 1) for i in range(self.minV, self.maxV + 1)]
            else:
                out.append(c)
        return Schema(out)

    def map(self, r):
        k = r[self._idx].toInt() - self.minV
        n = self.maxV - self.minV + 1
        return list(r[:self._idx]) + [IntWritable(1 if j == k else 0) for j in range(n)] + list(r[self._idx + 1:])


class IntegerToCategorical(_ColumnMap):
    newType = ColumnType.Categorical

    def __init__(self, column, states):
        super().__init__(column)
        self.states = dict(states) if isinstance(states, dict) else {i: s for i, s in enumerate(states)}

    def newMeta(self, old):
        return ColumnMetaData(old.name, ColumnType.Categorical, [self.states[k] for k in sorted(self.states)])

    def fn(self, w):
        return Text(self.states[w.toInt()])


class StringToCategorical(_ColumnMap):
    def __init__(self, column, states):
        super().__init__(column)
        self.states = list(states)

    def newMeta(self, old):
        return ColumnMetaData(old.name, ColumnType.Categorical, self.states)

    def fn(self, w):
        return w


class ScalarMathOp(_ColumnMap):
    def __init__(self, column, op, scalar, type_):
        super().__init__(column)
        self.op, self.scalar, self.newType = op, scalar, type_

    def fn(self, w):
        if self.newType in (ColumnType.Integer, ColumnType.Long, ColumnType.Time):
            a = w.toInt()
            v = a // self.scalar if self.op == MathOp.Divide else (self.scalar // a if self.op == MathOp.ReverseDivide else _math(self.op, a, self.scalar))
            return IntWritable(v) if self.newType == ColumnType.Integer else LongWritable(v)
        return DoubleWritable(_math(self.op, w.toDouble(), self.scalar))


class MathFunctionOp(_ColumnMap):
    newType = ColumnType.Double

    def __init__(self, column, fn):
        super().__init__(column)
        self.function = fn

    def fn(self, w):
        return DoubleWritable(MathFunction._F[self.function](w.toDouble()))


class ColumnsMathOp(_Step):
    def __init__(self, newName, op, cols, type_):
        self.newName, self.op, self.cols, self.type = newName, op, list(cols), type_

    def bind(self, s):
        self._idx = [s.getIndexOfColumn(c) for c in self.cols]
        return super().bind(s)

    def outputSchema(self, s):
        return Schema(list(s.columns) + [ColumnMetaData(self.newName, self.type)])

    def map(self, r):
        isint = self.type in (ColumnType.Integer, ColumnType.Long)
        vals = [r[i].toInt() if isint else r[i].toDouble() for i in self._idx]
        acc = vals[0]
        for v in vals[1:]:
            acc = (acc // v) if (isint and self.op == MathOp.Divide) else _math(self.op, acc, v)
        return list(r) + [IntWritable(acc) if isint else DoubleWritable(acc)]


class Normalize(_ColumnMap):
    MinMax, MinMax2, Standardize, SubtractMean, Log2Mean, Log2MeanExcludingMin = "MinMax", "MinMax2", "Standardize", "SubtractMean", "Log2Mean", "Log2MeanExcludingMin"
    newType = ColumnType.Double

    def __init__(self, column, kind, analysis):
        super().__init__(column)
        self.kind = kind
        a = analysis.getColumnAnalysis(column)
        self.min, self.max, self.mean, self.std = a.min, a.max, a.mean, a.stdev

    def fn(self, w):
        v = w.toDouble()
        if self.kind == self.MinMax:
            return DoubleWritable((v - self.min) / (self.max - self.min) if self.max > self.min else 0.0)
        if self.kind == self.MinMax2:
            return DoubleWritable(2 * (v - self.min) / (self.max - self.min) - 1 if self.max > self.min else 0.0)
        if self.kind == self.Standardize:
            return DoubleWritable((v - self.mean) / self.std if self.std > 0 else 0.0)
        if self.kind == self.SubtractMean:
            return DoubleWritable(v - self.mean)
        if self.kind == self.Log2Mean:
            return DoubleWritable(math.log2((v - self.min) / (self.mean - self.min) + 1))
        raise ValueError(self.kind)


class ConditionalReplace(_ColumnMap):
    def __init__(self, column, yes: Writable, cond: Condition, no: Optional[Writable] = None):
        super().__init__(column)
        self.yes, self.cond, self.no = yes, cond, no

    def bind(self, s):
        self.cond.bind(s)
        return super().bind(s)

    def map(self, r):
        r = list(r)
        if self.cond.condition(r):
            r[self._idx] = self.yes
        elif self.no is not None:
            r[self._idx] = self.no
        return r


class ConditionalCopy(_ColumnMap):
    def __init__(self, column, src, cond):
        super().__init__(column)
        self.src, self.cond = src, cond

    def bind(self, s):
        self.cond.bind(s)
        self._src = s.getIndexOfColumn(self.src)
        return super().bind(s)

    def map(self, r):
        r = list(r)
        if self.cond.condition(r):
            r[self._idx] = r[self._src]
        return r


class ConcatStrings(_Step):
    def __init__(self, new, delim, cols):
        self.new, self.delim, self.cols = new, delim, list(cols)

    def bind(self, s):
        self._idx = [s.getIndexOfColumn(c) for c in self.cols]
        return super().bind(s)

    def outputSchema(self, s):
        return Schema(list(s.columns) + [ColumnMetaData(self.new, ColumnType.String)])

    def map(self, r):
        return list(r) + [Text(self.delim.join(r[i].toString() for i in self._idx))]


class StringToTime(_ColumnMap):
    newType = ColumnType.Time

    def __init__(self, column, fmt):
        super().__init__(column)
        # Joda/Java pattern -> strptime
        self.fmt = (fmt.replace("yyyy", "%Y").replace("YYYY", "%Y").replace("MM", "%m").replace("dd", "%d").replace("HH", "%H")
                    .replace("mm", "%M").replace("ss", "%S").replace("SSS", "%f"))

    def fn(self, w):
        import calendar
        import datetime as dt
        t = dt.datetime.strptime(w.toString(), self.fmt)
        return LongWritable(calendar.timegm(t.timetuple()) * 1000 + t.microsecond // 1000)


class _SortedRank:
    def __init__(self, newCol, sortCol, ascending):
        self.newCol, self.sortCol, self.ascending = newCol, sortCol, ascending


class _ToSequence:
    def __init__(self, key, comparator):
        self.key, self.comparator = key, comparator

    def bind(self, s):
        self._key = s.getIndexOfColumn(self.key) if self.key is not None else None
        cmp = self.comparator
        self._sort = s.getIndexOfColumn(cmp.column) if cmp is not None else None
        return self


class NumericalColumnComparator:
    def __init__(self, column, ascending=True):
        self.column, self.ascending = column, ascending


class StringComparator(NumericalColumnComparator):
    pass


class _FromSequence:
    pass


class SequenceOffset(_Step):
    def __init__(self, cols, offset, op="InPlace"):
        self.cols, self.offset, self.op = list(cols), int(offset), op

    def bind(self, s):
        self._idx = [s.getIndexOfColumn(c) for c in self.cols]
        return super().bind(s)

    def mapSequence(self, seq):
        n, k = len(seq), abs(self.offset)
        if k >= n:
            return None
        out = []
        for t in range(k, n) if self.offset > 0 else range(0, n - k):
            row = list(seq[t])
            for i in self._idx:
                row[i] = seq[t - self.offset][i]
            out.append(row)
        return out


class SequenceTrim(_Step):
    def __init__(self, n, fromFirst=True):
        self.n, self.fromFirst = int(n), fromFirst

    def mapSequence(self, seq):
        return seq[self.n:] if self.fromFirst else seq[:len(seq) - self.n]


class SequenceMovingWindowReduce(_Step):
    def __init__(self, column, lookback, op):
        self.column, self.lookback, self.op = column, int(lookback), op

    def bind(self, s):
        self._idx = s.getIndexOfColumn(self.column)
        return super().bind(s)

    def outputSchema(self, s):
        return Schema(list(s.columns) + [ColumnMetaData("%s(%d,%s)" % (self.op.lower(), self.lookback, self.column), ColumnType.Double)])

    def mapSequence(self, seq):
        out = []
        for t, r in enumerate(seq):
            win = [seq[k][self._idx] for k in range(max(0, t - self.lookback + 1), t + 1)]
            out.append(list(r) + [_reduce(self.op, win)])
        return out


class SplitMaxLength(_Step):
    def __init__(self, maxLen):
        self.maxLen = int(maxLen)


# ----------------------------------------------------------------------------- executor
class LocalTransformExecutor:
    @staticmethod
    def execute(records: List[List[Writable]], tp: TransformProcess):
        """Flat records in -> flat records out (raises if the process ends in sequence form)."""
        out = LocalTransformExecutor._run(records, tp, False)
        if out and out[0] and isinstance(out[0][0], list):
            raise ValueError("TransformProcess returns sequences: use executeToSequence")
        return out

    @staticmethod
    def executeToSequence(records, tp):
        return LocalTransformExecutor._run(records, tp, False)

    @staticmethod
    def executeSequenceToSequence(sequences, tp):
        return LocalTransformExecutor._run(sequences, tp, True)

    @staticmethod
    def executeSequenceToSeparate(sequences, tp):
        return LocalTransformExecutor._run(sequences, tp, True)

    @staticmethod
    def executeJoin(join: Join, left, right):
        return join.execute(left, right)

    @staticmethod
    def _run(data, tp: TransformProcess, is_seq: bool):
        data = [list(r) for r in data]
        for k, st in enumerate(tp.steps):
            schema = tp._schemas[k]
            if isinstance(st, _ToSequence):
                if st._key is None:
                    data = [data]
                else:
                    groups: "OrderedDict[str, list]" = OrderedDict()
                    for r in data:
                        groups.setdefault(r[st._key].toString(), []).append(r)
                    data = list(groups.values())
                if st._sort is not None:
                    num = not isinstance(st.comparator, StringComparator)
                    for g in data:
                        g.sort(key=(lambda r: r[st._sort].toDouble()) if num else (lambda r: r[st._sort].toString()),
                               reverse=not st.comparator.ascending)
                is_seq = True
            elif isinstance(st, _FromSequence):
                data = [r for seq in data for r in seq]
                is_seq = False
            elif isinstance(st, Reducer):
                if is_seq:
                    raise ValueError("Cannot reduce sequences")
                data = st.reduce(schema, data)
            elif isinstance(st, _SortedRank):
                if is_seq:
                    raise ValueError("CalculateSortedRank on sequences: not supported (LocalTransformExecutor.java:472-476)")
                if hasattr(st, "execute"):
                    data = st.execute(data, schema)           # sorted order + rank column, as the reference executor
                else:
                    i = schema.getIndexOfColumn(st.sortCol)
                    order = sorted(range(len(data)), key=lambda j: data[j][i].toDouble(), reverse=not st.ascending)
                    data = [data[j] + [LongWritable(rk)] for rk, j in enumerate(order)]
            elif isinstance(st, SplitMaxLength):
                data = [seq[i:i + st.maxLen] for seq in data for i in range(0, len(seq), st.maxLen)]
            elif hasattr(st, "runAll"):              # steps that change the number of sequences (transform_ext.py)
                data, is_seq = st.runAll(data, schema, is_seq)
            elif is_seq:
                out = [st.mapSequence(seq) for seq in data]
                data = [s for s in out if s]
            else:
                out = [st.map(r) for r in data]
                data = [r for r in out if r is not None]
        return data


# ----------------------------------------------------------------------------- analysis
class ColumnAnalysis:
    def __init__(self, type_):
        self.type = type_
        self.count = self.countMissing = 0
        self.min = self.max = self.mean = self.stdev = float("nan")
        self.uniqueCount = 0
        self.stateCounts: Dict[str, int] = {}
        self.histogramBuckets = self.histogramCounts = None
        self.minLength = self.maxLength = 0

    getMin = lambda self: self.min
    getMax = lambda self: self.max
    getMean = lambda self: self.mean
    getSampleStdev = lambda self: self.stdev
    getCountTotal = lambda self: self.count

    def __repr__(self):
        if self.type in (ColumnType.Categorical, ColumnType.String):
            return "%s(count=%d, unique=%d)" % (self.type, self.count, self.uniqueCount)
        return "%s(count=%d, min=%g, max=%g, mean=%g, stdev=%g)" % (self.type, self.count, self.min, self.max, self.mean, self.stdev)


class DataAnalysis:
    def __init__(self, schema: Schema, analyses: List[ColumnAnalysis]):
        self.schema, self.columnAnalysis = schema, analyses

    def getColumnAnalysis(self, name_or_idx) -> ColumnAnalysis:
        return self.columnAnalysis[name_or_idx if isinstance(name_or_idx, int) else self.schema.getIndexOfColumn(name_or_idx)]

    def getSchema(self):
        return self.schema

    def __str__(self):
        return "\n".join("%-24s %s" % (c.name, a) for c, a in zip(self.schema.columns, self.columnAnalysis))


class AnalyzeLocal:
    """Kept for imports of the first implementation; datavec.analysis.AnalyzeLocal is the full one (typed column analyses, quality
    classes, sequences) and is what ``deeplearning4j_amd.datavec.AnalyzeLocal`` resolves to."""

    @staticmethod
    def analyze(schema: Schema, reader_or_records, maxHistogramBuckets=20) -> DataAnalysis:
        from . import analysis
        return analysis.AnalyzeLocal.analyze(schema, reader_or_records, maxHistogramBuckets)

    @staticmethod
    def _analyze_basic(schema: Schema, reader_or_records, maxHistogramBuckets=20) -> DataAnalysis:
        recs = list(reader_or_records)
        out = []
        for i, c in enumerate(schema.columns):
            a = ColumnAnalysis(c.type)
            col = [r[i] for r in recs]
            a.count = len(col)
            if c.type in (ColumnType.Integer, ColumnType.Long, ColumnType.Double, ColumnType.Float, ColumnType.Time):
                vals = []
                for w in col:
                    try:
                        v = w.toDouble()
                        if math.isnan(v):
                            a.countMissing += 1
                        else:
                            vals.append(v)
                    except (ValueError, TypeError):
                        a.countMissing += 1
                if vals:
                    v = np.asarray(vals)
                    a.min, a.max, a.mean = float(v.min()), float(v.max()), float(v.mean())
                    a.stdev = float(v.std(ddof=1)) if v.size > 1 else 0.0
                    a.uniqueCount = int(np.unique(v).size)
                    cnt, edges = np.histogram(v, bins=min(maxHistogramBuckets, max(1, a.uniqueCount)))
                    a.histogramBuckets, a.histogramCounts = edges.tolist(), cnt.tolist()
            else:
                ss = [w.toString() for w in col]
                for s in ss:
                    a.stateCounts[s] = a.stateCounts.get(s, 0) + 1
                a.uniqueCount = len(a.stateCounts)
                a.minLength, a.maxLength = (min(map(len, ss)), max(map(len, ss))) if ss else (0, 0)
            out.append(a)
        return DataAnalysis(schema, out)

    @staticmethod
    def analyzeQuality(schema: Schema, reader_or_records):
        from . import analysis
        return analysis.AnalyzeLocal.analyzeQuality(schema, reader_or_records)
