"""DataVec second tranche (datavec/transform_ext.py): transforms, conditions, sequence operations, windows, NDArray columns.
Reference tests: datavec-api/src/test/java/org/datavec/api/transform/{transform/TestTransforms,condition/TestConditions,
sequence/{TestSequenceSplit,TestWindowFunctions,TestReduceSequenceByWindowFunction},reduce/TestReductions}.java — expectations below are
worked out by hand from the documented semantics."""
import numpy as np
import pytest
import torch

from deeplearning4j_amd.datavec import (BooleanColumnCondition, ColumnType, ConditionFilter, ConditionOp, DeriveColumnsFromTime, Distance,
                                        FirstDigitTransform, InfiniteColumnCondition, LocalTransformExecutor, MathOp, NumericalColumnComparator,
                                        OverlappingTimeWindowFunction, RandomSplit, ReduceOp, Reducer, Schema, SequenceLengthCondition, SequenceMerge,
                                        SequenceTrimToLength, StringReduceOp, StringReducer, StringRegexColumnCondition, TimeWindowFunction,
                                        TransformProcess, TrivialColumnCondition)
from deeplearning4j_amd.datavec.writable import BooleanWritable, DoubleWritable, IntWritable, LongWritable, NDArrayWritable, Text, wrap


def W(*vals):
    return [wrap(v) for v in vals]


def vals(rec):
    return [w.toString() if isinstance(w, Text) else (w.value if not isinstance(w, NDArrayWritable) else w.value) for w in rec]


def test_first_digit_pivot_parse_and_integer_fill():
    s = Schema.Builder().addColumnDouble("x").addColumnCategorical("k", ["a", "b", "c"]).addColumnInteger("v").build()
    tp = (TransformProcess.Builder(s).firstDigitTransform("x", "fd").pivot("k", "v", 0).build())
    out_s = tp.getFinalSchema()
    assert out_s.getColumnNames() == ["x", "fd", "k[a]", "k[b]", "k[c]"]
    assert out_s.getMetaData("fd").stateNames == [str(d) for d in range(10)] + ["Other"] and out_s.getType("k[b]") == ColumnType.Integer
    rows = [W(123.4, "b", 7), W(-0.0056, "a", 9), W(float("nan"), "c", 1), W(0.0, "a", 2)]
    out = LocalTransformExecutor.execute(rows, tp)
    assert [vals(r)[1:] for r in out] == [["1", 0, 7, 0], ["5", 9, 0, 0], ["Other", 0, 0, 1], ["0", 2, 0, 0]]
    strict = TransformProcess.Builder(s).firstDigitTransform("x", "x", FirstDigitTransform.ExceptionOnInvalid).build()
    assert strict.getFinalSchema().getMetaData("x").stateNames == [str(d) for d in range(10)]
    with pytest.raises(ValueError):
        strict.execute(W(float("inf"), "a", 1))
    s2 = Schema.Builder().addColumnString("a").addColumnInteger("i").addColumnString("b").build()
    tp2 = TransformProcess.Builder(s2).replaceEmptyIntegerWithValue("i", -1).parseDouble().log2Normalizer("a", 5.0, 1.0, 2.0).build()
    assert tp2.getFinalSchema().getColumnTypes() == [ColumnType.Double, ColumnType.Integer, ColumnType.Double]
    r = tp2.execute([Text("9"), Text(""), Text("2.5")])
    assert abs(r[0].toDouble() - 2.0 * np.log2((9 - 1) / (5 - 1) + 1)) < 1e-12 and r[1].toInt() == -1 and r[2].toDouble() == 2.5


def test_string_list_transforms_and_string_reducer():
    s = Schema.Builder().addColumnString("tags").addColumnString("first").addColumnString("last").build()
    tp = (TransformProcess.Builder(s).stringListToCategoricalSet("tags", ["isA", "isB", "isC"], ["a", "b", "c"], ",")
          .mapAllStringsExceptList("first", "OTHER", ["ann", "bob"])
          .stringReduce(StringReducer.Builder(StringReduceOp.MERGE).appendColumns("first", "last").delimiter("_").outputColumnName("full").build()).build())
    assert tp.getFinalSchema().getColumnNames() == ["isA", "isB", "isC", "first", "last", "full"]
    out = LocalTransformExecutor.execute([W("a,c", "ann", "x"), W("", "zed", "y"), W("b", "bob", "z")], tp)
    assert [vals(r) for r in out] == [["true", "false", "true", "ann", "x", "ann_x"], ["false", "false", "false", "OTHER", "y", "OTHER_y"],
                                      ["false", "true", "false", "bob", "z", "bob_z"]]
    with pytest.raises(ValueError):
        tp.execute(W("a,q", "ann", "x"))
    for op, exp in ((StringReduceOp.APPEND, "annx"), (StringReduceOp.PREPEND, "xann"), (StringReduceOp.REPLACE, "x")):
        tpo = TransformProcess.Builder(s).stringReduce(StringReducer(["first", "last"], op, "o")).build()
        assert tpo.execute(W("t", "ann", "x"))[-1].toString() == exp
    vocab = ["a", "b", "c", "d"]
    s1 = Schema.Builder().addColumnString("t").build()
    cnt = TransformProcess.Builder(s1).stringListToCountsNDArray("t", vocab, ",").build().execute(W("a,c,a,d"))
    assert cnt[0].value.tolist() == [[2.0, 0.0, 1.0, 1.0]]
    bi = TransformProcess.Builder(s1).stringListToCountsNDArray("t", vocab, ",", binary=True, ignoreUnknown=True).build().execute(W("a,zz,a"))
    assert bi[0].value.tolist() == [[1.0, 0.0, 0.0, 0.0]]
    ix = TransformProcess.Builder(s1).stringListToIndicesNDArray("t", vocab, ",").build()
    assert ix.execute(W("d,a,b"))[0].value.tolist() == [[3.0, 0.0, 1.0]] and ix.getFinalSchema().getType("t") == ColumnType.NDArray
    with pytest.raises(ValueError):
        ix.execute(W("a,zz"))


def test_text_to_index_sequence_expansion():
    s = Schema.Builder().addColumnInteger("id").addColumnString("text").build()
    tp = TransformProcess.Builder(s).convertToSequence("id", None).textToCharacterIndex("text", "ch", {"a": 0, "b": 1, "c": 2}, False).build()
    seqs = LocalTransformExecutor.executeToSequence([W(1, "abc"), W(2, "cxa"), W(1, "b")], tp)
    assert [[vals(r) for r in seq] for seq in seqs] == [[[1, 0], [1, 1], [1, 2], [1, 1]], [[2, 2], [2, 0]]]
    assert tp.getFinalSchema().getType("ch") == ColumnType.Integer and tp.getFinalSchema().getMetaData("ch").max == 2
    tp2 = TransformProcess.Builder(s).convertToSequence("id", None).textToTermIndexSequence("text", "w", {"hello": 5, "world": 9}).build()
    assert [[vals(r) for r in q] for q in LocalTransformExecutor.executeToSequence([W(1, "hello world hello")], tp2)] == [[[1, 5], [1, 9], [1, 5]]]
    with pytest.raises(ValueError):
        LocalTransformExecutor.executeToSequence([W(1, "hello there")], tp2)
    with pytest.raises(ValueError):
        TransformProcess.Builder(s).textToCharacterIndex("text", "ch", {"a": 0}).build().execute(W(1, "a"))


def test_derive_columns_from_time():
    s = Schema.Builder().addColumnTime("t").addColumnInteger("v").build()
    ms = 1_468_411_445_678          # 2016-07-13 12:04:05.678 UTC, a Wednesday
    d = (DeriveColumnsFromTime.Builder("t").addIntegerDerivedColumn("year", "year").addIntegerDerivedColumn("month", "monthOfYear")
         .addIntegerDerivedColumn("dow", "dayOfWeek").addIntegerDerivedColumn("hour", "hourOfDay").addIntegerDerivedColumn("ms", "millisOfSecond")
         .addIntegerDerivedColumn("mod", "minuteOfDay").addStringDerivedColumn("date", "%Y-%m-%d").build())
    tp = TransformProcess.Builder(s).transform(d).build()
    assert tp.getFinalSchema().getColumnNames() == ["t", "year", "month", "dow", "hour", "ms", "mod", "date", "v"]
    assert vals(tp.execute(W(LongWritable(ms), 3)))[1:] == [2016, 7, 3, 12, 678, 12 * 60 + 4, "2016-07-13", 3]
    shifted = TransformProcess.Builder(s).transform(DeriveColumnsFromTime.Builder("t").timeZoneOffsetHours(13).insertAfter("v")
                                                     .addIntegerDerivedColumn("dom", "dayOfMonth").addIntegerDerivedColumn("h", "hourOfDay").build()).build()
    assert vals(shifted.execute(W(LongWritable(ms), 3)))[1:] == [3, 14, 1]
    with pytest.raises(ValueError):
        DeriveColumnsFromTime("t", [("x", "Integer", "fortnight")])


def test_ndarray_column_transforms():
    s = Schema.Builder().addColumnNDArray("a", [1, 3]).addColumnNDArray("b", [1, 3]).build()
    a, b = torch.tensor([[1.0, 2.0, 2.0]]), torch.tensor([[2.0, 0.0, 1.0]])
    tp = (TransformProcess.Builder(s).ndArrayColumnsMathOpTransform("sum", MathOp.Add, "a", "b").ndArrayDistanceTransform("cos", Distance.COSINE, "a", "b")
          .ndArrayDistanceTransform("l2", Distance.EUCLIDEAN, "a", "b").ndArrayDistanceTransform("l1", Distance.MANHATTAN, "a", "b")
          .ndArrayScalarOpTransform("a", MathOp.Multiply, 3).ndArrayMathFunctionTransform("b", "EXP").build())
    assert tp.getFinalSchema().getColumnTypes() == [ColumnType.NDArray] * 3 + [ColumnType.Double] * 3
    r = tp.execute([NDArrayWritable(a), NDArrayWritable(b)])
    torch.testing.assert_close(torch.as_tensor(r[2].value), a + b)
    assert abs(r[3].toDouble() - (1 - 4 / (3 * 5 ** 0.5))) < 1e-9 and abs(r[4].toDouble() - 6 ** 0.5) < 1e-9 and r[5].toDouble() == 4.0
    torch.testing.assert_close(torch.as_tensor(r[0].value).float(), 3 * a)
    torch.testing.assert_close(torch.as_tensor(r[1].value), torch.exp(b))
    with pytest.raises(ValueError):
        TransformProcess.Builder(s).ndArrayColumnsMathOpTransform("x", MathOp.Add, "a", "b").build().execute(
            [NDArrayWritable(a), NDArrayWritable(torch.ones(1, 2))])


def test_new_conditions():
    s = Schema.Builder().addColumnBoolean("flag").addColumnDouble("x").addColumnString("name").build()
    rows = [[BooleanWritable(True), DoubleWritable(1.0), Text("ab12")], [BooleanWritable(False), DoubleWritable(float("inf")), Text("zz")],
            [Text("true"), DoubleWritable(-float("inf")), Text("cd345")]]

    def kept(cond):
        return [i for i, r in enumerate(rows) if TransformProcess.Builder(s).filter(cond).build().execute(r) is not None]
    assert kept(BooleanColumnCondition("flag")) == [1]                       # the filter REMOVES matches
    assert kept(InfiniteColumnCondition("x")) == [0]
    assert kept(StringRegexColumnCondition("name", r"[a-z]{2}\d+")) == [1]
    assert kept(TrivialColumnCondition("x")) == []
    tp = TransformProcess.Builder(s).convertToSequence("name", None).filter(ConditionFilter(SequenceLengthCondition(ConditionOp.LessThan, 2))).build()
    many = rows + [[BooleanWritable(True), DoubleWritable(2.0), Text("zz")]]
    seqs = LocalTransformExecutor.executeToSequence(many, tp)
    assert len(seqs) == 1 and len(seqs[0]) == 2
    rng = SequenceLengthCondition(minLength=2, maxLength=3)
    assert rng.conditionSequence([1, 2]) and not rng.conditionSequence([1]) and not rng.conditionSequence([1, 2, 3, 4])
    bad = TransformProcess.Builder(s).filterInvalidNumColumns().build()
    assert bad.execute(rows[0]) is not None and bad.execute(rows[0][:2]) is None


def test_sequence_difference_trim_reduce_and_time_split():
    s = Schema.Builder().addColumnTime("t").addColumnInteger("v").addColumnDouble("d").build()
    seq = [W(LongWritable(t), v, float(v) / 2) for t, v in ((0, 1), (10, 4), (20, 9), (100, 16), (110, 25))]
    tp = TransformProcess.Builder(s).sequenceDifference("v", "dv").sequenceDifference("d", None, 2, "SpecifiedValue", -1.0).build()
    out = LocalTransformExecutor.executeSequenceToSequence([seq], tp)[0]
    assert [r[1].value for r in out] == [0, 3, 5, 7, 9] and [r[2].toDouble() for r in out] == [-1.0, -1.0, 4.0, 6.0, 8.0]
    assert tp.getFinalSchema().getColumnNames() == ["t", "dv", "d"] and tp.getFinalSchema().getType("dv") == ColumnType.Integer
    trim = TransformProcess.Builder(s).trimSequenceToLength(3).build()
    assert len(LocalTransformExecutor.executeSequenceToSequence([seq], trim)[0]) == 3
    pad = TransformProcess.Builder(s).trimOrPadSequenceToLength(7, [LongWritable(-1), 0, 0.0]).build()
    padded = LocalTransformExecutor.executeSequenceToSequence([seq, seq[:2]], pad)
    assert [len(q) for q in padded] == [7, 7] and padded[1][2][0].value == -1 and padded[1][1][1].value == 4
    with pytest.raises(ValueError):
        SequenceTrimToLength(4, SequenceTrimToLength.TrimOrPad)
    red = Reducer.Builder(ReduceOp.TakeFirst).sumColumns("v").meanColumns("d").build()
    rs = TransformProcess.Builder(s).reduceSequence(red).build()
    one = LocalTransformExecutor.executeSequenceToSequence([seq], rs)
    assert len(one) == 1 and len(one[0]) == 1 and one[0][0][1].toDouble() == 55.0 and abs(one[0][0][2].toDouble() - 5.5) < 1e-12
    assert rs.getFinalSchema().getColumnNames() == ["t", "sum(v)", "mean(d)"]
    sp = TransformProcess.Builder(s).splitSequenceByTimeSeparation("t", 50).build()
    parts = LocalTransformExecutor.executeSequenceToSequence([seq], sp)
    assert [[r[1].value for r in p] for p in parts] == [[1, 4, 9], [16, 25]]
    merged = SequenceMerge(NumericalColumnComparator("t")).mergeSequences(s, [seq[3:], seq[:3]])
    assert [r[0].value for r in merged] == [0, 10, 20, 100, 110]


def test_time_windows_and_reduce_by_window():
    s = Schema.Builder().addColumnTime("t").addColumnInteger("v").build()
    seq = [W(LongWritable(t), v) for t, v in ((1000, 1), (1500, 2), (2100, 3), (4100, 4), (4900, 5))]
    w = TimeWindowFunction("t", 1, "SECONDS")
    wins = w.applyToSequence(s, seq)
    assert [[r[1].value for r in q] for q in wins] == [[1, 2], [3], [4, 5]]                      # the empty 3000-3999 window is dropped
    keep = TimeWindowFunction("t", 1, "SECONDS", excludeEmptyWindows=False, addWindowStartTimeColumn=True, addWindowEndTimeColumn=True)
    wk = keep.applyToSequence(s, seq)
    assert [len(q) for q in wk] == [2, 1, 0, 2] and [w_.value for w_ in wk[0][0][2:]] == [1000, 2000]
    assert keep.outputSchema(s).getColumnNames() == ["t", "v", "windowStartTime", "windowEndTime"]
    ov = OverlappingTimeWindowFunction("t", 2, "SECONDS", 1, "SECONDS")
    wo = ov.applyToSequence(s, seq)
    assert [[r[1].value for r in q] for q in wo] == [[1, 2], [1, 2, 3], [3], [4, 5], [4, 5]]
    with pytest.raises(ValueError):
        OverlappingTimeWindowFunction("t", 1, "SECONDS", 2, "SECONDS")
    red = Reducer.Builder(ReduceOp.TakeFirst).sumColumns("v").build()
    tp = TransformProcess.Builder(s).reduceSequenceByWindow(red, w).build()
    out = LocalTransformExecutor.executeSequenceToSequence([seq], tp)[0]
    assert [(r[0].value, r[1].toDouble()) for r in out] == [(1000, 3.0), (2100, 3.0), (4100, 9.0)]
    assert tp.getFinalSchema().getColumnNames() == ["t", "sum(v)"]


def test_random_split_is_seeded_and_proportional():
    recs = list(range(2000))
    a, b = RandomSplit(0.7, seed=5).split(recs)
    a2, _ = RandomSplit(0.7, seed=5).split(recs)
    assert a == a2 and sorted(a + b) == recs and 1300 < len(a) < 1500
    with pytest.raises(ValueError):
        RandomSplit(1.5)
