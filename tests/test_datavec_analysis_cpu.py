"""DataVec analysis / quality / HTML / sorted rank (reference: datavec-api transform/{analysis,quality,ui,rank}, datavec-local
AnalyzeLocal + LocalTransformExecutor). Expected values are computed by hand or with NumPy on the same columns."""
import json
import math

import numpy as np
import pytest

from deeplearning4j_amd.datavec import (AnalyzeLocal, BytesWritable, CalculateSortedRank, CategoricalAnalysis, ColumnType, DataAnalysis,
                                        DataQualityAnalysis, DoubleAnalysis, DoubleQuality, DoubleWritable, DoubleWritableComparator,
                                        HtmlAnalysis, HtmlSequencePlotting, IntegerAnalysis, IntegerQuality, IntWritable,
                                        LocalTransformExecutor, LongWritable, NDArrayWritable, NullWritable, ReverseComparator, Schema,
                                        SequenceDataAnalysis, StringAnalysis, StringQuality, Text, TextWritableComparator,
                                        TransformProcess)


def _schema():
    return (Schema.Builder().addColumnInteger("id").addColumnDouble("x").addColumnCategorical("kind", ["a", "b", "c"])
            .addColumnString("note").build())


def _rows():
    xs = [1.5, -2.0, 0.0, 7.25, 0.0, float("nan")]
    kinds = ["a", "b", "a", "c", "a", "b"]
    notes = ["hello", "", "42", "a", "  ", "x_1"]
    return [[IntWritable(i), DoubleWritable(x), Text(k), Text(n)] for i, (x, k, n) in enumerate(zip(xs, kinds, notes))]


def test_typed_column_analyses():
    an = AnalyzeLocal.analyze(_schema(), _rows(), maxHistogramBuckets=4)
    assert isinstance(an, DataAnalysis)
    ai, ax, ak, an_ = (an.getColumnAnalysis(n) for n in ("id", "x", "kind", "note"))
    assert isinstance(ai, IntegerAnalysis) and isinstance(ax, DoubleAnalysis) and isinstance(ak, CategoricalAnalysis) and isinstance(an_, StringAnalysis)
    assert (ai.getMin(), ai.getMax(), ai.getCountTotal(), ai.getCountZero(), ai.getCountPositive(), ai.getCountNegative()) == (0, 5, 6, 1, 5, 0)
    assert isinstance(ai.getMin(), int) and ai.getCountMinValue() == 1 and ai.getCountMaxValue() == 1
    assert abs(ai.getMean() - 2.5) < 1e-12 and abs(ai.getSampleStdev() - np.std(np.arange(6), ddof=1)) < 1e-12
    v = np.array([1.5, -2.0, 0.0, 7.25, 0.0])
    assert ax.getCountNaN() == 1 and ax.getCountZero() == 2 and ax.getCountNegative() == 1 and ax.getCountPositive() == 2
    assert ax.getMin() == -2.0 and ax.getMax() == 7.25 and abs(ax.getMean() - v.mean()) < 1e-12
    assert abs(ax.getSampleVariance() - v.var(ddof=1)) < 1e-12
    assert len(ax.getHistogramBuckets()) == 5 and sum(ax.getHistogramBucketCounts()) == 5
    assert abs(ax.getQuantile(0.5) - np.quantile(v, 0.5)) < 1e-9
    assert ak.getMapOfCounts() == {"a": 3, "b": 2, "c": 1}
    assert an_.getCountUnique() == 6 and an_.getMinLength() == 0 and an_.getMaxLength() == 5
    assert abs(an_.getMeanLength() - np.mean([5, 0, 2, 1, 2, 3])) < 1e-12
    assert "IntegerAnalysis(min=0,max=5" in str(an) and '"kind"' in str(an)


def test_data_analysis_json_yaml_round_trip():
    an = AnalyzeLocal.analyze(_schema(), _rows())
    for back in (DataAnalysis.fromJson(an.toJson()), DataAnalysis.fromYaml(an.toYaml())):
        assert [c.name for c in back.getSchema().columns] == ["id", "x", "kind", "note"]
        for a, b in zip(an.getColumnAnalysis(), back.getColumnAnalysis()):
            assert type(a) is type(b)
            da, db = a.to_dict(), b.to_dict()
            for k in da:
                if isinstance(da[k], float) and math.isnan(da[k]):
                    assert math.isnan(db[k])
                elif k == "quantiles":
                    assert np.allclose(da[k], db[k])
                else:
                    assert da[k] == db[k], k
    assert json.loads(an.toJson())["DataAnalysis"][1]["analysis"]["countNaN"] == 1


def test_bytes_and_ndarray_analysis():
    s = Schema.Builder().addColumnBytes("b").addColumnNDArray("arr", [2, 2]).build()
    rows = [[BytesWritable(b"abc"), NDArrayWritable(np.array([[1.0, 2.0], [3.0, 4.0]]))],
            [BytesWritable(b""), NDArrayWritable(np.array([-1.0, 9.0, 0.5]))],
            [BytesWritable(b"0123456789"), NullWritable()]]
    an = AnalyzeLocal.analyze(s, rows)
    b, a = an.getColumnAnalysis("b"), an.getColumnAnalysis("arr")
    assert (b.countZeroLength, b.minNumBytes, b.maxNumBytes, b.count) == (1, 0, 10, 3)
    assert (a.countNull, a.minLength, a.maxLength, a.totalNDArrayValues) == (1, 3, 4, 7)
    assert a.countsByRank == {2: 1, 1: 1} and a.minValue == -1.0 and a.maxValue == 9.0
    assert sum(a.histogramCounts) == 7


def test_quality_analysis_rules_and_report():
    s = (Schema.Builder().addColumnInteger("i", 0, 10).addColumnDouble("d").addColumnCategorical("c", ["a", "b"])
         .addColumnString("s").build())
    rows = [[IntWritable(3), DoubleWritable(1.0), Text("a"), Text("abc")],
            [Text("x"), Text("nope"), Text("z"), Text("")],                 # non-integer / non-real / unknown state / empty
            [IntWritable(99), DoubleWritable(float("nan")), Text("b"), Text("123")],     # out of range, NaN
            [Text(""), DoubleWritable(float("inf")), Text("a"), Text("   ")],            # missing int, infinite
            [NullWritable(), NullWritable(), NullWritable(), NullWritable()]]
    q = AnalyzeLocal.analyzeQuality(s, rows)
    assert isinstance(q, DataQualityAnalysis)
    qi, qd, qc, qs = (q.getColumnQuality(n) for n in ("i", "d", "c", "s"))
    assert isinstance(qi, IntegerQuality) and isinstance(qd, DoubleQuality) and isinstance(qs, StringQuality)
    assert (qi.countValid, qi.countInvalid, qi.countMissing, qi.countTotal, qi.countNonInteger) == (1, 2, 2, 5, 3)
    assert (qd.countValid, qd.countInvalid, qd.countMissing, qd.countTotal) == (1, 3, 1, 5)
    assert (qd.countNonReal, qd.countNaN, qd.countInfinite) == (2, 1, 1)          # "nope" and the NullWritable's "" do not parse
    assert (qc.countValid, qc.countInvalid, qc.countMissing) == (3, 1, 1)
    assert (qs.countMissing, qs.countEmptyString, qs.countNumerical, qs.countWordCharacter, qs.countWhitespace) == (1, 2, 1, 2, 1)
    assert qs.countAlphabetic == 0 and qs.countApproxUnique == 4                   # "[a-zA-Z]" matches single letters only
    rep = str(q)
    assert rep.splitlines()[0].split() == ["idx", "name", "type", "quality", "details"]
    assert "FAIL" in rep and 'DoubleQuality(countValid=1, countInvalid=3, countMissing=1, countTotal=5, countNonReal=2, countNaN=1, countInfinite=1)' in rep
    # add(): counters of two partitions combine (QualityAnalysisCombineFunction)
    both = qd.add(qd)
    assert both.countTotal == 10 and both.countNaN == 2 and qs.add(qs).countApproxUnique == 4
    clean = AnalyzeLocal.analyzeQuality(s, rows[:1])
    assert "FAIL" not in str(clean) and clean["i"] == {"valid": 1, "invalid": 0, "missing": 0, "total": 1}


def test_sequence_analysis_and_html_reports(tmp_path):
    s = Schema.Builder().addColumnDouble("v").addColumnCategorical("state", ["on", "off"]).build()
    seqs = [[[DoubleWritable(float(t)), Text("on" if t % 2 else "off")] for t in range(n)] for n in (1, 4, 4, 0, 7)]
    sa = AnalyzeLocal.analyzeSequence(s, seqs)
    assert isinstance(sa, SequenceDataAnalysis)
    la = sa.getSequenceLengthAnalysis()
    assert (la.totalNumSequences, la.minSeqLength, la.maxSeqLength, la.countZeroLength, la.countOneLength) == (5, 0, 7, 1, 1)
    assert abs(la.meanLength - 3.2) < 1e-12 and sa.getColumnAnalysis("v").getCountTotal() == 16
    page = HtmlAnalysis.createHtmlAnalysisString(sa)
    assert page.startswith("<!DOCTYPE html>") and page.count("<svg") >= 3 and "Sequence lengths" in page and "DoubleAnalysis(" in page
    out = tmp_path / "analysis.html"
    HtmlAnalysis.createHtmlAnalysisFile(AnalyzeLocal.analyze(_schema(), _rows()), str(out))
    txt = out.read_text()
    assert "<h2 id='col1'>x" in txt and "countNaN" in txt and "&quot;" not in txt.split("<h2")[0].replace("&quot;", "")
    plots = HtmlSequencePlotting.createHtmlSequencePlots("seq 4", s, seqs[4])
    assert plots.count("<polyline") == 2 and "<h1>seq 4</h1>" in plots
    assert AnalyzeLocal.analyzeQualitySequence(s, seqs).getColumnQuality("v").countTotal == 16
    assert [w.toString() for w in AnalyzeLocal.getUnique("state", s, seqs[4])] == ["off", "on"]


def test_calculate_sorted_rank_follows_reference_executor():
    s = Schema.Builder().addColumnString("name").addColumnDouble("score").build()
    rows = [[Text("b"), DoubleWritable(2.5)], [Text("d"), DoubleWritable(-1.0)], [Text("a"), DoubleWritable(10.0)], [Text("c"), DoubleWritable(0.0)]]
    tp = TransformProcess.Builder(s).calculateSortedRank("rank", "score", DoubleWritableComparator()).build()
    assert tp.getFinalSchema().getColumnNames() == ["name", "score", "rank"] and tp.getFinalSchema().getType(2) == ColumnType.Long
    out = LocalTransformExecutor.execute(rows, tp)
    # output order = sorted order; rank = position
    assert [(r[0].toString(), r[2].toLong()) for r in out] == [("d", 0), ("c", 1), ("b", 2), ("a", 3)]
    desc = LocalTransformExecutor.execute(rows, TransformProcess.Builder(s).calculateSortedRank("rank", "score", DoubleWritableComparator(), False).build())
    assert [r[0].toString() for r in desc] == ["a", "b", "c", "d"] and [r[2].toLong() for r in desc] == [0, 1, 2, 3]
    by_name = LocalTransformExecutor.execute(rows, TransformProcess.Builder(s).calculateSortedRank("r", "name", ReverseComparator(TextWritableComparator())).build())
    assert [r[0].toString() for r in by_name] == ["d", "c", "b", "a"]
    # default comparator from the column type; bool in the comparator slot = ascending (earlier signature)
    dflt = LocalTransformExecutor.execute(rows, TransformProcess.Builder(s).calculateSortedRank("r", "score", False).build())
    assert [r[0].toString() for r in dflt] == ["a", "b", "c", "d"]
    op = CalculateSortedRank("rank", "score", DoubleWritableComparator())
    op.setInputSchema(s)
    assert op.outputColumnName() == "rank" and op.outputColumnNames() == ["name", "score", "rank"] and op.columnName() == "name"
    assert CalculateSortedRank.from_json(json.loads(json.dumps(op.to_json()))) == op
    assert 'CalculateSortedRank(newColumnName="rank"' in repr(op)
    js = json.loads(tp.toJson())
    assert js["steps"][0]["@class"].endswith("CalculateSortedRank") and js["steps"][0]["comparator"]["@class"] == "DoubleWritableComparator"
    with pytest.raises(Exception):
        TransformProcess.Builder(s).calculateSortedRank("r", "missing").build()
