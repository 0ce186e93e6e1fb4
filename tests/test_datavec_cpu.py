"""DataVec record readers / TransformProcess / image pipeline / DataSet bridges.  Reference tests: CSVRecordReaderTest,
SVMLightRecordReaderTest, TestTransforms, TestReduce, TestJoin, TestSequenceTransforms, RecordReaderDataSetiteratorTest."""
import json

import numpy as np
import pytest
import torch

from deeplearning4j_amd.datasets.record_reader import (RecordReaderDataSetIterator, RecordReaderMultiDataSetIterator,
                                                       SequenceRecordReaderDataSetIterator)
from deeplearning4j_amd.datavec import (AnalyzeLocal, BooleanCondition, CollectionRecordReader, CollectionSequenceRecordReader, ColumnType,
                                        ConditionFilter, ConditionOp, CSVRecordReader, CSVSequenceRecordReader, DoubleColumnCondition,
                                        DoubleWritable, FileSplit, IntWritable, JacksonLineRecordReader, Join, LocalTransformExecutor, MathOp,
                                        NumberedFileInputSplit, NumericalColumnComparator, Normalize, ReduceOp, Reducer, RegexLineRecordReader,
                                        Schema, StringSplit, SVMLightRecordReader, Text, TransformProcess, TransformProcessRecordReader,
                                        CategoricalColumnCondition)
from deeplearning4j_amd.datavec.image import (CropImageTransform, FlipImageTransform, ImageRecordReader, NativeImageLoader,
                                              ParentPathLabelGenerator, PipelineImageTransform, ResizeImageTransform)

IRIS = "5.1,3.5,1.4,0.2,0\n4.9,3.0,1.4,0.2,0\n7.0,3.2,4.7,1.4,1\n6.4,3.2,4.5,1.5,1\n6.3,3.3,6.0,2.5,2\n5.8,2.7,5.1,1.9,2\n"


def test_csv_reader_and_dataset_iterator(tmp_path):
    p = tmp_path / "iris.csv"
    p.write_text("a,b,c,d,label\n" + IRIS)
    rr = CSVRecordReader(1).initialize(FileSplit(p))
    first = rr.next()
    assert [w.toString() for w in first] == ["5.1", "3.5", "1.4", "0.2", "0"] and first[0].toDouble() == 5.1
    rr.reset()
    it = RecordReaderDataSetIterator(rr, 4, 4, 3)
    b1 = it.next()
    assert b1.features.shape == (4, 4) and b1.labels.tolist() == [[1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 1, 0]]
    b2 = it.next()
    assert b2.features.shape == (2, 4) and not it.hasNext()
    it.reset()
    assert it.hasNext()
    m = rr.loadAllNumeric()                                   # native bulk parser
    assert m.shape == (6, 5) and abs(float(m[2, 0]) - 7.0) < 1e-6 and float(m[5, 4]) == 2.0
    with pytest.raises(ValueError):
        RecordReaderDataSetIterator(CSVRecordReader(1).initialize(FileSplit(p)), 6, 4, 2).next()   # label 2 out of range
    reg = RecordReaderDataSetIterator.Builder(CSVRecordReader(1).initialize(FileSplit(p)), 6).regression(2, 3).build().next()
    assert reg.features.shape == (6, 3) and reg.labels.shape == (6, 2) and abs(float(reg.labels[0, 0]) - 1.4) < 1e-6
    # quoted fields / alternative delimiter through the python path
    q = CSVRecordReader(0, ";").initialize(StringSplit('1;"x;y";3\n'))
    assert [w.toString() for w in q.next()] == ["1", "x;y", "3"]


def test_svmlight_regex_json_readers(tmp_path):
    p = tmp_path / "d.svm"
    p.write_text("1 1:0.5 3:2.0 # c\n0 2:1.5\n")
    rr = SVMLightRecordReader(numFeatures=4).initialize(FileSplit(p))
    assert [w.toDouble() for w in rr.next()] == [0.5, 0.0, 2.0, 0.0, 1.0]
    assert [w.toDouble() for w in rr.next()] == [0.0, 1.5, 0.0, 0.0, 0.0]
    m = rr.loadAllNumeric()
    assert m.shape[0] == 2 and m[0, :4].tolist() == [0.5, 0.0, 2.0, 0.0]
    rx = RegexLineRecordReader(r"(\d+)-(\w+) (.*)").initialize(StringSplit("12-ab hello world\n7-z x"))
    assert [w.toString() for w in rx.next()] == ["12", "ab", "hello world"]
    js = JacksonLineRecordReader(["a", "b.c", "missing"], [None, None, "dflt"]).initialize(StringSplit('{"a": 1, "b": {"c": "x"}}\n'))
    r = js.next()
    assert r[0] == IntWritable(1) and r[1] == Text("x") and r[2] == Text("dflt")
    assert NumberedFileInputSplit("f_%d.csv", 2, 4).locations() == ["f_2.csv", "f_3.csv", "f_4.csv"]
    tr, te = FileSplit(tmp_path).sample([0.5, 0.5], 1)
    assert tr.length() + te.length() == 1


def _schema():
    return (Schema.Builder().addColumnString("id").addColumnCategorical("kind", "a", "b", "c").addColumnDouble("x").addColumnInteger("n").build())


def _rows():
    return [[Text("r0"), Text("a"), DoubleWritable(1.0), IntWritable(3)], [Text("r1"), Text("b"), DoubleWritable(-2.0), IntWritable(4)],
            [Text("r2"), Text("c"), DoubleWritable(5.0), IntWritable(5)], [Text("r3"), Text("a"), DoubleWritable(float("nan")), IntWritable(6)]]


def test_transform_process_columns_math_categorical():
    s = _schema()
    tp = (TransformProcess.Builder(s).filter(DoubleColumnCondition("x", ConditionOp.LessThan, 0.0))
          .filter(ConditionFilter(BooleanCondition.OR(DoubleColumnCondition("x", ConditionOp.Equal, 123456.0),
                                                      DoubleColumnCondition("x", ConditionOp.GreaterThan, 1e9))))
          .doubleMathOp("x", MathOp.Multiply, 2.0).integerMathOp("n", MathOp.Add, 10)
          .doubleColumnsMathOp("x+n", MathOp.Add, "x", "n").categoricalToOneHot("kind").renameColumn("id", "key").removeColumns("n")
          .build())
    fs = tp.getFinalSchema()
    assert fs.getColumnNames() == ["key", "kind[a]", "kind[b]", "kind[c]", "x", "x+n"]
    assert fs.getType("x+n") == ColumnType.Double and fs.getType("kind[a]") == ColumnType.Integer
    out = LocalTransformExecutor.execute(_rows()[:3], tp)
    assert len(out) == 2                                                  # the x<0 row was filtered
    assert [w.toString() for w in out[0][:4]] == ["r0", "1", "0", "0"] and out[0][4].toDouble() == 2.0 and out[0][5].toDouble() == 15.0
    assert [w.toInt() for w in out[1][1:4]] == [0, 0, 1] and out[1][5].toDouble() == 25.0
    assert tp.execute(_rows()[1]) is None and tp.execute(_rows()[0])[4].toDouble() == 2.0
    json.loads(tp.toJson())
    with pytest.raises(KeyError):
        TransformProcess.Builder(s).removeColumns("nope").build()
    tp2 = TransformProcess.Builder(s).categoricalToInteger("kind").conditionalReplaceValueTransform(
        "x", 0.0, DoubleColumnCondition("x", ConditionOp.LessThan, 0.0)).stringToCategorical("id", ["r0", "r1", "r2"]).build()
    o2 = LocalTransformExecutor.execute(_rows()[:3], tp2)
    assert [r[1].toInt() for r in o2] == [0, 1, 2] and [r[2].toDouble() for r in o2] == [1.0, 0.0, 5.0]
    assert tp2.getFinalSchema().getMetaData("id").stateNames == ["r0", "r1", "r2"]
    rr = TransformProcessRecordReader(CollectionRecordReader(_rows()[:3]), tp).initialize(None)
    assert len(list(rr)) == 2


def test_analysis_normalize_reduce_join():
    s = _schema()
    an = AnalyzeLocal.analyze(s, _rows())
    ax = an.getColumnAnalysis("x")
    assert ax.countMissing == 1 and ax.min == -2.0 and ax.max == 5.0 and abs(ax.mean - 4.0 / 3) < 1e-12
    assert an.getColumnAnalysis("kind").stateCounts == {"a": 2, "b": 1, "c": 1}
    tp = TransformProcess.Builder(s).normalize("x", Normalize.MinMax, an).normalize("n", Normalize.Standardize, an).build()
    out = LocalTransformExecutor.execute(_rows()[:3], tp)
    assert [round(r[2].toDouble(), 6) for r in out] == [round(3 / 7, 6), 0.0, 1.0]
    assert abs(sum(r[3].toDouble() for r in LocalTransformExecutor.execute(_rows(), tp))) < 1e-9
    red = (Reducer.Builder(ReduceOp.TakeFirst).keyColumns("kind").sumColumns("n").meanColumns("x").countColumns("id").build())
    tpr = TransformProcess.Builder(s).filter(CategoricalColumnCondition("id", ConditionOp.Equal, "r3")).reduce(red).build()
    assert tpr.getFinalSchema().getColumnNames() == ["count(id)", "kind", "mean(x)", "sum(n)"]
    rows = _rows() + [[Text("r4"), Text("a"), DoubleWritable(3.0), IntWritable(1)]]
    ro = LocalTransformExecutor.execute(rows, tpr)
    byk = {r[1].toString(): r for r in ro}
    assert byk["a"][0].toInt() == 2 and byk["a"][2].toDouble() == 2.0 and byk["a"][3].toDouble() == 4.0 and len(ro) == 3
    left = Schema.Builder().addColumnInteger("k").addColumnString("l").build()
    right = Schema.Builder().addColumnInteger("k").addColumnDouble("r").build()
    L = [[IntWritable(1), Text("x")], [IntWritable(2), Text("y")]]
    R = [[IntWritable(2), DoubleWritable(0.5)], [IntWritable(3), DoubleWritable(0.7)]]
    j = Join.Builder(Join.Inner).setSchemas(left, right).setJoinColumns("k").build()
    assert j.outputSchema().getColumnNames() == ["k", "l", "r"]
    assert [[w.toString() for w in r] for r in j.execute(L, R)] == [["2", "y", "0.5"]]
    full = Join.Builder(Join.FullOuter).setSchemas(left, right).setJoinColumns("k").build().execute(L, R)
    assert len(full) == 3 and full[0][2].toString() == "" and full[2][0].toInt() == 3


def test_sequence_transforms_and_iterators(tmp_path):
    s = Schema.Builder().addColumnString("dev").addColumnInteger("t").addColumnDouble("v").build()
    rows = [[Text("A"), IntWritable(2), DoubleWritable(2.0)], [Text("B"), IntWritable(1), DoubleWritable(10.0)],
            [Text("A"), IntWritable(1), DoubleWritable(1.0)], [Text("A"), IntWritable(3), DoubleWritable(4.0)]]
    tp = (TransformProcess.Builder(s).convertToSequence("dev", NumericalColumnComparator("t"))
          .sequenceMovingWindowReduce("v", 2, ReduceOp.Mean).build())
    seqs = LocalTransformExecutor.executeToSequence(rows, tp)
    assert len(seqs) == 2 and [r[1].toInt() for r in seqs[0]] == [1, 2, 3]
    assert [r[3].toDouble() for r in seqs[0]] == [1.0, 1.5, 3.0]
    off = TransformProcess.Builder(s).convertToSequence("dev", NumericalColumnComparator("t")).offsetSequence(["v"], 1).build()
    o = LocalTransformExecutor.executeToSequence(rows, off)
    assert len(o) == 1 and [r[2].toDouble() for r in o[0]] == [1.0, 2.0] and [r[1].toInt() for r in o[0]] == [2, 3]
    # CSV sequences with unequal lengths -> padded + masked
    for i, n in enumerate((3, 2)):
        (tmp_path / ("f_%d.csv" % i)).write_text("".join("%d,%d\n" % (t, t + i) for t in range(n)))
        (tmp_path / ("l_%d.csv" % i)).write_text("".join("%d\n" % (t % 2) for t in range(n)))
    fr = CSVSequenceRecordReader().initialize(NumberedFileInputSplit(str(tmp_path / "f_%d.csv"), 0, 1))
    lr = CSVSequenceRecordReader().initialize(NumberedFileInputSplit(str(tmp_path / "l_%d.csv"), 0, 1))
    ds = SequenceRecordReaderDataSetIterator(fr, lr, 2, 2, alignmentMode="ALIGN_START").next()
    assert ds.features.shape == (2, 2, 3) and ds.labels.shape == (2, 2, 3)
    assert ds.featuresMask.tolist() == [[1, 1, 1], [1, 1, 0]] and ds.labels[0, :, 1].tolist() == [0, 1]
    single = SequenceRecordReaderDataSetIterator(CollectionSequenceRecordReader([[[0.5, 1], [0.7, 0]]]), 1, 2, 1).next()
    assert single.features.shape == (1, 1, 2) and single.labels[0, :, 0].tolist() == [0, 1]
    multi = (RecordReaderMultiDataSetIterator.Builder(2).addReader("r", CollectionRecordReader([[1, 2, 0], [3, 4, 1]]))
             .addInput("r", 0, 1).addOutputOneHot("r", 2, 2).build().next())
    assert multi.features[0].tolist() == [[1, 2], [3, 4]] and multi.labels[0].tolist() == [[1, 0], [0, 1]]


def test_image_pipeline(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    for lab, colour in (("red", (255, 0, 0)), ("blue", (0, 0, 255))):
        d = tmp_path / lab
        d.mkdir()
        for k in range(3):
            a = np.zeros((20, 30, 3), dtype=np.uint8)
            a[:, :] = colour
            a[0, 0] = rng.integers(0, 255, 3)
            Image.fromarray(a).save(d / ("img%d.png" % k))
    rr = ImageRecordReader(16, 16, 3, ParentPathLabelGenerator()).initialize(FileSplit(tmp_path, ["png"]))
    assert rr.getLabels() == ["blue", "red"]
    rec = rr.next()
    img = rec[0].value
    assert img.shape == (1, 3, 16, 16) and rec[1].toInt() == 0
    assert float(img[0, 0].mean()) > 200 and float(img[0, 2].mean()) < 50         # BGR order: blue image -> channel 0 bright
    rr.reset()
    it = RecordReaderDataSetIterator(rr, 4, 1, 2)
    ds = it.next()
    assert ds.features.shape == (4, 3, 16, 16) and ds.labels.sum(0).tolist() == [3, 1]
    loader = NativeImageLoader(10, 10, 1)
    g = loader.asMatrix(tmp_path / "red" / "img0.png")
    assert g.shape == (1, 1, 10, 10)
    t = torch.arange(2 * 4 * 6, dtype=torch.float32).reshape(2, 4, 6)
    import random
    assert torch.equal(FlipImageTransform(1).transform(t, random.Random(0)), t.flip(2))
    assert CropImageTransform(1, 2, 1, 2, random_=False).transform(t, random.Random(0)).shape == (2, 2, 2)
    pipe = PipelineImageTransform(ResizeImageTransform(8, 8), (FlipImageTransform(0), 1.0))
    assert pipe.transform(t, random.Random(0)).shape == (2, 8, 8)
    back = NativeImageLoader.toImage(img)
    assert back.size == (16, 16) and back.getpixel((8, 8)) == (0, 0, 255)


def test_arrow_jdbc_excel_readers(tmp_path):
    import sqlite3
    from deeplearning4j_amd.datavec.arrow import ArrowConverter, ArrowRecordReader, ArrowRecordWriter
    from deeplearning4j_amd.datavec.jdbc import JDBCRecordReader
    from deeplearning4j_amd.datavec.excel import ExcelRecordReader, ExcelRecordWriter
    from deeplearning4j_amd.datavec.records import CollectionInputSplit
    from deeplearning4j_amd.datavec.transform import Schema
    from deeplearning4j_amd.datavec.writable import DoubleWritable, IntWritable, Text, BooleanWritable
    schema = Schema.Builder().addColumnInteger("id").addColumnDouble("x").addColumnString("name").addColumnBoolean("flag").build()
    recs = [[IntWritable(i), DoubleWritable(i * 0.5), Text("n%d" % i), BooleanWritable(i % 2 == 0)] for i in range(7)]
    # arrow + parquet round trips
    for fmt, ext in (("arrow", ".arrow"), ("parquet", ".parquet")):
        p = tmp_path / ("d" + ext)
        w = ArrowRecordWriter(schema, fmt).initialize(p)
        w.writeBatch(recs)
        w.close()
        rr = ArrowRecordReader().initialize(CollectionInputSplit([str(p)]))
        back = list(rr)
        assert [[v.get() for v in r] for r in back] == [[v.get() for v in r] for r in recs]
        assert rr.schema.getColumnNames() == ["id", "x", "name", "flag"]
    t = ArrowConverter.toArrowTable(schema, recs)
    m = ArrowConverter.toTensor(t, ["id", "x"])
    assert m.shape == (7, 2) and float(m[3, 1]) == 1.5
    # jdbc over sqlite
    db = tmp_path / "t.db"
    conn = sqlite3.connect(db)
    conn.execute("create table coffee (id integer primary key, name text, price real)")
    conn.executemany("insert into coffee values (?, ?, ?)", [(i, " c%d " % i, 1.0 + i) for i in range(5)])
    conn.commit()
    jr = JDBCRecordReader("select * from coffee order by id", conn, "select * from coffee where id = ?", [0])
    jr.setTrimStrings(True)
    jr.initialize()
    rows = []
    while jr.hasNext():
        rows.append(jr.nextRecord())
    assert len(rows) == 5 and rows[2].getRecord()[1].toString() == "c2" and rows[4].getRecord()[2].toDouble() == 5.0
    again = jr.loadFromMetaData(rows[3].getMetaData())
    assert again.getRecord()[0].toInt() == 3
    jr.reset()
    assert jr.hasNext() and jr.columnNames == ["id", "name", "price"]
    # excel
    xp = tmp_path / "b.xlsx"
    ew = ExcelRecordWriter().initialize(xp)
    ew.write(["id", "x", "name", "flag"])
    ew.writeBatch(recs)
    ew.close()
    er = ExcelRecordReader(skipNumLines=1).initialize(CollectionInputSplit([str(xp)]))
    rows = list(er)
    assert len(rows) == 7 and rows[3][0].toInt() == 3 and rows[3][1].toDouble() == 1.5 and rows[3][2].toString() == "n3" and rows[2][3].get() is True
