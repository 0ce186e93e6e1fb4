"""Arrow / Parquet interchange for DataVec records (reference: datavec/datavec-arrow ArrowConverter, ArrowRecordReader,
ArrowRecordWriter, ArrowWritableRecordBatch; nd4j-serde/nd4j-arrow tensor conversion).

Columnar batches are the natural staging format for the GPU input pipeline: a numeric Arrow table converts to one
[rows, cols] tensor without per-record Python objects (``toTensor``), which is what the RecordReaderDataSetIterator
fast path consumes.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .records import InputSplit, RecordReader
from .transform import ColumnType, Schema
from .writable import (BooleanWritable, BytesWritable, DoubleWritable, FloatWritable, IntWritable, LongWritable, NDArrayWritable,
                       NullWritable, Text, Writable, wrap)


def _pa():
    import pyarrow as pa
    return pa


def _arrow_type(ct):
    pa = _pa()
    return {ColumnType.Integer: pa.int32(), ColumnType.Long: pa.int64(), ColumnType.Double: pa.float64(), ColumnType.Float: pa.float32(),
            ColumnType.Boolean: pa.bool_(), ColumnType.Time: pa.int64(), ColumnType.String: pa.string(), ColumnType.Categorical: pa.string(),
            ColumnType.NDArray: pa.binary(), ColumnType.Bytes: pa.binary() if hasattr(ColumnType, "Bytes") else pa.binary()}.get(ct, pa.string())


def _py(w, ct):
    if isinstance(w, NullWritable) or w is None:
        return None
    if ct in (ColumnType.Integer, ColumnType.Long, ColumnType.Time):
        return w.toLong()
    if ct in (ColumnType.Double, ColumnType.Float):
        return w.toDouble()
    if ct == ColumnType.Boolean:
        return bool(w.get()) if isinstance(w, BooleanWritable) else w.toString().lower() == "true"
    if ct == ColumnType.NDArray:
        import io
        buf = io.BytesIO()
        np.save(buf, torch.as_tensor(w.get()).cpu().numpy(), allow_pickle=False)
        return buf.getvalue()
    if isinstance(w, BytesWritable):
        return w.get()
    return w.toString()


def _writable(v, ct):
    if v is None:
        return NullWritable()
    if ct == ColumnType.Integer:
        return IntWritable(int(v))
    if ct in (ColumnType.Long, ColumnType.Time):
        return LongWritable(int(v))
    if ct == ColumnType.Double:
        return DoubleWritable(float(v))
    if ct == ColumnType.Float:
        return FloatWritable(float(v))
    if ct == ColumnType.Boolean:
        return BooleanWritable(bool(v))
    if ct == ColumnType.NDArray:
        import io
        return NDArrayWritable(torch.from_numpy(np.load(io.BytesIO(v), allow_pickle=False)))
    return Text(str(v))


class ArrowConverter:
    @staticmethod
    def toArrowSchema(schema: Schema):
        pa = _pa()
        return pa.schema([pa.field(c.name, _arrow_type(c.type)) for c in schema.columns])

    @staticmethod
    def toDatavecSchema(arrow_schema) -> Schema:
        pa = _pa()
        b = Schema.Builder()
        for f in arrow_schema:
            t = f.type
            if pa.types.is_int64(t) or pa.types.is_uint64(t):
                b.addColumnLong(f.name)
            elif pa.types.is_integer(t):
                b.addColumnInteger(f.name)
            elif pa.types.is_float32(t) or pa.types.is_float16(t):
                b.addColumnFloat(f.name)
            elif pa.types.is_floating(t):
                b.addColumnDouble(f.name)
            elif pa.types.is_boolean(t):
                b.addColumnBoolean(f.name)
            elif pa.types.is_binary(t) or pa.types.is_large_binary(t):
                b.addColumnNDArray(f.name)
            else:
                b.addColumnString(f.name)
        return b.build()

    @staticmethod
    def toArrowTable(schema: Schema, records: Sequence[Sequence[Writable]]):
        pa = _pa()
        cols = []
        for j, c in enumerate(schema.columns):
            cols.append(pa.array([_py(wrap(r[j]), c.type) for r in records], type=_arrow_type(c.type)))
        return pa.Table.from_arrays(cols, schema=ArrowConverter.toArrowSchema(schema))

    @staticmethod
    def toRecords(table, schema: Schema = None) -> List[List[Writable]]:
        schema = schema or ArrowConverter.toDatavecSchema(table.schema)
        cols = [table.column(j).to_pylist() for j in range(table.num_columns)]
        types = schema.getColumnTypes()
        return [[_writable(cols[j][i], types[j]) for j in range(len(cols))] for i in range(table.num_rows)]

    @staticmethod
    def toTensor(table, columns: Sequence[str] = None, dtype=torch.float32) -> torch.Tensor:
        """Numeric columns -> one [rows, cols] tensor, column-wise (no per-record objects)."""
        names = list(columns) if columns else table.schema.names
        arrs = [np.asarray(table.column(n).to_numpy(zero_copy_only=False), dtype=np.float64) for n in names]
        return torch.from_numpy(np.stack(arrs, 1)).to(dtype)

    @staticmethod
    def writeRecordBatchTo(records, schema: Schema, path, format="arrow"):
        pa = _pa()
        t = ArrowConverter.toArrowTable(schema, records)
        if format == "parquet":
            import pyarrow.parquet as pq
            pq.write_table(t, str(path))
        else:
            with pa.OSFile(str(path), "wb") as sink, pa.ipc.new_file(sink, t.schema) as w:
                w.write_table(t)
        return path

    @staticmethod
    def readFromFile(path):
        pa = _pa()
        p = str(path)
        if p.endswith(".parquet"):
            import pyarrow.parquet as pq
            return pq.read_table(p)
        with pa.memory_map(p, "rb") as src:
            return pa.ipc.open_file(src).read_all()


class ArrowRecordWriter:
    def __init__(self, schema: Schema, format="arrow"):
        self.schema, self.format, self._rows, self._path = schema, format, [], None

    def initialize(self, path):
        self._path = path
        return self

    def write(self, record):
        self._rows.append(list(record))

    def writeBatch(self, records):
        self._rows.extend(list(r) for r in records)

    def close(self):
        if self._path is not None:
            ArrowConverter.writeRecordBatchTo(self._rows, self.schema, self._path, self.format)
            self._rows = []


class ArrowRecordReader(RecordReader):
    """Reads every .arrow / .parquet file of the split, one record per row."""

    def __init__(self):
        super().__init__()
        self._records, self._pos, self.schema = [], 0, None

    def reset(self):
        self._records, self._pos = [], 0
        for loc in self.split.locations():
            t = ArrowConverter.readFromFile(loc)
            self.schema = self.schema or ArrowConverter.toDatavecSchema(t.schema)
            self._records.extend(ArrowConverter.toRecords(t, self.schema))

    def hasNext(self):
        return self._pos < len(self._records)

    def next(self):
        r = self._records[self._pos]
        self._pos += 1
        return r

    def batchesSupported(self):
        return True
