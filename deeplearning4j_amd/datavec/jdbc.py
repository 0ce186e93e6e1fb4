"""JDBC-style record reader over any Python DB-API 2.0 connection (reference: datavec/datavec-jdbc JDBCRecordReader:
query + optional metadata query, ``initialize(InputSplit)`` runs the query, records are rows; ``resetQuery``).
sqlite3 ships with Python and is what the tests use; any driver object with ``cursor()`` works."""
from __future__ import annotations

import sqlite3
from typing import Optional, Sequence

from .records import Record, RecordMetaData, RecordReader
from .writable import wrap


class JDBCRecordReader(RecordReader):
    TRIM_STRINGS, JDBC_URL = "datavec.jdbc.trim_strings", "datavec.jdbc.url"

    def __init__(self, query: str, dataSource=None, metadataQuery: Optional[str] = None, metadataIndices: Optional[Sequence[int]] = None):
        super().__init__()
        self.query, self.metadataQuery, self.metadataIndices = query, metadataQuery, list(metadataIndices or [])
        self.conn = sqlite3.connect(dataSource) if isinstance(dataSource, str) else dataSource
        self.trimStrings = False
        self._cur = None
        self._peek = None
        self._idx = 0
        self.columnNames = []

    def setTrimStrings(self, b):
        self.trimStrings = bool(b)

    def initialize(self, split=None, conf: Optional[dict] = None):
        if conf:
            if conf.get(self.JDBC_URL) and self.conn is None:
                self.conn = sqlite3.connect(conf[self.JDBC_URL].replace("jdbc:sqlite:", ""))
            self.trimStrings = bool(conf.get(self.TRIM_STRINGS, self.trimStrings))
        if self.conn is None:
            raise ValueError("JDBCRecordReader needs a data source (connection, sqlite path or %s)" % self.JDBC_URL)
        self.split = split
        self.reset()
        return self

    def reset(self):
        self._cur = self.conn.cursor()
        self._cur.execute(self.query)
        self.columnNames = [d[0] for d in (self._cur.description or [])]
        self._peek = self._cur.fetchone()
        self._idx = 0

    def hasNext(self):
        return self._peek is not None

    def _convert(self, row):
        out = []
        for v in row:
            if isinstance(v, str) and self.trimStrings:
                v = v.strip()
            out.append(wrap(v))
        return out

    def next(self):
        if self._peek is None:
            raise StopIteration
        row = self._peek
        self._peek = self._cur.fetchone()
        self._idx += 1
        return self._convert(row)

    def nextRecord(self):
        row = self._peek
        rec = self.next()
        params = [row[i] for i in self.metadataIndices] if self.metadataIndices else [self._idx - 1]
        md = RecordMetaData("jdbc:" + (self.metadataQuery or self.query), self._idx - 1)
        md.params = params
        return Record(rec, md)

    def loadFromMetaData(self, md):
        if not self.metadataQuery:
            raise ValueError("loadFromMetaData needs a metadataQuery")
        cur = self.conn.cursor()
        cur.execute(self.metadataQuery, list(md.params))
        row = cur.fetchone()
        return Record(self._convert(row), md) if row is not None else None

    def close(self):
        if self._cur is not None:
            self._cur.close()
