"""Excel (.xlsx) record reader / writer without third-party packages (reference: datavec/datavec-excel
ExcelRecordReader — one record per row of every sheet, ``skipNumLines`` header rows — and ExcelRecordWriter, both built
on Apache POI there). An .xlsx file is a zip of SpreadsheetML parts: workbook.xml lists the sheets,
sharedStrings.xml holds the string table, sheetN.xml the cells; that is all this module reads and writes."""
from __future__ import annotations

import re
import zipfile
from typing import List
from xml.etree import ElementTree as ET
from xml.sax.saxutils import escape

from .records import RecordReader
from .writable import BooleanWritable, DoubleWritable, IntWritable, NullWritable, Text, Writable

_NS = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main"}


def _col_index(ref: str) -> int:
    n = 0
    for ch in re.match(r"[A-Z]+", ref).group(0):
        n = n * 26 + (ord(ch) - 64)
    return n - 1


def _col_name(i: int) -> str:
    s = ""
    i += 1
    while i:
        i, r = divmod(i - 1, 26)
        s = chr(65 + r) + s
    return s


def read_xlsx(path) -> List[List[List[Writable]]]:
    """All sheets -> rows -> Writables."""
    with zipfile.ZipFile(str(path)) as z:
        names = z.namelist()
        shared = []
        if "xl/sharedStrings.xml" in names:
            root = ET.fromstring(z.read("xl/sharedStrings.xml"))
            for si in root.findall("m:si", _NS):
                shared.append("".join(t.text or "" for t in si.iter("{%s}t" % _NS["m"])))
        sheets = sorted((n for n in names if re.match(r"xl/worksheets/sheet\d+\.xml$", n)), key=lambda n: int(re.findall(r"\d+", n)[-1]))
        out = []
        for sn in sheets:
            root = ET.fromstring(z.read(sn))
            rows = []
            for row in root.iter("{%s}row" % _NS["m"]):
                cells = {}
                for c in row.findall("m:c", _NS):
                    j = _col_index(c.get("r", "A1"))
                    t = c.get("t")
                    v = c.find("m:v", _NS)
                    if t == "inlineStr":
                        is_ = c.find("m:is", _NS)
                        cells[j] = Text("".join(x.text or "" for x in is_.iter("{%s}t" % _NS["m"])) if is_ is not None else "")
                    elif v is None or v.text is None:
                        cells[j] = NullWritable()
                    elif t == "s":
                        cells[j] = Text(shared[int(v.text)])
                    elif t == "b":
                        cells[j] = BooleanWritable(v.text == "1")
                    elif t in ("str", "e"):
                        cells[j] = Text(v.text)
                    else:
                        f = float(v.text)
                        cells[j] = IntWritable(int(f)) if f.is_integer() and "." not in v.text and "E" not in v.text.upper() else DoubleWritable(f)
                if cells:
                    rows.append([cells.get(j, NullWritable()) for j in range(max(cells) + 1)])
            out.append(rows)
        return out


def write_xlsx(path, sheets: List[List[List]], sheetNames=None):
    """Minimal writer: numbers as numeric cells, everything else as inline strings."""
    sheetNames = sheetNames or ["Sheet%d" % (i + 1) for i in range(len(sheets))]
    with zipfile.ZipFile(str(path), "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("[Content_Types].xml", '<?xml version="1.0" encoding="UTF-8"?><Types xmlns="http://schemas.openxmlformats.org/package/2006/content-types">'
                   '<Default Extension="rels" ContentType="application/vnd.openxmlformats-package.relationships+xml"/><Default Extension="xml" ContentType="application/xml"/>'
                   '<Override PartName="/xl/workbook.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.sheet.main+xml"/>'
                   + "".join('<Override PartName="/xl/worksheets/sheet%d.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.worksheet+xml"/>' % (i + 1)
                             for i in range(len(sheets))) + "</Types>")
        z.writestr("_rels/.rels", '<?xml version="1.0" encoding="UTF-8"?><Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
                   '<Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/officeDocument" Target="xl/workbook.xml"/></Relationships>')
        z.writestr("xl/workbook.xml", '<?xml version="1.0" encoding="UTF-8"?><workbook xmlns="%s" xmlns:r="http://schemas.openxmlformats.org/officeDocument/2006/relationships"><sheets>' % _NS["m"]
                   + "".join('<sheet name="%s" sheetId="%d" r:id="rId%d"/>' % (escape(n), i + 1, i + 1) for i, n in enumerate(sheetNames)) + "</sheets></workbook>")
        z.writestr("xl/_rels/workbook.xml.rels", '<?xml version="1.0" encoding="UTF-8"?><Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
                   + "".join('<Relationship Id="rId%d" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/worksheet" Target="worksheets/sheet%d.xml"/>' % (i + 1, i + 1)
                             for i in range(len(sheets))) + "</Relationships>")
        for si, rows in enumerate(sheets):
            parts = ['<?xml version="1.0" encoding="UTF-8"?><worksheet xmlns="%s"><sheetData>' % _NS["m"]]
            for ri, row in enumerate(rows):
                parts.append('<row r="%d">' % (ri + 1))
                for ci, v in enumerate(row):
                    ref = "%s%d" % (_col_name(ci), ri + 1)
                    if isinstance(v, Writable):
                        v = None if isinstance(v, NullWritable) else v.get()
                    if v is None:
                        continue
                    if isinstance(v, bool):
                        parts.append('<c r="%s" t="b"><v>%d</v></c>' % (ref, int(v)))
                    elif isinstance(v, (int, float)):
                        parts.append('<c r="%s"><v>%r</v></c>' % (ref, v))
                    else:
                        parts.append('<c r="%s" t="inlineStr"><is><t>%s</t></is></c>' % (ref, escape(str(v))))
                parts.append("</row>")
            parts.append("</sheetData></worksheet>")
            z.writestr("xl/worksheets/sheet%d.xml" % (si + 1), "".join(parts))
    return path


class ExcelRecordReader(RecordReader):
    def __init__(self, skipNumLines=0):
        super().__init__()
        self.skipNumLines = skipNumLines
        self._rows, self._pos = [], 0

    def reset(self):
        self._rows, self._pos = [], 0
        for loc in self.split.locations():
            for sheet in read_xlsx(loc):
                self._rows.extend(sheet[self.skipNumLines:])

    def hasNext(self):
        return self._pos < len(self._rows)

    def next(self):
        r = self._rows[self._pos]
        self._pos += 1
        return r


class ExcelRecordWriter:
    def __init__(self, sheetName="Sheet1"):
        self.sheetName, self._rows, self._path = sheetName, [], None

    def initialize(self, path):
        self._path = path
        return self

    def write(self, record):
        self._rows.append(list(record))

    def writeBatch(self, records):
        self._rows.extend(list(r) for r in records)

    def close(self):
        if self._path is not None:
            write_xlsx(self._path, [self._rows], [self.sheetName])
