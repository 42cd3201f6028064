"""Column / Table / Row -- mirrors of core/src/main/scala/immutabledb/{Column,Table,Record}.scala
and the `_table.meta` JSON they read and write."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Dict, List


class ColumnType:                  # Column.scala:13-16
    INT, TINYINT, STRING = "INT", "TINYINT", "STRING"


class CodecType:                   # codec/Codec.scala:21-24
    PFOR_INT, DENSE_INT, DENSE_TINYINT, DENSE_STRING = "PFOR_INT", "DENSE_INT", "DENSE_TINYINT", "DENSE_STRING"
    ORDER = ["PFOR_INT", "DENSE_INT", "DENSE_TINYINT", "DENSE_STRING"]
    # EXTENSION, not in the reference's enumeration: blocks written by SnappyCodec.encode (codec/SnappyCodec.scala:15-43).
    # The reference cannot name or read such a column (no CodecType, decode = ???); ids follow include/imm3.h.
    SNAPPY_INT, SNAPPY_TINYINT, SNAPPY_STRING = "SNAPPY_INT", "SNAPPY_TINYINT", "SNAPPY_STRING"
    SNAPPY = {"SNAPPY_INT": 16, "SNAPPY_TINYINT": 17, "SNAPPY_STRING": 18}
    INT_CODECS = ("DENSE_INT", "PFOR_INT", "SNAPPY_INT")
    TINYINT_CODECS = ("DENSE_TINYINT", "SNAPPY_TINYINT")
    STRING_CODECS = ("DENSE_STRING", "SNAPPY_STRING")

    @staticmethod
    def id_of(name: str) -> int:
        return CodecType.SNAPPY[name] if name in CodecType.SNAPPY else CodecType.ORDER.index(name)


@dataclass(frozen=True)
class Column:                      # Column.scala:18
    name: str
    columnType: str
    codec: str
    dtypeAttrs: tuple = ()         # ((key, value), ...) -- Map[String, String]

    @staticmethod
    def make(name: str, codec: str, dtypeAttrs: Dict[str, str] | None = None) -> "Column":  # Column.scala:45-55
        ctype = {
            CodecType.DENSE_INT: ColumnType.INT,
            CodecType.PFOR_INT: ColumnType.INT,
            CodecType.DENSE_TINYINT: ColumnType.TINYINT,
            CodecType.DENSE_STRING: ColumnType.STRING,
            CodecType.SNAPPY_INT: ColumnType.INT,
            CodecType.SNAPPY_TINYINT: ColumnType.TINYINT,
            CodecType.SNAPPY_STRING: ColumnType.STRING,
        }.get(codec)
        if ctype is None:
            raise Exception("")
        return Column(name, ctype, codec, tuple((dtypeAttrs or {}).items()))

    @property
    def attrs(self) -> Dict[str, str]:
        return dict(self.dtypeAttrs)

    @property
    def width(self) -> int:
        """dtype.size of the column's codec (DataType.scala:34,54; Column.scala:60 for strings)."""
        if self.codec in CodecType.INT_CODECS:
            return 4
        if self.codec in CodecType.TINYINT_CODECS:
            return 1
        if self.codec in CodecType.STRING_CODECS:
            return int(self.attrs["size"])
        raise Exception("")

    def to_json(self):             # Column.toJsonValue, Column.scala:21-29
        return {"name": self.name, "columnType": self.columnType, "codec": self.codec, "dtypeAttrs": dict(self.dtypeAttrs)}

    @staticmethod
    def from_json(j) -> "Column":  # Column.fromJsonValue, Column.scala:31-38
        return Column(j["name"], j["columnType"], j["codec"], tuple((k, str(v)) for k, v in j["dtypeAttrs"].items()))


@dataclass(frozen=True)
class Table:                       # Table.scala:9-14
    name: str
    columns: tuple
    blockSize: int

    def __init__(self, name, columns, blockSize):
        object.__setattr__(self, "name", name)
        object.__setattr__(self, "columns", tuple(columns))
        object.__setattr__(self, "blockSize", int(blockSize))

    def getColumn(self, colName: str) -> Column:
        for c in self.columns:
            if c.name == colName:
                return c
        raise Exception(f"Column {colName} does not exist in table {self.name}")


class TableIO:                     # Table.scala:26-66
    fileName = "_table.meta"

    @staticmethod
    def load(dataDir: str, tableName: str) -> Table:
        with open(os.path.join(dataDir, tableName, TableIO.fileName)) as f:
            j = json.load(f)
        return Table(j["name"], [Column.from_json(c) for c in j["columns"]], int(j["blockSize"]))

    @staticmethod
    def store(dataDir: str, table: Table):
        path = os.path.join(dataDir, table.name)
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, TableIO.fileName), "w") as f:
            json.dump({"name": table.name, "columns": [c.to_json() for c in table.columns], "blockSize": table.blockSize},
                      f, separators=(",", ":"))

    @staticmethod
    def clear(dataDir: str, table: Table):
        path = os.path.join(dataDir, table.name)
        if os.path.isdir(path):
            for fn in os.listdir(path):
                fp = os.path.join(path, fn)
                if os.path.isfile(fp):
                    os.remove(fp)


class Row(tuple):                  # Record.scala:7-14
    """Row(xs: Any*): Int / Byte / String values in SELECT-list order."""

    def __new__(cls, *xs):
        return super().__new__(cls, xs)

    @staticmethod
    def fromSeq(xs) -> "Row":
        return Row(*xs)

    def getByte(self, idx):
        return self[idx]

    def getInt(self, idx):
        return self[idx]

    def getString(self, idx):
        return self[idx]

    def __repr__(self):
        return "Row(" + ",".join(str(x) for x in self) + ")"
