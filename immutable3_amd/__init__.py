"""immutable3_amd -- MI355X-native scan / filter / project path of markosski/immutable3.

csrc/        hand-written HIP kernels (gfx950) + the C ABI of include/imm3.h  -> lib/libimm3.so
native.py    ctypes binding of that ABI (the only road to the GPU; no CPU fallback)
operators.py ScanOp / SelectOp / ProjectOp / Engine mirrors of the reference's operator interface
storage.py   the reference's on-disk format: SegmentManager (reader) and SegmentWriter / loader (writer)
schema.py, query.py   Column / Table / Row and the Query ADT
synth.py     seeded synthetic tables of BASELINE.json's configs
"""
from .query import (EQ, GT, LT, And, Avg, Count, Match, Max, Min, NoOp, NoSelect, NotMatch, Or, Project,  # noqa: F401
                    ProjectAgg, Query, Select, Sum)
from .schema import CodecType, Column, ColumnType, Row, Table, TableIO  # noqa: F401

__version__ = "0.1.0"
