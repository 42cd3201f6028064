"""Query ADT -- mirror of core/src/main/scala/immutabledb/Query.scala:3-46 (same names, same fields)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional


class SelectCondition:
    pass


@dataclass(frozen=True)
class Match(SelectCondition):      # Query.scala:4
    values: tuple

    def __init__(self, values):
        object.__setattr__(self, "values", tuple(values))


@dataclass(frozen=True)
class NotMatch(SelectCondition):   # Query.scala:5 (SelectOp rejects it, Select.scala:22)
    values: tuple

    def __init__(self, values):
        object.__setattr__(self, "values", tuple(values))


@dataclass(frozen=True)
class EQ(SelectCondition):         # Query.scala:6
    eq: float


@dataclass(frozen=True)
class GT(SelectCondition):         # Query.scala:7
    gt: float


@dataclass(frozen=True)
class LT(SelectCondition):         # Query.scala:8
    lt: float


@dataclass(frozen=True)
class _NoOp(SelectCondition):      # Query.scala:9
    pass


NoOp = _NoOp()


class SelectADT:
    pass


@dataclass(frozen=True)
class And(SelectADT):              # Query.scala:12
    op1: SelectADT
    op2: SelectADT


@dataclass(frozen=True)
class Or(SelectADT):               # Query.scala:13 (executed exactly like And: Engine.scala:240 ignores the tag)
    op1: SelectADT
    op2: SelectADT


@dataclass(frozen=True)
class Select(SelectADT):           # Query.scala:14
    col: str
    cond: SelectCondition


@dataclass(frozen=True)
class _NoSelect(SelectADT):        # Query.scala:15
    pass


NoSelect = _NoSelect()


class ProjectADT:
    pass


@dataclass(frozen=True)
class Project(ProjectADT):         # Query.scala:29
    cols: tuple
    limit: int = 0

    def __init__(self, cols, limit: int = 0):
        object.__setattr__(self, "cols", tuple(cols))
        object.__setattr__(self, "limit", int(limit))


@dataclass(frozen=True)
class Aggregate:                   # Query.scala:17-20
    col: str
    alias: Optional[str] = None


class Sum(Aggregate):              # Query.scala:21 (parsed, then rejected: Engine.scala:152)
    pass


class Avg(Aggregate):              # Query.scala:22
    pass


class Min(Aggregate):              # Query.scala:23
    pass


class Max(Aggregate):              # Query.scala:24
    pass


class Count(Aggregate):            # Query.scala:25
    pass


@dataclass(frozen=True)
class ProjectAgg(ProjectADT):      # Query.scala:30
    aggs: tuple
    groupBy: tuple = ()

    def __init__(self, aggs, groupBy=()):
        object.__setattr__(self, "aggs", tuple(aggs))
        object.__setattr__(self, "groupBy", tuple(groupBy))


@dataclass(frozen=True)
class Query:                       # Query.scala:42-46
    table: str
    select: SelectADT
    project: ProjectADT
