"""The projection planner's cost model, in Python: the mirror of csrc/imm3_plan.h (same features, the coefficients read from
csrc/imm3_plan_coef.h), for tools/plan_fit.py -- which fits the coefficients to a sweep of tools/plan_sweep.py -- and for the tests
that hold the two implementations together.

Plans of an unlimited projection over one uniform segment:
  A  one launch (k_filter_project): the filter kernel writes the rows itself; gathered dense int32 columns ride along as streamed
     tile columns
  B  survivor records: filter + records -> k_scan -> k_emit
  C  the bitmap path: plain filter -> k_scan -> k_gather
Inputs: n rows; sigma = survivors per row; sloc = survivors per row where there are survivors (= sigma when they are spread evenly,
-> 1 for a range of a sorted key); full = the share of the survivors that sit in stretches where EVERY row survives (plan A copies
those; 1 for a range of a sorted key, 0 for 99 % of the rows spread evenly); pred = [(width, n_match)] of the predicate (tile) columns; proj = [(width, is_pred)] of the
SELECT list's distinct columns; rec_bytes = bytes of a survivor record (plan B)."""
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))


def coefficients(path=None):
    text = open(path or os.path.join(_HERE, "csrc", "imm3_plan_coef.h")).read()
    out = {}
    for plan in "ABC":
        m = re.search(r"kPlanCoef%s\[\] = \{([^}]*)\}" % plan, text)
        out[plan] = [float(x) for x in m.group(1).split(",")]
    return out


def lines_mb(n, w, sigma, sloc):
    """MB of 128-byte lines a gather of one column of width w touches"""
    if sigma <= 0.0:
        return 0.0
    sloc = min(1.0, max(sloc, sigma))
    return n * w / 1e6 * (sigma / sloc) * (1.0 - (1.0 - sloc) ** (128 // w))


def features(plan, n, sigma, sloc, full, pred, proj, rec_bytes):
    n6 = n / 1e6
    sloc = min(1.0, max(sloc, sigma))
    clustered = 1.0 if (sloc >= 0.9 and sloc > 1.5 * sigma) else 0.0   # a run of rows that all survive, in part of the segment
    rows = sigma * n6
    m = max([k for _, k in pred] + [0])
    stream_w = sum(w for w, _ in pred)
    out_b = 4 + sum(w for w, _ in proj)
    gathered = [w for w, is_pred in proj if not is_pred]
    if plan == "C":
        R = sum(lines_mb(n, w, sigma, sloc) for w, _ in proj)
        return [1.0, n6 * stream_w, n6 * m, n6 * m * m, R, rows, rows * out_b, rows * clustered, clustered, sloc if sigma > 0.0 else 0.0]   # (last: a work-group's time for one span of 16 tiles, whatever the segment's size)
    if plan == "B":
        R = sum(lines_mb(n, w, sigma, sloc) for w in gathered)
        n_i8 = sum(1 for w, _ in pred if w == 1)     # (the staging instance's LDS transposes cost most on 1-byte columns)
        return [1.0, n6 * stream_w, n6, n6 * m, rows * rec_bytes, R, rows, rows * out_b, clustered, n6 * n_i8, sloc if sigma > 0.0 else 0.0]
    if plan == "A":
        streamed = sum(gathered)                 # (plan A with gathered columns: every one of them is streamed)
        dense = max(0.0, rows - 0.08 * n6)
        return [1.0, n6, n6 * (stream_w + streamed), n6 * m, rows, dense * len(proj) * (1.0 - full), dense * full, dense * (1.0 if streamed else 0.0), dense * (out_b - 4) * (1.0 - full)]
    raise ValueError(plan)


def cost(plan, n, sigma, sloc, full, pred, proj, rec_bytes, coef=None):
    c = (coef or coefficients())[plan]
    f = features(plan, n, sigma, sloc, full, pred, proj, rec_bytes)
    return sum(a * b for a, b in zip(c, f))
