#!/usr/bin/env python3
"""Fits the projection planner's cost model (csrc/imm3_plan.h) to a sweep of tools/plan_sweep.py and says how good the model's
choices would be on that sweep: per plan (A one launch, B survivor records, C bitmap path) a linear model over features that
follow the kernels' traffic and per-row work, least squares on relative error.
usage: plan_fit.py sweep.json [coefficient header to write]"""
import json, re, sys
import numpy as np
from scipy.optimize import nnls
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import plan_model

d = json.load(open(sys.argv[1]))
cells = d["cells"]

W = {"id": 4, "state": 2, "age": 1}


def parse(shape):
    """-> pred cols [(name, n_match)], projected cols [names], clustered"""
    lhs, rhs = shape.split(" -> ")
    proj = [x.strip() for x in rhs.split(",")]
    preds = []
    for part in lhs.split(" and "):
        part = part.strip()
        if part.startswith("age"):
            preds.append(("age", 0))
        elif part.startswith("id"):
            preds.append(("id", 0))
        elif part.startswith("state in"):
            preds.append(("state", int(part.split()[2])))
    return preds, proj


def inputs(c):
    """a sweep cell -> the model's inputs"""
    n = c["rows"]
    sigma = c["selected"] / n
    clustered = c["kind"] == "clustered"
    preds, proj = parse(c["shape"])
    pred_names = [p for p, _ in preds]
    pred = [(W[p], k) for p, k in preds]
    pj = [(W[p], p in pred_names) for p in dict.fromkeys(proj)]
    rec_bytes = 8 if "id" in pred_names else 4          # rec_layout(): {pos | narrow values} in one dword, an int32 value in a second
    return dict(n=n, sigma=sigma, sloc=1.0 if clustered else sigma, full=1.0 if clustered else 0.0, pred=pred, proj=pj, rec_bytes=rec_bytes)


def features(c, plan):
    return plan_model.features(plan, **inputs(c))


def letter(p):
    return "A" if p.startswith("one") else ("B" if p == "records" else "C")


def measured(c):
    """per plan letter: the best steady time any variant reached with it.  A through variant 8 with a gathered narrow column is the
    in-kernel gather nobody would plan: only the planner's own kind of A counts (v0 / v9), or v8 when all projected columns stream"""
    preds, proj = parse(c["shape"])
    pred_names = [p for p, _ in preds]
    narrow_gather = any(p not in pred_names and p != "id" for p in proj)
    per = {}
    for k, x in c["variants"].items():
        l = letter(x["plan"])
        if l == "A" and narrow_gather:
            continue
        per[l] = min(per.get(l, 1e9), x["steady_us"])
    return per


coef = {}
for plan in "ABC":
    X, y = [], []
    for c in cells:
        per = measured(c)
        if plan in per:
            X.append(features(c, plan))
            y.append(per[plan])
    X, y = np.array(X, float), np.array(y, float)
    w = 1.0 / y
    sol, _ = nnls(X * w[:, None], y * w)                    # non-negative: every term is a cost
    coef[plan] = sol
    pred = X @ sol
    rel = pred / y
    print(f"plan {plan}: {len(y)} points, model/measured: median {np.median(rel):.2f}, 5 % .. 95 %: {np.percentile(rel, 5):.2f} .. {np.percentile(rel, 95):.2f}, worst {rel.min():.2f} / {rel.max():.2f}")
    print("   coefficients:", ", ".join(f"{v:.4g}" for v in sol))

ratios, ratios_now = [], []
bad = []
for c in cells:
    per = measured(c)
    pred = {p: float(np.dot(features(c, p), coef[p])) for p in per}
    choice = min(pred, key=pred.get)
    best = min(per.values())
    ratios.append(per[choice] / best)
    now = letter(c["variants"]["0"]["plan"])
    ratios_now.append(c["variants"]["0"]["steady_us"] / best)
    if per[choice] / best > 1.10:
        bad.append(f"{c['rows'] // 1000000:3d}M {c['shape']:32s} model picks {choice} ({per[choice]:.0f}, predicted " + "/".join(f"{p}{pred[p]:.0f}" for p in sorted(pred)) + ") measured " + "/".join(f"{p}{per[p]:.0f}" for p in sorted(per)))
r = np.array(ratios)
rn = np.array(ratios_now)
print(f"planner today: within 10 % of the best plan in {100 * (rn <= 1.10).mean():.1f} % of {rn.size} cells (worst {rn.max():.2f})")
print(f"model's choice: within 10 % in {100 * (r <= 1.10).mean():.1f} %, within 5 % in {100 * (r <= 1.05).mean():.1f} % (worst {r.max():.2f})")
print("\n".join(bad))
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        f.write("// imm3_plan_coef.h -- coefficients of the projection planner's cost model (imm3_plan.h), microseconds per feature.\n"
                "// WRITTEN BY tools/plan_fit.py from a sweep of tools/plan_sweep.py on one MI355X (profiles/README.md says which): do not edit by hand.\n"
                "#pragma once\nnamespace imm3 {\n")
        for p in "ABC":
            f.write(f"static const double kPlanCoef{p}[] = {{" + ", ".join(f"{v:.6g}" for v in coef[p]) + "};\n")
        f.write("} // namespace imm3\n")
