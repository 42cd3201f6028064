// imm3_plan.h -- the projection planner's cost model (host side).
//
// An unlimited projection over one uniform segment can run as
//   A  one launch (k_filter_project): the filter kernel writes the rows itself; gathered dense int32 columns ride along as
//      streamed tile columns,
//   B  survivor records: filter + records -> k_scan -> k_emit,
//   C  the bitmap path: plain filter -> k_scan -> k_gather,
// and which is fastest depends on the rows, on how many survive and where, and on the widths of what is streamed and gathered
// (profiles/r04_plan_sweep.txt: at 100 M rows A wins C3's shape below 30 % survivors and loses it above 60 %; at 4 M rows C wins
// nearly everything).  Rounds 2-3 decided with thresholds measured on a handful of shapes at 100 M rows; this is the measured
// replacement: per plan a linear model -- microseconds = sum of coefficient x feature -- whose features follow the kernels'
// traffic (bytes streamed, 128-byte lines a gather touches, record bytes) and per-row work, and whose coefficients
// (imm3_plan_coef.h) tools/plan_fit.py fits to a sweep of tools/plan_sweep.py: rows x survivors x spread / clustered x SELECT-list
// shape, every plan forced through the tuning hook.  The planner takes the cheapest eligible plan; a plan in use is kept unless
// another is predicted 3 % cheaper (an estimate and the count that follows it do not flip the plan).
// immutable3_amd/plan_model.py is the same model in Python (tests/test_host.py holds the two together).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "imm3_plan_coef.h"

namespace imm3 {

constexpr int kPlanMaxCols = 8;

struct PlanShape {
    int64_t n_rows = 0;
    int32_t n_pred = 0;                    // predicate (tile) columns
    int32_t pred_width[kPlanMaxCols] = {}; // bytes per value
    int32_t pred_match[kPlanMaxCols] = {}; // values of an IN-list (string columns), else 0
    int32_t n_proj = 0;                    // distinct SELECT-list columns
    int32_t proj_width[kPlanMaxCols] = {};
    bool proj_is_pred[kPlanMaxCols] = {};
    int32_t rec_bytes = 4;                 // bytes of a survivor record (plan B)
};

// Where the survivors are: sigma = survivors per row; sloc = survivors per row where there are survivors; full = the share of
// the survivors that sit in stretches where every row survives (plan A copies those).
struct PlanDensity {
    double sigma = 0.1, sloc = 0.1, full = 0.0;
};

// MB of 128-byte lines a gather of one column of width w touches
inline double plan_lines_mb(double n, int w, double sigma, double sloc) {
    if (!(sigma > 0.0) || w <= 0) return 0.0;
    sloc = std::min(1.0, std::max(sloc, sigma));
    return n * w / 1e6 * (sigma / sloc) * (1.0 - std::pow(1.0 - sloc, (double)(128 / w)));
}

// plan: 'A', 'B' or 'C'.  Predicted microseconds of one run's kernels.
inline double plan_cost(char plan, const PlanShape &s, PlanDensity d) {
    const double n = (double)s.n_rows, n6 = n / 1e6;
    d.sigma = std::min(1.0, std::max(0.0, d.sigma));
    d.sloc = std::min(1.0, std::max(d.sloc, d.sigma));
    const double clustered = (d.sloc >= 0.9 && d.sloc > 1.5 * d.sigma) ? 1.0 : 0.0; // a run of rows that all survive, in part of the segment
    const double rows = d.sigma * n6;
    double m = 0.0, stream_w = 0.0, out_b = 4.0;
    for (int i = 0; i < s.n_pred; ++i) {
        m = std::max(m, (double)s.pred_match[i]);
        stream_w += s.pred_width[i];
    }
    for (int i = 0; i < s.n_proj; ++i) out_b += s.proj_width[i];
    double f[11] = {0};
    const double *c = nullptr;
    int nf = 0;
    if (plan == 'C') {
        double R = 0.0;
        for (int i = 0; i < s.n_proj; ++i) R += plan_lines_mb(n, s.proj_width[i], d.sigma, d.sloc);
        // (last: a work-group's time for one span of 16 tiles -- the gather's floor whatever the segment's size)
        const double v[10] = {1.0, n6 * stream_w, n6 * m, n6 * m * m, R, rows, rows * out_b, rows * clustered, clustered, d.sigma > 0.0 ? d.sloc : 0.0};
        std::copy(v, v + 10, f);
        c = kPlanCoefC;
        nf = 10;
    } else if (plan == 'B') {
        double R = 0.0;
        for (int i = 0; i < s.n_proj; ++i)
            if (!s.proj_is_pred[i]) R += plan_lines_mb(n, s.proj_width[i], d.sigma, d.sloc);
        double n_i8 = 0.0; // (the staging instance's LDS transposes cost most on 1-byte columns)
        for (int i = 0; i < s.n_pred; ++i) n_i8 += s.pred_width[i] == 1 ? 1.0 : 0.0;
        const double v[11] = {1.0, n6 * stream_w, n6, n6 * m, rows * s.rec_bytes, R, rows, rows * out_b, clustered, n6 * n_i8, d.sigma > 0.0 ? d.sloc : 0.0};
        std::copy(v, v + 11, f);
        c = kPlanCoefB;
        nf = 11;
    } else {
        double streamed = 0.0;
        for (int i = 0; i < s.n_proj; ++i)
            if (!s.proj_is_pred[i]) streamed += s.proj_width[i]; // (plan A with gathered columns: every one of them is streamed)
        const double dense = std::max(0.0, rows - 0.08 * n6);
        const double v[9] = {1.0, n6, n6 * (stream_w + streamed), n6 * m, rows, dense * s.n_proj * (1.0 - d.full), dense * d.full, dense * (streamed > 0.0 ? 1.0 : 0.0),
                             dense * (out_b - 4.0) * (1.0 - d.full)};
        std::copy(v, v + 9, f);
        c = kPlanCoefA;
        nf = 9;
    }
    double t = 0.0;
    for (int i = 0; i < nf; ++i) t += c[i] * f[i];
    return t;
}

constexpr double kPlanKeepMargin = 0.97; // a plan in use is left only for one predicted at least 3 % cheaper

} // namespace imm3
