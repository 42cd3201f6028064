// imm3_plan_coef.h -- coefficients of the projection planner's cost model (imm3_plan.h), microseconds per feature.
// WRITTEN BY tools/plan_fit.py from a sweep of tools/plan_sweep.py on one MI355X (profiles/README.md says which): do not edit by hand.
#pragma once
namespace imm3 {
static const double kPlanCoefA[] = {22.895, 0.505048, 0.0472583, 0.120623, 1.46741, 0, 1.0264, 0, 0.625851};
static const double kPlanCoefB[] = {15.3837, 0.117582, 0.119709, 0.0640089, 0.110955, 0.145229, 0, 0.376986, 7.54919, 0.160703, 26.2146};
static const double kPlanCoefC[] = {12.7313, 0.170183, 0, 0.00215593, 0.155046, 1.87586, 0.0881826, 0.485262, 6.23943, 10.9129};
} // namespace imm3
