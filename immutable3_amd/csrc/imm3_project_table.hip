// imm3_project_table.hip -- the TABLE instances of k_filter_project (imm3_project.hip): ScanOp -> SelectOp* -> ProjectOp over
// every segment a GPU owns in ONE launch (imm3_table's tile table; the reference merges its per-segment pipelines into one result,
// engine/src/main/scala/immutabledb/engine/Engine.scala:176-196, and its on-disk shape is ~98 segments per 100 M rows,
// README.md:10).  A translation unit of its own so that the two sets of sixteen kernels compile side by side.
#define IMM3_PROJECT_TABLE_TU 1
#include "imm3_project.hip"
