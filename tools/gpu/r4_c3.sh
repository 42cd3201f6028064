set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_single_pass.py tests/test_gpu_fault_injection.py -x -q -m gpu > gpurun_out/r4_c3_tests.log 2>&1 || { tail -30 gpurun_out/r4_c3_tests.log; exit 1; }
tail -3 gpurun_out/r4_c3_tests.log
IMM3_CASES="C3,age11%->id+age,id50%,age50%,age30%+id,age99%,age1%+id,C4" python tools/proj_bench.py > gpurun_out/r4_c3_proj.log 2>&1
cat gpurun_out/r4_c3_proj.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_sq1 -- python3 $GRAFT_REPO_ROOT/tools/sp_explore.py C3 0 > $GRAFT_REPO_ROOT/gpurun_out/r4_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_sq2 -- python3 $GRAFT_REPO_ROOT/tools/sp_explore.py C3 0 > $GRAFT_REPO_ROOT/gpurun_out/r4_sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/r4_sq1 k_filter_project
python tools/pmc_summary.py gpurun_out/r4_sq2 k_filter_project
