cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_single_pass.py -x -q -m gpu > gpurun_out/r4_d2_tests.log 2>&1 || { tail -30 gpurun_out/r4_d2_tests.log; exit 1; }
tail -3 gpurun_out/r4_d2_tests.log
for L in libimm3_ablate.so libimm3_d1.so; do
export IMM3_LIB_PATH=$GRAFT_REPO_ROOT/immutable3_amd/lib/$L
echo == $L
python tools/sp_explore.py C3 0 51 57 185 0:0:204 0:0:208 2>&1 | tee -a gpurun_out/r4_d2.log
done
export IMM3_LIB_PATH=$GRAFT_REPO_ROOT/immutable3_amd/lib/libimm3_ablate.so
python tools/sp_timeline.py 0 2>&1 | head -9 | tee -a gpurun_out/r4_d2.log
unset IMM3_LIB_PATH
IMM3_CASES="C3,age11%->id+age,id50%,age50%,age30%+id,age99%,age1%+id" python tools/proj_bench.py 2>&1 | tee -a gpurun_out/r4_d2.log
