cd $GRAFT_REPO_ROOT
export IMM3_LIB_PATH=$GRAFT_REPO_ROOT/immutable3_amd/lib/libimm3_ablate.so
python tools/sp_explore.py C3 57 65 57:0:207 57:0:208 57:0:204 57:0:205 57:0:212 0:0:207 0:0:205 > gpurun_out/r4_abl2.log 2>&1
cat gpurun_out/r4_abl2.log
python tools/sp_timeline.py 0 > gpurun_out/r4_tl0.log 2>&1; cat gpurun_out/r4_tl0.log
python tools/sp_timeline.py 51 > gpurun_out/r4_tl51.log 2>&1; head -8 gpurun_out/r4_tl51.log
