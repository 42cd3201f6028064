cd $GRAFT_REPO_ROOT
export IMM3_LIB_PATH=$GRAFT_REPO_ROOT/immutable3_amd/lib/libimm3_ablate.so
timeout -k 10 300 python tools/sp_explore.py C4 0 20 21 22 0 20 21 2>&1 | tee gpurun_out/r4_s2.txt
