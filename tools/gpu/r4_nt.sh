cd $GRAFT_REPO_ROOT
export IMM3_LIB_PATH=$GRAFT_REPO_ROOT/immutable3_amd/lib/libimm3_ablate.so
timeout -k 10 300 python tools/sp_explore.py C3 0 82 0 82 0 82 2>&1 | tee gpurun_out/r4_nt.txt
