cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_agg.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r4_agg_tests.log 2>&1 || { tail -40 gpurun_out/r4_agg_tests.log; exit 1; }
tail -2 gpurun_out/r4_agg_tests.log
IMM3_LIB_PATH=$GRAFT_REPO_ROOT/immutable3_amd/lib/libimm3_ablate.so timeout -k 10 300 python tools/gpu/agg_n.py
