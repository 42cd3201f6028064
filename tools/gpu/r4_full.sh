cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_full_tests.log 2>&1 || { tail -40 gpurun_out/r4_full_tests.log; exit 1; }
tail -3 gpurun_out/r4_full_tests.log
bash tools/gpu/r4_profile.sh r04b
