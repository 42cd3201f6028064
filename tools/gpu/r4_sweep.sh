cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4_full_tests.log 2>&1 || { tail -60 gpurun_out/r4_full_tests.log; exit 1; }
tail -2 gpurun_out/r4_full_tests.log
timeout -k 10 900 python tools/plan_sweep.py gpurun_out/r4_plan_sweep2.json > gpurun_out/r4_plan_sweep2.txt 2>&1 || { tail -20 gpurun_out/r4_plan_sweep2.txt; exit 1; }
tail -34 gpurun_out/r4_plan_sweep2.txt
