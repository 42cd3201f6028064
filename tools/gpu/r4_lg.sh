cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_limit.py tests/test_gpu_parity.py tests/test_gpu_operators.py tests/test_gpu_cli.py tests/test_gpu_lifetime.py -x -q -m gpu > gpurun_out/r4_lg_tests.log 2>&1 || { tail -40 gpurun_out/r4_lg_tests.log; exit 1; }
tail -2 gpurun_out/r4_lg_tests.log
timeout -k 10 300 python tools/limit_probe.py 100000000 5 50000000 2>&1 | tee gpurun_out/r04a_limit_tool_output.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_limit_prof -o lim -- python3 $GRAFT_REPO_ROOT/tools/limit_probe.py 100000000 5 > /dev/null 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r4_limit_prof/lim_kernel_stats.csv | cut -c1-120
