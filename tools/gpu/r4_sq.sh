cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04a_sq
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_plan_quality.py -x -q -m gpu > gpurun_out/r4_pq.log 2>&1 || { tail -30 gpurun_out/r4_pq.log; exit 1; }
tail -2 gpurun_out/r4_pq.log
timeout -k 10 600 python3 bench.py > gpurun_out/r04a_bench_plain.json 2>gpurun_out/r04a_bench_plain.err || { tail -5 gpurun_out/r04a_bench_plain.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $O/c3_pmc$i -o p -- python3 $R/tools/sp_explore.py C3 0 > $O/c3_pmc$i.log 2>&1 || { tail -5 $O/c3_pmc$i.log; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $O/id50_pmc$i -o p -- python3 $R/tools/sp_explore.py id50% 0 > $O/id50_pmc$i.log 2>&1 || { tail -5 $O/id50_pmc$i.log; exit 1; }
done
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $O/agg_pmc$i -o p -- python3 $R/tools/aggexp.py 0 > $O/agg_pmc$i.log 2>&1 || { tail -5 $O/agg_pmc$i.log; exit 1; }
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/id50_trace -o t -- python3 $R/tools/sp_explore.py id50% 0 > $O/id50_trace.log 2>&1
cd $R
for d in $O/c3_pmc* $O/id50_pmc* $O/agg_pmc*; do python3 tools/pmc_summary.py $d > $d.summary.txt; done
cat $O/*.summary.txt | grep -v rocclr | grep "k_filter_project\|k_group_agg"
