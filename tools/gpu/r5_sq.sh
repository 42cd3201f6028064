# round 5: SQ counters of the fused aggregation launch (tools/aggexp.py: fused = tuning 0, two launches = tuning 17) and of the in-lane
# narrow filter instances (tools/kinds_bench.py), plus a kernel trace of the kinds tool.  usage: bash tools/gpu/r5_sq.sh <tag>
TAG=${1:-r05a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}_sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $O/agg_pmc$i -o p -- python3 $R/tools/aggexp.py 0 17 > $O/agg_pmc$i.log 2>&1 || { tail -5 $O/agg_pmc$i.log; exit 1; }
  IMM3_KINDS="I8,S2,I8+I8,I8+S2,S2 in(4)" timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $O/kinds_pmc$i -o p -- python3 $R/tools/kinds_bench.py > $O/kinds_pmc$i.log 2>&1 || { tail -5 $O/kinds_pmc$i.log; exit 1; }
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kinds_trace -o t -- python3 $R/tools/kinds_bench.py > $O/kinds_trace.log 2>&1 || { tail -5 $O/kinds_trace.log; exit 1; }
cd $R
for d in $O/agg_pmc* $O/kinds_pmc*; do [ -d $d ] && python3 tools/pmc_summary.py $d > $d.summary.txt; done
cat $O/agg_pmc*.summary.txt | grep k_group_agg
cat $O/kinds_pmc*.summary.txt | grep k_filter_tile
find $O -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kinds_kernel_stats.csv \;
cat $O/kinds_trace.log
