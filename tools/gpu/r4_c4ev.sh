cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_c4
mkdir -p $O
timeout -k 10 200 python tools/sp_explore.py C4 0 3 > $O/timing.txt 2>&1 && timeout -k 10 200 python tools/sp_explore.py C4-id-age 0 3 >> $O/timing.txt 2>&1 || { cat $O/timing.txt; exit 1; }
cat $O/timing.txt
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $O/c4_pmc$i -o p -- python3 $R/tools/sp_explore.py C4 0 > $O/c4_pmc$i.log 2>&1 || { tail -5 $O/c4_pmc$i.log; exit 1; }
  echo "c4 pass $i done"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_trace -o t -- python3 $R/tools/sp_explore.py C4 0 > $O/c4_trace.log 2>&1 || exit 1
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $O/id50_pmc$i -o p -- python3 $R/tools/sp_explore.py id50% 0 > $O/id50_pmc$i.log 2>&1 || { tail -5 $O/id50_pmc$i.log; exit 1; }
  echo "id50 pass $i done"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/id50_trace -o t -- python3 $R/tools/sp_explore.py id50% 0 > $O/id50_trace.log 2>&1 || exit 1
cd $R
for d in $O/c4_pmc* $O/id50_pmc*; do python3 tools/pmc_summary.py $d > $d.summary.txt; done
du -sh $O; find $O -name "*counter_collection.csv" -size +20M -delete; du -sh $O
