cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b_sq
mkdir -p $O
timeout -k 10 600 python3 bench.py > gpurun_out/r04b_bench_plain.json 2>gpurun_out/r04b_bench_plain.err || { tail -5 gpurun_out/r04b_bench_plain.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $O/agg_pmc$i -o p -- python3 $R/tools/aggexp.py 0 > $O/agg_pmc$i.log 2>&1 || { tail -5 $O/agg_pmc$i.log; exit 1; }
done
cd $R
for d in $O/agg_pmc*; do python3 tools/pmc_summary.py $d > $d.summary.txt; done
cat $O/*.summary.txt | grep "k_group_agg"
python3 -c "
import json
d=json.loads(open('gpurun_out/r04b_bench_plain.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
x=d['extra']
print('c3', x['c3_range_age_id_project']['frac'], 'c5', x['c5']['roofline']['frac'], x['c5']['ms_per_step'])
"
