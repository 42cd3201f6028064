cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4_full_tests.log 2>&1 || { tail -60 gpurun_out/r4_full_tests.log; exit 1; }
tail -2 gpurun_out/r4_full_tests.log
timeout -k 10 600 python3 bench.py > gpurun_out/r04b_bench_plain.json 2>gpurun_out/r04b_bench_plain.err || { tail -5 gpurun_out/r04b_bench_plain.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r04b_bench_plain.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
x=d['extra']
for k in ('c3_range_age_id_project','c4_match_state_project','agg_group_by_state_all_rows','agg_group_by_state_range_age','limit10_met_in_the_first_rows','limit10_met_in_the_second_half'):
    print(k, {kk:x[k].get(kk) for kk in ('ms_per_query','frac','one_shot_ms','kernel_ms_sum','kernel_ms')})
print('c5', x['c5']['ms_per_step'], x['c5']['roofline']['frac'], x['c5'].get('abandoned_runs'))
"
