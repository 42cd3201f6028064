# the round's profiling session (profiles/README.md): kernel trace + stats, FETCH_SIZE and WRITE_SIZE passes of the bench command,
# the un-profiled tools.  usage: bash tools/gpu/r5_profile.sh <tag>
TAG=${1:-r05a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G=gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $G/${TAG}_trace -o $TAG -- python3 bench.py --no-cpu-baseline --no-limit > $G/${TAG}_trace.log 2>$G/${TAG}_trace.err || { tail -5 $G/${TAG}_trace.err; exit 1; }
echo trace done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $G/${TAG}_fetch -o $TAG -- python3 bench.py --no-cpu-baseline --no-limit --steps 20 --warmup 3 > $G/${TAG}_fetch.log 2>$G/${TAG}_fetch.err || { tail -5 $G/${TAG}_fetch.err; exit 1; }
echo fetch done
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $G/${TAG}_write -o $TAG -- python3 bench.py --no-cpu-baseline --no-limit --steps 20 --warmup 3 > $G/${TAG}_write.log 2>$G/${TAG}_write.err || { tail -5 $G/${TAG}_write.err; exit 1; }
echo write done
timeout -k 10 600 python3 bench.py > $G/${TAG}_bench_plain.json 2>$G/${TAG}_bench_plain.err || { tail -5 $G/${TAG}_bench_plain.err; exit 1; }
echo bench done
timeout -k 10 300 python3 tools/kinds_bench.py > $G/${TAG}_kinds_tool_output.txt 2>&1
IMM3_VARIANTS=0,6,3 timeout -k 10 400 python3 tools/proj_bench.py > $G/${TAG}_projection_tool_output.txt 2>&1
timeout -k 10 300 python3 tools/first_run.py > $G/${TAG}_first_run_tool_output.txt 2>&1
timeout -k 10 300 python3 tools/limit_probe.py > $G/${TAG}_limit_tool_output.txt 2>&1
find $G/${TAG}_trace $G/${TAG}_fetch $G/${TAG}_write -name "*.csv" -size +30M -exec ls -la {} \;
du -sh $G/${TAG}_*
python3 -c "
import json
d=json.loads(open('$G/${TAG}_bench_plain.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])
x=d['extra']
for k in ('c3_range_age_id_project','c4_match_state_project','agg_group_by_state_all_rows','agg_group_by_state_range_age','limit10_met_in_the_first_rows','limit10_met_in_the_second_half'):
    print(k, {kk:x[k].get(kk) for kk in ('ms_per_query','frac','one_shot_ms','kernel_ms_sum')})
print('c5', x['c5']['ms_per_step'], x['c5']['roofline']['frac'], x['c5'].get('abandoned_runs'))
"
