cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_prims.py tests/test_gpu_parity.py tests/test_gpu_dist_native.py -m gpu -x -q -k "not full_size and not config4 and not s40m" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log)
grep -E "passed|failed|rc=" $O/tests.log | tail -3
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04h/bench_line.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_host_to_host','scaling')}, d['roofline']['frac'], d['roofline']['kernel_ms'])
print(d['stages_ms'])
print({k:(v.get('value'),v.get('ms_per_step'),v.get('k3_frac'),v.get('ms_levelset')) for k,v in d['other_configs'].items()})
print('hbm_bound', d['splat_hbm_bound'].get('ms_per_step'), d['splat_hbm_bound'].get('roofline',{}).get('frac'), d['splat_hbm_bound'].get('roofline',{}).get('kernel_ms'))
print('e2e', d['e2e_host_u64'], d['pcie_inclusive'].get('value'), d['pcie_pipelined'].get('value'))
print('cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('simd_loop'))
PY
timeout 600 python bench.py --pseudo-ranks 8 > $O/pseudo8.json 2> $O/pseudo8.err; echo "pseudo rc=$?"; tail -c 1500 $O/pseudo8.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tank -o run -- python bench.py --main-only --steps 10 --warmup 2 > $O/stats_tank.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04h/stats_tank/run_kernel_stats.csv')))
for r in rows[:28]:
    print("%-80s calls %4s avg %9.1f us  %5.1f%%"%(r['Name'].replace('void ','')[:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
