cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O; rm -f $O/ab.jsonl
run() { timeout 300 python tools/ab_kernels.py --digest --tag $1 "${@:2}" >> $O/ab.jsonl 2>> $O/ab.err || echo "{\"tag\": \"$1\", \"failed\": true}" >> $O/ab.jsonl; }
for rep in 1 2; do
SPLASH_K1_OVERLAP=0 run serial --workload s10m_tank --steps 8
run overlap --workload s10m_tank --steps 8
done
for w in s10m_cube s1m config1 config5; do
SPLASH_K1_OVERLAP=0 run serial --workload $w --steps 10
run overlap --workload $w --steps 10
done
python - <<'PY'
import json
for l in open('gpurun_out/r04q/ab.jsonl'):
    d=json.loads(l)
    if d.get('failed'): print(d); continue
    print("%-8s %-10s total %7.3f (min %7.3f) dec %6.3f dens %6.3f (k %5.3f) ls %7.3f mc %5.3f dig %s"%(d['tag'],d['workload'],d['ms_total'],d['ms_total_min'],d['ms_decomposition'],d['ms_density'],d['ms_density_kernel'],d['ms_levelset'],d['ms_marching_cubes'],d.get('digest')))
PY
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist_native.py tests/test_gpu_simd.py -m gpu -x -q -k "not full_size and not config4 and not s40m" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log)
grep -E "passed|failed|rc=" $O/tests.log | tail -3
