cd $GRAFT_REPO_ROOT
O=gpurun_out/r04r; mkdir -p $O; rm -f $O/ab.jsonl
run() { timeout 300 python tools/ab_kernels.py --digest --tag $1 "${@:2}" >> $O/ab.jsonl 2>> $O/ab.err || echo "{\"tag\": \"$1\", \"failed\": true}" >> $O/ab.jsonl; }
run new --workload s10m_cube --steps 4
run new --workload s1m --steps 10
run new --workload s10m_tank --cube-size 2.0 --steps 6
run new --workload s10m_tank --cube-size 1.0 --steps 6
run new --workload s10m_tank --steps 8
run new --workload config1 --steps 20
python - <<'PY'
import json
for l in open('gpurun_out/r04r/ab.jsonl'):
    d=json.loads(l)
    if d.get('failed'): print(d); continue
    print("%-5s %-10s total %7.3f (min %7.3f) dens %6.3f ls %7.3f (gather %6.3f acc %6.3f p2 %5.3f) mc %5.3f big %d dig %s"%(d['tag'],d['workload'],d['ms_total'],d['ms_total_min'],d['ms_density'],d['ms_levelset'],d['ms_levelset_gather'],d['ms_levelset_accumulate'],d['ms_levelset_accumulate_pass2'],d['ms_marching_cubes'],d['n_large'],d.get('digest')))
PY
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_simd.py tests/test_gpu_dist_native.py -m gpu -x -q -k "not full_size and not config4 and not s40m" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log)
grep -E "passed|failed|rc=" $O/tests.log | tail -3
