cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O; rm -f $O/ab.jsonl
run() { SPLASHSURF_HIP_LIB=$2 timeout 300 python tools/ab_kernels.py --digest --tag $1 "${@:3}" >> $O/ab.jsonl 2>> $O/ab.err || echo "{\"tag\": \"$1\", \"failed\": true}" >> $O/ab.jsonl; }
run new "" --workload s10m_cube --steps 4
run new "" --workload s1m --steps 10
run new "" --workload s10m_tank --steps 8
run new "" --workload config5 --steps 20
python - <<'PY'
import json
for l in open('gpurun_out/r04p/ab.jsonl'):
    d=json.loads(l)
    if d.get('failed'): print(d); continue
    print("%-8s %-10s total %7.3f (min %7.3f) dens %6.3f (k %5.3f) ls %7.3f (acc %6.3f) cert %.3f dig %s"%(d['tag'],d['workload'],d['ms_total'],d['ms_total_min'],d['ms_density'],d['ms_density_kernel'],d['ms_levelset'],d['ms_levelset_accumulate'],d['certified_frac'],d.get('digest')))
PY
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not full_size and not config4" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log)
grep -E "passed|failed|rc=" $O/tests.log | tail -3
