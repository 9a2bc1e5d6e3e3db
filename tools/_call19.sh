cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; mkdir -p $O; rm -f $O/ab.jsonl
V=$PWD/splashsurf_amd/variants
run() { SPLASHSURF_HIP_LIB=$2 timeout 300 python tools/ab_kernels.py --digest --tag $1 "${@:3}" >> $O/ab.jsonl 2>> $O/ab.err || echo "{\"tag\": \"$1\", \"failed\": true}" >> $O/ab.jsonl; }
for rep in 1 2; do
run new "" --workload s10m_tank --steps 8
run lscan $V/libsplashsurf_hip_lscan.so --workload s10m_tank --steps 8
done
run new "" --workload s10m_tank --steps 6 --simd 0
run lscan $V/libsplashsurf_hip_lscan.so --workload s10m_tank --steps 6 --simd 0
run new "" --workload config5 --steps 20
run lscan $V/libsplashsurf_hip_lscan.so --workload config5 --steps 20
for v in new cert96; do
L=""; [ $v = cert96 ] && L=$V/libsplashsurf_hip_cert96.so
run $v "$L" --workload s10m_cube --steps 4
run $v "$L" --workload s10m_tank --cube-size 2.0 --steps 6
run $v "$L" --workload s1m --steps 10
done
python - <<'PY'
import json
for l in open('gpurun_out/r04s/ab.jsonl'):
    d=json.loads(l)
    if d.get('failed'): print(d); continue
    print("%-6s %-10s simd%d total %7.3f (min %7.3f) dens %6.3f ls %7.3f (gather %6.3f acc %6.3f p2 %5.3f) mc %5.3f cert %.4f dig %s"%(d['tag'],d['workload'],d['simd'],d['ms_total'],d['ms_total_min'],d['ms_density'],d['ms_levelset'],d['ms_levelset_gather'],d['ms_levelset_accumulate'],d['ms_levelset_accumulate_pass2'],d['ms_marching_cubes'],d['certified_frac'],d.get('digest')))
PY
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not full_size and not config4" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log)
grep -E "passed|failed|rc=" $O/tests.log | tail -3
