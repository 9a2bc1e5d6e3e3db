cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O; rm -f $O/ab.jsonl
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_simd.py tests/test_gpu_fuzz.py tests/test_gpu_dist_native.py -m gpu -x -q -k "not full_size and not config4 and not s40m" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log)
grep -E "passed|failed|rc=" $O/tests.log | tail -3
V=$PWD/splashsurf_amd/variants
run() { SPLASHSURF_HIP_LIB=$2 timeout 300 python tools/ab_kernels.py --digest --tag $1 "${@:3}" >> $O/ab.jsonl 2>> $O/ab.err || echo "{\"tag\": \"$1\", \"failed\": true}" >> $O/ab.jsonl; }
for rep in 1 2; do
run pos4 $V/libsplashsurf_hip_pos4.so --workload s10m_tank --steps 8
run new "" --workload s10m_tank --steps 8
done
for w in s10m_cube s1m; do
run pos4 $V/libsplashsurf_hip_pos4.so --workload $w --steps 6
run new "" --workload $w --steps 6
done
python - <<'PY'
import json
for l in open('gpurun_out/r04l/ab.jsonl'):
    d=json.loads(l)
    if d.get('failed'): print(d); continue
    print("%-5s %-10s total %7.3f (min %7.3f) dec %6.3f dens %6.3f (k %5.3f) lsprep %6.3f ls %7.3f (gather %6.3f acc %6.3f p2 %5.3f) mc %5.3f st %5.3f act %d cert %.3f big %d dig %s"%(d['tag'],d['workload'],d['ms_total'],d['ms_total_min'],d['ms_decomposition'],d['ms_density'],d['ms_density_kernel'],d['ms_levelset_prepare'],d['ms_levelset'],d['ms_levelset_gather'],d['ms_levelset_accumulate'],d['ms_levelset_accumulate_pass2'],d['ms_marching_cubes'],d['ms_stitching'],d['n_active'],d['certified_frac'],d['n_large'],d.get('digest')))
PY
