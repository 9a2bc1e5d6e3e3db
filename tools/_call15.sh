cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O; rm -f $O/ab.jsonl
V=$PWD/splashsurf_amd/variants
run() { SPLASHSURF_HIP_LIB=$2 timeout 300 python tools/ab_kernels.py --digest --tag $1 "${@:3}" >> $O/ab.jsonl 2>> $O/ab.err || echo "{\"tag\": \"$1\", \"failed\": true}" >> $O/ab.jsonl; }
run new "" --workload s10m_tank --steps 8
for v in rn58 rn62 pool196 pool228 dq12 dq24 dc8 dc2; do
run $v $V/libsplashsurf_hip_$v.so --workload s10m_tank --steps 8
done
run new "" --workload s10m_tank --steps 8
for v in dq12 dq24 dc8 dc2; do
run $v $V/libsplashsurf_hip_$v.so --workload s10m_cube --steps 4
done
run new "" --workload s10m_cube --steps 4
python - <<'PY'
import json
for l in open('gpurun_out/r04o/ab.jsonl'):
    d=json.loads(l)
    if d.get('failed'): print(d); continue
    print("%-8s %-10s total %7.3f (min %7.3f) dens %6.3f (k %5.3f) ls %7.3f (acc %6.3f) cert %.3f dig %s"%(d['tag'],d['workload'],d['ms_total'],d['ms_total_min'],d['ms_density'],d['ms_density_kernel'],d['ms_levelset'],d['ms_levelset_accumulate'],d['certified_frac'],d.get('digest')))
PY
