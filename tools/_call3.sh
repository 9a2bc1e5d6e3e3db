cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O; rm -f $O/ab.jsonl
V=$PWD/splashsurf_amd/variants
run() { SPLASHSURF_HIP_LIB=$2 timeout 300 python tools/ab_kernels.py --digest --tag $1 "${@:3}" >> $O/ab.jsonl 2>> $O/ab.err || echo "{\"tag\": \"$1\", \"failed\": true}" >> $O/ab.jsonl; }
for rep in 1 2; do
run base $V/libsplashsurf_hip_base.so --workload s10m_tank --steps 8
run k1 $V/libsplashsurf_hip_k1.so --workload s10m_tank --steps 8
run k2 "" --workload s10m_tank --steps 8
done
for w in s10m_cube s1m config1 config5; do
run base $V/libsplashsurf_hip_base.so --workload $w --steps 6
run k1 $V/libsplashsurf_hip_k1.so --workload $w --steps 6
run k2 "" --workload $w --steps 6
done
run base $V/libsplashsurf_hip_base.so --workload s10m_tank --cube-size 2.0 --steps 6
run k1 $V/libsplashsurf_hip_k1.so --workload s10m_tank --cube-size 2.0 --steps 6
run k2 "" --workload s10m_tank --cube-size 2.0 --steps 6
SPLASHSURF_HIP_LIB=$V/libsplashsurf_hip_k1prof.so python tools/phase_prof.py > $O/phase_k1.txt 2>&1
SPLASHSURF_HIP_LIB=$V/libsplashsurf_hip_k2prof.so python tools/phase_prof.py > $O/phase_k2.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04c/ab.jsonl'):
    d=json.loads(l)
    if d.get('failed'): print(d); continue
    print("%-5s %-10s simd%s total %7.3f dec %6.3f dens %6.3f lsprep %6.3f ls %7.3f (gather %6.3f acc %6.3f p2 %5.3f) mc %5.3f st %5.3f act %d cert %.3f big %d nv %d dig %s"%(d['tag'],d['workload'],d['simd'],d['ms_total'],d['ms_decomposition'],d['ms_density'],d['ms_levelset_prepare'],d['ms_levelset'],d['ms_levelset_gather'],d['ms_levelset_accumulate'],d['ms_levelset_accumulate_pass2'],d['ms_marching_cubes'],d['ms_stitching'],d['n_active'],d['certified_frac'],d['n_large'],d['n_vertices'],d.get('digest')))
PY
cat $O/phase_k1.txt $O/phase_k2.txt
