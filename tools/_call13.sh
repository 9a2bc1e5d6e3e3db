cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O; rm -f $O/ab.jsonl
(timeout 600 python -m pytest tests/test_gpu_prims.py tests/test_gpu_parity.py -m gpu -x -q -k "not full_size and not config4" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log)
grep -E "passed|failed|rc=" $O/tests.log | tail -3
run() { SPLASHSURF_HIP_LIB=$2 timeout 300 python tools/ab_kernels.py --digest --tag $1 "${@:3}" >> $O/ab.jsonl 2>> $O/ab.err || echo "{\"tag\": \"$1\", \"failed\": true}" >> $O/ab.jsonl; }
for rep in 1 2; do
run new "" --workload s10m_tank --steps 8
done
run new "" --workload s40m_tank --steps 4
python - <<'PY'
import json
for l in open('gpurun_out/r04m/ab.jsonl'):
    d=json.loads(l)
    if d.get('failed'): print(d); continue
    print("%-5s %-10s total %7.3f (min %7.3f) dec %6.3f dens %6.3f (k %5.3f) lsprep %6.3f ls %7.3f (gather %6.3f acc %6.3f p2 %5.3f) mc %5.3f st %5.3f act %d cert %.3f big %d dig %s"%(d['tag'],d['workload'],d['ms_total'],d['ms_total_min'],d['ms_decomposition'],d['ms_density'],d['ms_density_kernel'],d['ms_levelset_prepare'],d['ms_levelset'],d['ms_levelset_gather'],d['ms_levelset_accumulate'],d['ms_levelset_accumulate_pass2'],d['ms_marching_cubes'],d['ms_stitching'],d['n_active'],d['certified_frac'],d['n_large'],d.get('digest')))
PY
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prims -o run -- python tools/prims_timing.py > $O/prims_timing.log 2>&1
python3 - <<'PY'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/r04m/prims/run_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'k_rs_pass' in n or 'k_rs_hist' in n or 'k_chained' in n:
        d[(n.split('(')[0][-40:], r.get('Grid_Size_X') or r.get('Grid_Size',''))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:(kv[0][0], -max(kv[1]))):
    v.sort(); print("%-45s grid %-10s n=%3d median %8.1f us min %8.1f"%(k[0],k[1],len(v),v[len(v)//2],v[0]))
PY
