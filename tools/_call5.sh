cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_prims.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log)
tail -15 $O/tests.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prims -o run -- python tools/prims_timing.py > $O/prims_timing.log 2>&1
cat $O/prims_timing.log | grep "^{"
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04e/prims/run_kernel_stats.csv')))
for r in rows[:14]:
    print("%-90s calls %4s avg %9.1f us"%(r['Name'].replace('void ','')[:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
