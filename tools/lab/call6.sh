#!/bin/bash
# r05 call 6: K1 chain and the active-block detection launched behind the scans the host waits for (two host waits hidden): tests + A/B against HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05f; mkdir -p $O
V=$PWD/splashsurf_amd/variants
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
for rep in 1 2; do
  for wl in config1 config5 s1m s10m_tank; do
    st=20; [ $wl = s10m_tank ] && st=8
    for lib in "" base; do
      SPLASHSURF_HIP_LIB=${lib:+$V/libsplashsurf_hip_$lib.so} timeout 300 python tools/ab_kernels.py --workload $wl --simd 0 --steps $st --digest --tag "${lib:-new}" >> $O/ab.jsonl 2>> $O/ab.err
    done
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r05f/ab.jsonl'):
    d=json.loads(l); print(d['workload'], d['tag'], 'wall_ms', d.get('ms_wall'), 'ms_total', round(d['ms_total'],4), 'min', round(d.get('ms_total_min',0),4), 'dens', round(d['ms_density'],3), 'prep', round(d['ms_levelset_prepare'],3), d['digest'])
PY
