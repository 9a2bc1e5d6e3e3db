#!/bin/bash
# tools/final_collect.sh TAG -- the round's final collection on the GPU box (gpurun): full GPU test suite, rocprofv3 stats + PMC of the headline workload in all
# three arithmetic modes (incl. the wait / LDS counters), the bench line (with roofline.traffic of THIS build), the secondary configurations with counters,
# pseudo-rank runs, host-to-host frames.  Everything lands under gpurun_out/; tools/make_profiles.py / make_cfg_profiles.py turn it into profiles/ at home.
cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
O=gpurun_out/${TAG}final; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log)
grep -E "passed|failed|rc=" $O/tests.log | tail -3
ROUND=$TAG MODES="0 1 2" EXTRA_PMC="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" bash tools/collect_profiles.sh > $O/collect.log 2>&1
python tools/make_profiles.py ${TAG}prof $TAG > $O/make_profiles.log 2>&1
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
bash tools/profile_configs.sh $TAG s10m_cube s1m r2 config1 config5 > $O/cfg.log 2>&1
for n in 8 4 2; do timeout 600 python bench.py --pseudo-ranks $n > $O/pseudo$n.json 2> $O/pseudo$n.err; done
timeout 600 python bench.py --pseudo-ranks 8 --slices > $O/pseudo8_slices.json 2> $O/pseudo8_slices.err   # rank r holds the r-th contiguous eighth (the distribution of the round 3-6 profiles; the default is brick-resident particles)
python tools/e2e_frames.py > $O/e2e_frames.log 2>&1
ls $O
