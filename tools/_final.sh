# final collection of a round: full GPU test suite, bench line, rocprofv3 stats + PMC of the headline workload (all three arithmetic modes),
# the secondary configurations, pseudo-rank runs.  Everything lands under gpurun_out/; tools/make_profiles.py / make_cfg_profiles.py turn it into profiles/.
cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
O=gpurun_out/${TAG}final; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log)
grep -E "passed|failed|rc=" $O/tests.log | tail -3
ROUND=$TAG MODES="1 0 2" bash tools/collect_profiles.sh > $O/collect.log 2>&1
# (profiles/splat_traffic.json of THIS build first, so that the bench line below carries roofline.traffic; the same command is run again
# on the merged gpurun_out/ at home to produce the tracked files)
python tools/make_profiles.py ${TAG}prof $TAG > $O/make_profiles.log 2>&1
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
bash tools/profile_configs.sh $TAG s10m_cube s1m r2 > $O/cfg.log 2>&1
NO_PMC=1 bash tools/profile_configs.sh $TAG config1 config5 >> $O/cfg.log 2>&1
for n in 8 4 2; do timeout 600 python bench.py --pseudo-ranks $n > $O/pseudo$n.json 2> $O/pseudo$n.err; done
python tools/e2e_frames.py > $O/e2e_frames.log 2>&1
ls $O
