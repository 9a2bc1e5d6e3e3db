cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
(timeout 1500 python -m pytest tests/test_gpu_dist_rccl.py tests/test_gpu_parity.py -m gpu -x -q -k "rccl or twin or full_size" > gpurun_out/r04a/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04a/tests.log)
tail -5 gpurun_out/r04a/tests.log
bash tools/profile_configs.sh r04base s10m_cube s1m r2 > gpurun_out/r04a/prof.log 2>&1
NO_PMC=1 bash tools/profile_configs.sh r04base config1 config5 >> gpurun_out/r04a/prof.log 2>&1
tail -20 gpurun_out/r04a/prof.log
