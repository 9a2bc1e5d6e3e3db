cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "staged or inplace or error or golden" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log)
grep -E "passed|failed|rc=|Error" $O/tests.log | tail -5
python tools/e2e_frames.py 2>&1 | grep "^{"
