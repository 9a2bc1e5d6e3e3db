#!/bin/bash
# r05 call 1: GPU tests (incl. the full-size wheel digests), default bench line, pseudo-rank projection with / without partition feedback
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > $O/pytest.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --pseudo-ranks 8 --steps 10 --warmup 2 > $O/pseudo8_fb.json 2> $O/pseudo8_fb.err
timeout 300 python bench.py --pseudo-ranks 8 --steps 10 --warmup 2 --no-balance-feedback > $O/pseudo8_nofb.json 2> $O/pseudo8_nofb.err
tail -5 $O/pytest.log
