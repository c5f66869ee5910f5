#!/bin/bash
# round 3, call M: radix sort with 9/10-bit digits (LBVH: 3 passes, lists: 4 instead of 5)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3m
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
python tools/build_once.py soup10m 4 > $OUT/build_soup10m.jsonl 2>&1
python tools/build_once.py torus1m 4 > $OUT/build_torus1m.jsonl 2>&1
python tools/build_bench.py bunny torus1m soup10m > $OUT/build_bench.jsonl 2>&1
python tools/refit_loop.py torus1m 512 40 > $OUT/refit_loop.jsonl 2>&1
exit 0
