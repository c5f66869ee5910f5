#!/bin/bash
# round 3, call R: digit-total scan folded into the scatter kernel: build and list build times, sort / build tests
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3r
rm -rf $OUT; mkdir -p $OUT
(time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "build or sort or keys or lists_equal or refit or hierarchy") > $OUT/pytest_gpu.log 2>&1
python tools/build_once.py torus1m 6 > $OUT/build_torus1m.jsonl 2>&1
python tools/build_once.py soup10m 4 > $OUT/build_soup10m.jsonl 2>&1
python tools/refit_loop.py torus1m 512 60 > $OUT/refit_loop.jsonl 2>&1
exit 0
