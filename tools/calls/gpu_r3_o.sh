#!/bin/bash
# round 3, call O: refit that stops at the pyramid + vertex upload beside the launch in flight: tests and the refit loop
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3o
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "refit or upload or update or frames or blob") > $OUT/pytest_gpu.log 2>&1
python tools/refit_loop.py torus1m 512 60 > $OUT/refit_loop.jsonl 2>&1
python tools/refit_loop.py bunny16 512 40 >> $OUT/refit_loop.jsonl 2>&1
exit 0
