#!/bin/bash
# the driver's round-end checks: GPU suite + smoke + default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/suite
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
exit 0
