#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
(time python tools/configs.py) > $OUT/configs7.log 2>&1
python bench.py --steps 20 --warmup 3 > $OUT/bench7.log 2>&1
exit 0
