#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu14.log 2>&1
python tools/sweep.py --meshes torus1m,bunny,dragon --grids 256,512 --bricks 4 --stacks 0 --modes reference,parity --reps 7 > $OUT/sweep14.log 2>&1
exit 0
