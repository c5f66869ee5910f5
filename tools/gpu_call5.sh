#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu5.log 2>&1
: > $OUT/sweep5.log
for q in 0 1; do for s0 in 16 24 32; do
echo "# queue=$q stack0=$s0" >> $OUT/sweep5.log
python tools/sweep.py --meshes torus1m,bunny,dragon --grids 512 --bricks 4 --stacks 0 --reps 5 --opts queue=$q,stack0=$s0 >> $OUT/sweep5.log 2>&1
done; done
exit 0
