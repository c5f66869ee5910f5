#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
: > $OUT/sweep16.log
for rb in 4 6 8; do
echo "# region=$rb" >> $OUT/sweep16.log
python tools/sweep.py --meshes torus1m,bunny,dragon --grids 512 --bricks 4,5,6,7 --stacks 0 --reps 5 --opts region=$rb >> $OUT/sweep16.log 2>&1
done
exit 0
