#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu4.log 2>&1
: > $OUT/sweep4.log
for rb in 3 6 9 12 15 18; do
echo "# region=$rb" >> $OUT/sweep4.log
python tools/sweep.py --meshes torus1m,bunny,dragon --grids 512 --bricks 2,4,6 --stacks 0 --reps 5 --opts morton=1,region=$rb >> $OUT/sweep4.log 2>&1
done
exit 0
