#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu9.log 2>&1
: > $OUT/sweep9.log
for q in 0 1; do
echo "# queue=$q" >> $OUT/sweep9.log
python tools/sweep.py --meshes torus1m,bunny,dragon --grids 512 --bricks 4 --stacks 0 --modes reference,parity --reps 5 --opts queue=$q >> $OUT/sweep9.log 2>&1
done
exit 0
