#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
: > $OUT/sweep12.log
for s0 in 16 20 24; do
echo "# stack0=$s0" >> $OUT/sweep12.log
python tools/sweep.py --meshes torus1m,bunny,dragon --grids 512 --bricks 4 --stacks 0 --modes reference,parity --reps 5 --opts stack0=$s0 >> $OUT/sweep12.log 2>&1
done
exit 0
