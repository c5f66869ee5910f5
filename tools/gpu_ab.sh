#!/bin/bash
# A/B two builds of libdxv.so in one GPU session, interleaved rounds.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
: > $OUT/ab.log
for round in 1 2 3; do
for v in old new; do
echo "# $v round $round" >> $OUT/ab.log
DXV_LIBRARY=$GRAFT_REPO_ROOT/dxrvoxelizer_amd/libdxv_$v.so python tools/sweep.py --meshes torus1m,bunny --grids 512 --bricks 4 --stacks 0 --reps 7 >> $OUT/ab.log 2>&1
done; done
exit 0
