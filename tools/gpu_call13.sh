#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu13.log 2>&1
: > $OUT/sweep13.log
for r in 0 1; do
echo "# rows=$r" >> $OUT/sweep13.log
python tools/sweep.py --meshes torus1m,bunny,dragon --grids 256,512 --bricks 4 --stacks 0 --modes parity --reps 5 --opts rows=$r >> $OUT/sweep13.log 2>&1
done
python tools/sweep.py --meshes soup10m --grids 512 --bricks 4 --stacks 0 --modes reference,parity --reps 3 >> $OUT/sweep13.log 2>&1
exit 0
