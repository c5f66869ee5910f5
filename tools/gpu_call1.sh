#!/bin/bash
# First GPU session: parity tests, smoke, variant sweep, bench line, rocprof kernel stats.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12 > $OUT/rocminfo.txt 2>&1
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
python tools/sweep.py --meshes torus1m,bunny --grids 256,512 --bricks 0,1,2,3 --stacks 0,64 --reps 3 > $OUT/sweep1.log 2>&1
python tools/sweep.py --meshes torus1m --grids 512 --bricks 1 --stacks 0 --modes parity --reps 3 >> $OUT/sweep1.log 2>&1
python bench.py --steps 10 --warmup 2 > $OUT/bench1.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof1.log 2>&1
ls -R $OUT/prof1 | head -30 >> $OUT/prof1.log
exit 0
