#!/bin/bash
OUT=gpurun_out/r4p; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "lists or list or refit or deferred or queue or fuzz" > $OUT/pytest_subset.log 2>&1; tail -15 $OUT/pytest_subset.log
for how in 1 2; do
DXV_OPTIONS="listbuild=$how" timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny --reps 5 > $OUT/q_$how.jsonl 2>&1; cut -c1-170 $OUT/q_$how.jsonl
done
timeout 600 python tools/refit_loop.py torus1m 512 60 > $OUT/refit_loop.jsonl 2>&1; cut -c1-260 $OUT/refit_loop.jsonl | tail -3
