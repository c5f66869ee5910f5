#!/bin/bash
# round 3, call H: randomised soak on the current library (dispatch plans, repeated launches, class check in the draw) + bench line
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3h
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
python tools/gpu_soak.py 420 30301 > $OUT/soak_30301.jsonl 2>&1
python tools/gpu_soak.py 420 30302 > $OUT/soak_30302.jsonl 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python tools/rank_times.py torus1m 512 lists=2 noparity zb8 > $OUT/rank_times.jsonl 2>&1
exit 0
