#!/bin/bash
# long soak of the round's final library (two seeds) + the driver's round-end checks
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3soak
rm -rf $OUT; mkdir -p $OUT
python tools/gpu_soak.py 1200 30321 > $OUT/soak_30321.jsonl 2>&1
python tools/gpu_soak.py 1200 30322 > $OUT/soak_30322.jsonl 2>&1
(time python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
exit 0
