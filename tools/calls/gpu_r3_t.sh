#!/bin/bash
# bench.py with no warm-up at all and with the driver's defaults: the scene's launch structures are built before the clock either way
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3t
rm -rf $OUT; mkdir -p $OUT
python bench.py --gpus 1 --steps 20 --warmup 0 --no-cpu-baseline > $OUT/bench_w0.json 2> $OUT/bench_w0.err
python bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_w1.json 2> $OUT/bench_w1.err
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --gpus 2 --backend gloo --same-device --warmup 0 > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err
python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "bench or launcher" > $OUT/pytest.log 2>&1
exit 0
