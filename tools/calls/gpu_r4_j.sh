#!/bin/bash
# round 4, call J: one-sweep radix sort (option sort) -- correctness through the suite's word-for-word tests, then build / list build / refit loop A/B
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4j
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
for round in 1 2; do for o in 1 0; do
echo "# sort=$o" >> $OUT/build.jsonl
DXV_OPTIONS=sort=$o timeout 600 python tools/build_bench.py bunny torus1m soup10m >> $OUT/build.jsonl 2>&1
done; done
for o in 1 0; do
echo "# sort=$o" >> $OUT/refit.jsonl
DXV_OPTIONS=sort=$o timeout 600 python tools/refit_loop.py torus1m 512 60 >> $OUT/refit.jsonl 2>&1
done
cat $OUT/build.jsonl | cut -c1-400; cat $OUT/refit.jsonl | cut -c1-260
exit 0
