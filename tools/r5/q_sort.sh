#!/bin/bash
# new radix sort: stage / list identity tests, then build timings
OUT=gpurun_out/r5q; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stages or lists or refit or build" > $OUT/sort_tests.log 2>&1; tail -3 $OUT/sort_tests.log
for m in torus1m soup10m; do timeout 120 python tools/build_once.py $m 4 2>&1 | tail -3 | cut -c1-400; done > $OUT/sort_build.log 2>&1
cat $OUT/sort_build.log
