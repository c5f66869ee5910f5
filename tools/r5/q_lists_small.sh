#!/bin/bash
OUT=gpurun_out/r5q; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lists or stages" > $OUT/lists_tests.log 2>&1; tail -3 $OUT/lists_tests.log
for m in bunny dragon dragon9 bunny16 torus1m; do timeout 120 python tools/build_once.py $m 3 2>&1 | tail -1 | cut -c1-400; done > $OUT/lists_small.log 2>&1
cat $OUT/lists_small.log
