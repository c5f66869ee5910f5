#!/bin/bash
OUT=gpurun_out/r5q; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lists or refit" > $OUT/lists_tests.log 2>&1; tail -2 $OUT/lists_tests.log
for m in bunny dragon dragon9 bunny16 torus1m soup10m; do echo "$m: $(timeout 120 python tools/build_once.py $m 4 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['list_ms'], d['list_entries'])")"; done > $OUT/emit_exp.log 2>&1
cat $OUT/emit_exp.log
