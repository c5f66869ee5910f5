#!/bin/bash
# two more seeds of the randomised soak on the round's final library
OUT=gpurun_out/r4ae; mkdir -p $OUT
timeout 1300 python tools/gpu_soak.py 1200 40431 > $OUT/soak_40431.jsonl 2>&1
timeout 1300 python tools/gpu_soak.py 1200 40432 > $OUT/soak_40432.jsonl 2>&1
tail -n 1 $OUT/soak_*.jsonl | cut -c1-300
