#!/bin/bash
# long randomised soak of the round's final library (options drawn include the work queue's dispatch), two seeds
OUT=gpurun_out/r4x; mkdir -p $OUT
timeout 1300 python tools/gpu_soak.py 1200 40411 > $OUT/soak_40411.jsonl 2>&1
timeout 1300 python tools/gpu_soak.py 1200 40412 > $OUT/soak_40412.jsonl 2>&1
tail -n 2 $OUT/soak_*.jsonl | cut -c1-300
