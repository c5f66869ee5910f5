#!/bin/bash
# randomised soak of the round's FINAL library (texel culling, per-texel radial ranges, map policy, dispatch of kept queues)
OUT=gpurun_out/r4ab; mkdir -p $OUT
timeout 1300 python tools/gpu_soak.py 1200 40421 > $OUT/soak_40421.jsonl 2>&1
tail -n 2 $OUT/soak_40421.jsonl | cut -c1-300
