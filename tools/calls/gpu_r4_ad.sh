#!/bin/bash
# config 5 on other maps now that texels outside the outline get no entry
OUT=gpurun_out/r4ad; mkdir -p $OUT
for res in 128 512; do timeout 600 python tools/quick_times.py --meshes soup10m --reps 3 --set listres=$res > $OUT/soup_res$res.jsonl 2>&1; cut -c1-260 $OUT/soup_res$res.jsonl; done
