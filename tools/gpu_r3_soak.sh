#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3soak
rm -rf $OUT; mkdir -p $OUT
python tools/gpu_soak.py 900 30311 > $OUT/soak_30311.jsonl 2>&1
python tools/gpu_soak.py 900 30312 > $OUT/soak_30312.jsonl 2>&1
exit 0
