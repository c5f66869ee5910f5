#!/bin/bash
# one more 24-minute soak seed on the round's final library
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3soak3
rm -rf $OUT; mkdir -p $OUT
python tools/gpu_soak.py 1440 30351 > $OUT/soak_30351.jsonl 2>&1
exit 0
