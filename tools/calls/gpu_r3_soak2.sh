#!/bin/bash
# two more 25-minute soak seeds on the round's final library
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3soak2
rm -rf $OUT; mkdir -p $OUT
python tools/gpu_soak.py 1500 30331 > $OUT/soak_30331.jsonl 2>&1
python tools/gpu_soak.py 1500 30332 > $OUT/soak_30332.jsonl 2>&1
exit 0
