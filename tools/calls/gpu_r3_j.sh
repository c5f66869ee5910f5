#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3j
rm -rf $OUT; mkdir -p $OUT
python tools/step_loop.py torus1m 512 8 4 200 > $OUT/step_loop.jsonl 2>&1
python tools/step_loop.py torus1m 512 1 8 100 >> $OUT/step_loop.jsonl 2>&1
exit 0
