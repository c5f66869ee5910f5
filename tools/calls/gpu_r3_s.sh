#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3s
rm -rf $OUT; mkdir -p $OUT
python tools/refit_loop.py torus1m 512 80 > $OUT/refit_loop.jsonl 2>&1
python tools/refit_loop.py bunny16 512 40 >> $OUT/refit_loop.jsonl 2>&1
exit 0
