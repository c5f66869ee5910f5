#!/bin/bash
# round 4, call D: bench line with fresh_step, refit loop, per-rank shares
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4d
rm -rf $OUT; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
timeout 600 python tools/refit_loop.py torus1m 512 60 > $OUT/refit_loop.jsonl 2>&1
timeout 600 python tools/rank_times.py torus1m 512 "lists=2" noparity > $OUT/rank_times.jsonl 2>&1
timeout 600 python tools/rank_times.py torus1m 512 "lists=2,plan=2" noparity zb8 > $OUT/rank_times_fresh.jsonl 2>&1
cat $OUT/refit_loop.jsonl; grep '"world": 8' $OUT/rank_times.jsonl; grep '"world": 8' $OUT/rank_times_fresh.jsonl
exit 0
