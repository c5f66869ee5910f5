#!/bin/bash
# round 3, call A: dispatch plan -- correctness (whole GPU suite) and A/B timings
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
python tools/ab_option.py plan 0,2 --set lists=2 --meshes torus1m,bunny16,dragon9,bunny,dragon --grid 512 > $OUT/ab_plan.jsonl 2>&1
python tools/ab_option.py plan 0,2 --set lists=2 --meshes torus1m,bunny --grid 256 >> $OUT/ab_plan.jsonl 2>&1
python tools/ab_option.py plan 0,2 --set lists=2 --meshes soup10m --grid 512 --reps 3 --rounds 2 >> $OUT/ab_plan.jsonl 2>&1
python tools/rank_times.py torus1m 512 plan=0,lists=2 noparity zb8 > $OUT/rank_times.jsonl 2>&1
python tools/rank_times.py torus1m 512 plan=2,lists=2 noparity zb8 >> $OUT/rank_times.jsonl 2>&1
python tools/rank_times.py dragon9 1024 plan=0,lists=2 noparity zb8 >> $OUT/rank_times.jsonl 2>&1
python tools/rank_times.py dragon9 1024 plan=2,lists=2 noparity zb8 >> $OUT/rank_times.jsonl 2>&1
python tools/ab_option.py listres 256,512 --set lists=2,plan=2 --meshes torus1m,bunny16 --grid 512 --rounds 2 > $OUT/ab_listres.jsonl 2>&1
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
exit 0
