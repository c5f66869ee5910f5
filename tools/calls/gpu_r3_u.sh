#!/bin/bash
# round 3, call U: stop codes of the short lists through an LDS stage (k_dm_stops): full suite, list build times, refit loop, a soak
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3u
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(time python -m pytest tests -m gpu -q -p no:cacheprovider -x) > $OUT/pytest_gpu.log 2>&1
for m in torus1m bunny16 dragon9 bunny dragon soup10m; do python tools/list_build_once.py $m 6 >> $OUT/list_build.jsonl 2>&1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python3 tools/list_build_once.py torus1m 10 > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/list_build_kernel_stats.csv \;
rm -rf $OUT/prof
python tools/refit_loop.py torus1m 512 60 > $OUT/refit_loop.jsonl 2>&1
python tools/gpu_soak.py 600 30341 > $OUT/soak_30341.jsonl 2>&1
exit 0
