#!/bin/bash
# round 3, call E: LBVH build with the half-float copy from k_refit_ranges' registers and the four-box copy from the half-float one
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3e
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
python tools/build_once.py soup10m 3 > $OUT/build_soup10m.jsonl 2>&1
python tools/build_once.py torus1m 4 > $OUT/build_torus1m.jsonl 2>&1
python tools/build_bench.py bunny torus1m soup10m > $OUT/build_bench.jsonl 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_soup -- python3 $GRAFT_REPO_ROOT/tools/build_once.py soup10m 3 > $OUT/prof_soup.log 2>&1
cd $GRAFT_REPO_ROOT
for d in prof_soup; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv; done
python tools/refit_loop.py torus1m 512 40 > $OUT/refit_loop.jsonl 2>&1
exit 0
