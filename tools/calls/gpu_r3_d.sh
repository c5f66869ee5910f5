#!/bin/bash
# round 3, call D: kernel-level times of the LBVH build and the list build (10 M and 1 M triangles); GPU suite on the current library
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3d
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
python tools/build_once.py soup10m 3 > $OUT/build_soup10m.jsonl 2>&1
python tools/build_once.py torus1m 4 > $OUT/build_torus1m.jsonl 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_soup -- python3 $GRAFT_REPO_ROOT/tools/build_once.py soup10m 3 > $OUT/prof_soup.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_torus -- python3 $GRAFT_REPO_ROOT/tools/build_once.py torus1m 4 > $OUT/prof_torus.log 2>&1
cd $GRAFT_REPO_ROOT
for d in prof_soup prof_torus; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv; done
python tools/rank_times.py torus1m 512 plan=2,lists=2 noparity zb8 > $OUT/rank_times.jsonl 2>&1
exit 0
