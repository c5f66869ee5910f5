#!/bin/bash
# which kernel of the list build is slow on the 100 k-triangle dragon on the 512 map
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4aa; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/tools/quick_times.py --meshes dragon --reps 2 > $OUT/log.txt 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-150
find $OUT/prof -name "*.csv" -size +2M -delete
