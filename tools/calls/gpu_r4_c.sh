#!/bin/bash
# round 4, call C: when do the persistent waves start and end (diagnostic library)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c
rm -rf $OUT; mkdir -p $OUT
export DXV_LIBRARY=$GRAFT_REPO_ROOT/dxrvoxelizer_amd/libdxv_qtimes.so
for m in torus1m dragon9 bunny16; do
for opt in "" "queuesteal=0" "queuebalance=1"; do
echo "# $m $opt" >> $OUT/times.log
timeout 300 python tools/queue_times.py $m 512 "$opt" >> $OUT/times.log 2>&1
done; done
cat $OUT/times.log
exit 0
