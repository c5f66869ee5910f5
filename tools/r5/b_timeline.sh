#!/bin/bash
# round 5, call b: time lines of short launches (diagnostic library with per-workgroup stamps)
OUT=gpurun_out/r5b; mkdir -p $OUT
L=$PWD/dxrvoxelizer_amd/libdxv_qtimes.so
for m in torus1m bunny16; do
  DXV_LIBRARY=$L timeout 300 python tools/wg_times.py $m 512 8 4 planregion=8 >> $OUT/wg_times.jsonl 2>> $OUT/err.log
  DXV_LIBRARY=$L timeout 300 python tools/wg_times.py $m 512 8 4 planregion=6 >> $OUT/wg_times.jsonl 2>> $OUT/err.log
done
DXV_LIBRARY=$L timeout 300 python tools/wg_times.py torus1m 512 1 4 >> $OUT/wg_times.jsonl 2>> $OUT/err.log
DXV_LIBRARY=$L timeout 300 python tools/wg_times.py torus1m 256 1 4 >> $OUT/wg_times.jsonl 2>> $OUT/err.log
DXV_LIBRARY=$L QT_WORLD=8 timeout 300 python tools/queue_times.py torus1m 512 plan=2 >> $OUT/queue_times.jsonl 2>> $OUT/err.log
tail -5 $OUT/err.log
cat $OUT/wg_times.jsonl | cut -c1-2500
cat $OUT/queue_times.jsonl
