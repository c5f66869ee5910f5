#!/bin/bash
# A/B on one box: regions of 256 bricks dealt to the eight queues (committed) against single bricks dealt round-robin
# (queue x = the bricks whose Morton number is x mod 8) -> gpurun_out/r4l/
OUT=gpurun_out/r4l; mkdir -p $OUT
D=$PWD/dxrvoxelizer_amd
for rep in 1 2; do
  for lib in libdxv.so libdxv_deal.so; do
    DXV_LIBRARY=$D/$lib timeout 300 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny --reps 15 > $OUT/q_${lib}_$rep.jsonl 2>&1
    DXV_LIBRARY=$D/$lib timeout 300 python tools/rank_times.py torus1m 512 lists=2 noparity zb8 > $OUT/rank_${lib}_$rep.jsonl 2>&1
  done
done
DXV_LIBRARY=$D/libdxv.so timeout 300 python tools/rank_times.py bunny16 512 lists=2 noparity zb8 > $OUT/rank_bunny16_libdxv.so.jsonl 2>&1
DXV_LIBRARY=$D/libdxv_deal.so timeout 300 python tools/rank_times.py bunny16 512 lists=2 noparity zb8 > $OUT/rank_bunny16_libdxv_deal.so.jsonl 2>&1
for lib in libdxv_qtimes.so libdxv_deal_qt.so; do
  DXV_LIBRARY=$D/$lib timeout 300 python tools/queue_times.py torus1m 512 > $OUT/qt_$lib.jsonl 2>&1
  QT_WORLD=8 DXV_LIBRARY=$D/$lib timeout 300 python tools/queue_times.py torus1m 512 >> $OUT/qt_$lib.jsonl 2>&1
done
grep -h lists_ms $OUT/q_*_1.jsonl | cut -c1-120; grep -h '"world": 8' $OUT/rank_*; cut -c1-300 $OUT/qt_*.jsonl
