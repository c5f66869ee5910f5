#!/bin/bash
# 256^3 on the 1 M-triangle mesh: where does the queue launch's time go (wave timeline), against the brick box and round 3's library
OUT=gpurun_out/r4t; mkdir -p $OUT
D=$PWD/dxrvoxelizer_amd
DXV_LIBRARY=$D/libdxv_qtimes.so timeout 300 python tools/queue_times.py torus1m 256 > $OUT/qt_256.jsonl 2>&1
DXV_LIBRARY=$D/libdxv_qtimes.so timeout 300 python tools/queue_times.py bunny 256 >> $OUT/qt_256.jsonl 2>&1
timeout 300 python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 --reps 15 --fresh > $OUT/q256_new.jsonl 2>&1
for w in 2048 3584 5120; do timeout 300 python tools/quick_times.py --meshes torus1m --grid 256 --reps 15 --set queuewaves=$w > $OUT/q256_waves$w.jsonl 2>&1; done
(cd .ab_old && timeout 300 python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 --reps 15 > ../$OUT/q256_round3.jsonl 2>&1)
cat $OUT/qt_256.jsonl | cut -c1-1500; grep -h lists_ms $OUT/q256_*.jsonl | cut -c1-200
