#!/bin/bash
# A/B on one box: the committed persistent kernel (72 registers, seven waves) against the software-pipelined loop
# (next brick's first step before this brick's stores; 80 registers, six waves) -> gpurun_out/r4k/
OUT=gpurun_out/r4k; mkdir -p $OUT
for rep in 1 2 3; do
  for lib in libdxv.so libdxv_dxv_pipe.so; do
    DXV_LIBRARY=$PWD/dxrvoxelizer_amd/$lib timeout 300 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny --reps 15 --fresh > $OUT/q_${lib}_$rep.jsonl 2>&1
  done
done
for lib in libdxv.so libdxv_dxv_pipe.so; do
  DXV_LIBRARY=$PWD/dxrvoxelizer_amd/$lib timeout 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline > $OUT/bench_$lib.json 2>$OUT/bench_$lib.err
  DXV_LIBRARY=$PWD/dxrvoxelizer_amd/$lib timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench20_$lib.json 2>$OUT/bench20_$lib.err
done
tail -n 3 $OUT/q_*_1.jsonl; cat $OUT/bench*.json | cut -c1-300
