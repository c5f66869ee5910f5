#!/bin/bash
# round 3, call F: suite (C++ multi-GPU host, wide copy on demand), build timings
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
python tools/build_once.py soup10m 4 > $OUT/build_soup10m.jsonl 2>&1
python tools/build_once.py torus1m 4 > $OUT/build_torus1m.jsonl 2>&1
python tools/build_bench.py bunny torus1m soup10m > $OUT/build_bench.jsonl 2>&1
python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon,soup10m --tree > $OUT/quick_times.jsonl 2>&1
exit 0
