#!/bin/bash
# round 3, call V: lanes of a wave that look into the same texel start their scans together (merged line accesses)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3v
rm -rf $OUT; mkdir -p $OUT
python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "golden or lists_equal or configs or plan or fuzz" > $OUT/pytest_gpu.log 2>&1
for round in 1 2; do for v in old align; do
echo "# $v round $round" >> $OUT/ab.log
DXV_LIBRARY=$GRAFT_REPO_ROOT/dxrvoxelizer_amd/libdxv_$v.so python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,soup10m >> $OUT/ab.log 2>&1
done; done
exit 0
