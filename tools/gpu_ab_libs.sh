#!/bin/bash
# A/B several builds of libdxv.so (dxrvoxelizer_amd/libdxv_<name>.so) in one GPU session, interleaved rounds.
# usage: gpu_ab_libs.sh name1 name2 ...
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
: > $OUT/ab_libs.log
for round in 1 2 3; do
for v in "$@"; do
echo "# $v round $round" >> $OUT/ab_libs.log
DXV_LIBRARY=$GRAFT_REPO_ROOT/dxrvoxelizer_amd/libdxv_$v.so python tools/sweep.py --meshes torus1m,bunny --grids 512 --bricks 4 --stacks 0 --reps 7 >> $OUT/ab_libs.log 2>&1
done; done
grep -v build_ms $OUT/ab_libs.log | python -c "
import sys, json
cur=None; agg={}
for l in sys.stdin:
    if l.startswith('#'): cur=l.split()[1]; continue
    try: d=json.loads(l)
    except: continue
    agg.setdefault((d['mesh'],cur),[]).append(d['ms'])
for k,v in sorted(agg.items()): print(k, [round(x,3) for x in v])
"
exit 0
