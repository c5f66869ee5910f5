#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3k
rm -rf $OUT; mkdir -p $OUT
: > $OUT/ab_libs.log
for round in 1 2 3; do
for v in old new; do
echo "# $v round $round" >> $OUT/ab_libs.log
DXV_LIBRARY=$GRAFT_REPO_ROOT/dxrvoxelizer_amd/libdxv_$v.so python tools/sweep.py --meshes torus1m,bunny16,dragon9 --grids 512 --bricks 4 --stacks 0 --reps 7 --opts lists=2,plan=2 >> $OUT/ab_libs.log 2>&1
done; done
for v in old new; do
echo "# $v soup" >> $OUT/ab_libs.log
DXV_LIBRARY=$GRAFT_REPO_ROOT/dxrvoxelizer_amd/libdxv_$v.so python tools/sweep.py --meshes soup10m --grids 512 --bricks 4 --stacks 0 --reps 3 --opts lists=2,plan=2 >> $OUT/ab_libs.log 2>&1
DXV_LIBRARY=$GRAFT_REPO_ROOT/dxrvoxelizer_amd/libdxv_$v.so python tools/sweep.py --meshes torus1m,bunny --grids 256 --bricks 4 --stacks 0 --reps 7 --opts lists=2,plan=2 >> $OUT/ab_libs.log 2>&1
done
grep -v build_ms $OUT/ab_libs.log | python -c "
import sys, json
cur=None; agg={}
for l in sys.stdin:
    if l.startswith('#'): cur=l.split()[1]; continue
    try: d=json.loads(l)
    except: continue
    agg.setdefault((d['mesh'],d.get('N'),cur),[]).append(d['ms'])
for k,v in sorted(agg.items()): print(k, [round(x,3) for x in v])
" > $OUT/ab_libs_summary.txt
exit 0
