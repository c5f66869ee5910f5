#!/bin/bash
# round 3, call B: order of the regions inside a dispatch plan; list resolution against grid size
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "plan or update_vertices or kept_memset or launcher") > $OUT/pytest_gpu.log 2>&1
for o in 0 1 2 3; do
python tools/rank_times.py torus1m 512 plan=2,lists=2,planorder=$o noparity zb8 >> $OUT/rank_times.jsonl 2>&1
done
for o in 0 1 2 3; do
python tools/rank_times.py bunny16 512 plan=2,lists=2,planorder=$o noparity zb8 >> $OUT/rank_times.jsonl 2>&1
done
python tools/ab_option.py planorder 0,1,2,3 --set lists=2,plan=2 --meshes torus1m,bunny16,dragon9 --grid 512 --rounds 2 > $OUT/ab_planorder.jsonl 2>&1
python tools/ab_option.py listres 128,256 --set lists=2,plan=2 --meshes torus1m,bunny,dragon --grid 256 --rounds 2 > $OUT/ab_listres.jsonl 2>&1
python tools/ab_option.py listres 256,512 --set lists=2,plan=2 --meshes dragon9,torus1m --grid 1024 --rounds 2 --reps 5 >> $OUT/ab_listres.jsonl 2>&1
python tools/ab_option.py listres 64,128 --set lists=2,plan=2 --meshes bunny,torus1m --grid 128 --rounds 2 >> $OUT/ab_listres.jsonl 2>&1
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python bench.py --gpus 2 --backend gloo --same-device > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err
exit 0
