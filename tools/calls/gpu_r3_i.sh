#!/bin/bash
# round 3, call I: region size of the dispatch plan, Z block of the partition
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3i
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "plan or config") > $OUT/pytest_gpu.log 2>&1
python tools/ab_option.py planregion 6,7,8,9 --set lists=2,plan=2 --meshes torus1m,bunny16,dragon9 --grid 512 --rounds 2 > $OUT/ab_planregion.jsonl 2>&1
python tools/ab_option.py planregion 6,7,8,9 --set lists=2,plan=2 --meshes torus1m --grid 256 --rounds 2 >> $OUT/ab_planregion.jsonl 2>&1
for r in 6 7 8 9; do
python tools/rank_times.py torus1m 512 plan=2,lists=2,planregion=$r noparity zb8 >> $OUT/rank_times_region.jsonl 2>&1
done
python tools/rank_times.py torus1m 512 plan=2,lists=2 noparity > $OUT/rank_times_zb.jsonl 2>&1
python tools/rank_times.py bunny16 512 plan=2,lists=2 noparity >> $OUT/rank_times_zb.jsonl 2>&1
exit 0
