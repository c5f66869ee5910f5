#!/bin/bash
# round 3, call N: plan order 4 (XCDs by direction sector) against order 3: time and fabric traffic
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3n
rm -rf $OUT; mkdir -p $OUT
python tools/ab_option.py planorder 3,4 --set lists=2,plan=2 --meshes torus1m,bunny16,dragon9,bunny,dragon --grid 512 --rounds 3 > $OUT/ab_planorder4.jsonl 2>&1
python tools/ab_option.py planorder 3,4 --set lists=2,plan=2 --meshes torus1m,bunny --grid 256 --rounds 3 >> $OUT/ab_planorder4.jsonl 2>&1
python tools/ab_option.py planorder 3,4 --set lists=2,plan=2 --meshes dragon9 --grid 1024 --rounds 2 --reps 5 >> $OUT/ab_planorder4.jsonl 2>&1
python tools/ab_option.py planorder 3,4 --set lists=2,plan=2 --meshes soup10m --grid 512 --rounds 2 --reps 3 >> $OUT/ab_planorder4.jsonl 2>&1
python tools/rank_times.py torus1m 512 plan=2,lists=2,planorder=3 noparity > $OUT/rank_times.jsonl 2>&1
python tools/rank_times.py torus1m 512 plan=2,lists=2,planorder=4 noparity >> $OUT/rank_times.jsonl 2>&1
export PMC_LAUNCHES=5
bash tools/gpu_pmc_quick.sh order4 torus1m 512 plan=2 planorder=4 > $OUT/pmc_order4.log 2>&1
cp gpurun_out/pmcq/order4/summary.json $OUT/pmc_order4_summary.json
bash tools/gpu_pmc_quick.sh order4_bunny16 bunny16 512 plan=2 planorder=4 > $OUT/pmc_order4_bunny16.log 2>&1
cp gpurun_out/pmcq/order4_bunny16/summary.json $OUT/pmc_order4_bunny16_summary.json
(time python -m pytest tests -m gpu -q -p no:cacheprovider -k "plan or config") > $OUT/pytest_gpu.log 2>&1
exit 0
