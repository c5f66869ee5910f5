#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3l
rm -rf $OUT; mkdir -p $OUT
python tools/ab_option.py quadaxis 0,1 --set lists=2,plan=2 --meshes torus1m,bunny16,dragon9,bunny,dragon --grid 512 --rounds 3 > $OUT/ab_quadaxis.jsonl 2>&1
python tools/ab_option.py quadaxis 0,1 --set lists=2,plan=2 --meshes torus1m,bunny --grid 256 --rounds 3 >> $OUT/ab_quadaxis.jsonl 2>&1
python tools/ab_option.py quadaxis 0,1 --set lists=2,plan=2 --meshes soup10m --grid 512 --rounds 2 --reps 3 >> $OUT/ab_quadaxis.jsonl 2>&1
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
exit 0
