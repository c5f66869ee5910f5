#!/bin/bash
# round 4, call I: suite, bench (1 rank; 2 ranks on one GPU through gloo), PMC of the queue kernel
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4i
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=5 -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python bench.py --gpus 2 --backend gloo --same-device --steps 100 > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; tail -2 $OUT/bench_2rank.err
export PMC_LAUNCHES=5
bash tools/gpu_pmc_quick.sh torus1m torus1m 512 > $OUT/pmc_torus1m.log 2>&1
cp gpurun_out/pmcq/torus1m/summary.json $OUT/pmc_torus1m_summary.json
tail -40 $OUT/pmc_torus1m.log
exit 0
