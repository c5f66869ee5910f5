#!/bin/bash
# Refresh only the bench line and the rocprof kernel stats of the same command (gpurun_out/final/).
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/prof_bench.log 2>&1
exit 0
