#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4m; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REFIT_ONLY=device rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/refit_loop.py ${MESH:-torus1m} 512 12 > $OUT/loop.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/trace_gaps.py $OUT/trace 3 > $OUT/gaps.jsonl; cat $OUT/gaps.jsonl | cut -c1-6000; tail -3 $OUT/loop.log
find $OUT/trace -name "*.csv" -size +2M -delete
