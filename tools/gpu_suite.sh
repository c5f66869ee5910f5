#!/bin/bash
# the whole -m gpu suite + smoke, logs under gpurun_out/suite
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/suite
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
tail -3 $OUT/smoke.log
exit 0
