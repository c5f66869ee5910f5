#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider --durations=5) > $OUT/pytest_gpu17.log 2>&1
exit 0
