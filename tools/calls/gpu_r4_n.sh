#!/bin/bash
# deferred list check + counting pass in the refit: tests that touch refit / lists / frames, then the refit loop
OUT=gpurun_out/r4n; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "refit or update or frame or list or queue or import or export or soak or dynamic or option" > $OUT/pytest_subset.log 2>&1; tail -5 $OUT/pytest_subset.log
timeout 600 python tools/refit_loop.py torus1m 512 60 > $OUT/refit_loop.jsonl 2>&1; cut -c1-260 $OUT/refit_loop.jsonl
timeout 600 python tools/refit_loop.py bunny16 512 30 >> $OUT/refit_loop.jsonl 2>&1; tail -6 $OUT/refit_loop.jsonl | cut -c1-260
