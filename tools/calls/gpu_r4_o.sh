#!/bin/bash
OUT=gpurun_out/r4o; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deferred or refit or update or frames" > $OUT/pytest_subset.log 2>&1; tail -30 $OUT/pytest_subset.log
