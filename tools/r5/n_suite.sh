#!/bin/bash
OUT=gpurun_out/r5n; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python tools/refit_loop.py torus1m 512 40 > $OUT/refit_loop.jsonl 2>&1; tail -3 $OUT/refit_loop.jsonl | cut -c1-600
timeout 600 python tools/gpu_soak.py 120 51515 > $OUT/soak.jsonl 2>&1; tail -2 $OUT/soak.jsonl | cut -c1-400
