#!/bin/bash
# round 5, call a: GPU suite on the new queue build (fused clear, two headers, run length by partition size), the A/B of the levers
# on one box, the L1 / dispatch micro-benchmark
OUT=gpurun_out/r5a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 900 python tools/tail_ab.py --check --reps 7 --sets "base:planregion=8,fuse=0;r8f1:planregion=8,fuse=1;r7f1:planregion=7,fuse=1;r6f1:planregion=6,fuse=1;r6f0:planregion=6,fuse=0" > $OUT/tail_ab.jsonl 2> $OUT/tail_ab.err; tail -3 $OUT/tail_ab.err
timeout 120 tools/micro/l1_roof > $OUT/l1_roof.jsonl 2>&1
cat $OUT/l1_roof.jsonl
python - <<'PY'
import json
for ln in open("gpurun_out/r5a/tail_ab.jsonl"):
    d=json.loads(ln)
    print(d["mesh"], d["set"], "kept", d["kept"]["full_ms"], d["kept"]["slowest_rank_ms"], d["kept"]["ideal_speedup"], d["kept"]["g256_ms"], "| fresh", d["fresh"]["full_ms"], d["fresh"]["slowest_rank_ms"], d["fresh"]["ideal_speedup"], d["fresh"]["g256_ms"])
PY
