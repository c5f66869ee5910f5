#!/bin/bash
# round 5, call h: heads per queue of the persistent waves
OUT=gpurun_out/r5h; mkdir -p $OUT
timeout 900 python tools/tail_ab.py --check --reps 7 --meshes torus1m,bunny16 --sets "h8:queueheads=8;h4:queueheads=4;h2:queueheads=2;h1:queueheads=1" > $OUT/tail_heads.jsonl 2>> $OUT/err.log
tail -5 $OUT/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5h/tail_*.jsonl")):
    for ln in open(f):
        d=json.loads(ln)
        print(d["mesh"], d["set"], "kept", d["kept"]["full_ms"], d["kept"]["slowest_rank_ms"], d["kept"]["ideal_speedup"], d["kept"]["g256_ms"], "| fresh", d["fresh"]["full_ms"], d["fresh"]["slowest_rank_ms"], d["fresh"]["ideal_speedup"], d["fresh"]["g256_ms"], d["fresh"].get("queue_build_ms"))
PY
