#!/bin/bash
OUT=gpurun_out/r5o; mkdir -p $OUT
timeout 900 python tools/tail_ab.py --reps 9 --meshes torus1m,bunny16 --sets "off:queuemin=0;m8:queuemin=8;m10:queuemin=10;m12:queuemin=12;m16:queuemin=16" > $OUT/tail_queuemin.jsonl 2>> $OUT/err.log
tail -5 $OUT/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5o/tail_*.jsonl")):
    for ln in open(f):
        d=json.loads(ln)
        print(d["mesh"], d["set"], "kept", d["kept"]["full_ms"], d["kept"]["slowest_rank_ms"], d["kept"]["ideal_speedup"], d["kept"]["g256_ms"], "| fresh", d["fresh"]["full_ms"], d["fresh"]["slowest_rank_ms"], d["fresh"]["ideal_speedup"], d["fresh"]["g256_ms"], d["fresh"].get("queue_build_ms"))
PY
