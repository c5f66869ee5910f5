#!/bin/bash
OUT=gpurun_out/r5j; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "work_queue or config_grid or headline or first_second" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 900 python tools/tail_ab.py --check --reps 9 --meshes torus1m,bunny16 --sets "steal0:queuesteal=0;steal1:queuesteal=1" > $OUT/tail_steal.jsonl 2>> $OUT/err.log
tail -5 $OUT/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5j/tail_*.jsonl")):
    for ln in open(f):
        d=json.loads(ln)
        print(d["mesh"], d["set"], "kept", d["kept"]["full_ms"], d["kept"]["slowest_rank_ms"], d["kept"]["ideal_speedup"], d["kept"]["g256_ms"], "| fresh", d["fresh"]["full_ms"], d["fresh"]["slowest_rank_ms"], d["fresh"]["ideal_speedup"], d["fresh"]["g256_ms"], d["fresh"].get("queue_build_ms"))
PY
