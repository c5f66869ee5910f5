#!/bin/bash
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny --reps 9 --set listres=1024 > $OUT/q512_res1024.jsonl 2>&1
for res in 128 256 512; do timeout 600 python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 --reps 9 --set listres=$res > $OUT/q256_res$res.jsonl 2>&1; done
for res in 512 1024; do timeout 600 python tools/quick_times.py --meshes dragon9,bunny,torus1m --grid 1024 --reps 3 --set listres=$res > $OUT/q1024_res$res.jsonl 2>&1; done
for res in 128 256 512; do timeout 600 python tools/quick_times.py --meshes bunny,dragon --grid 128 --reps 9 --set listres=$res > $OUT/q128_res$res.jsonl 2>&1; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4z/q*_res*.jsonl")):
    for ln in open(f):
        if ln.startswith("{"):
            d=json.loads(ln); print(f.split("/")[-1], d["mesh"], d["N"], d["lists_ms"], d["entries"], round(d["entries"]/(6*d["res"]**2),2), d["list_ms"])
PY
