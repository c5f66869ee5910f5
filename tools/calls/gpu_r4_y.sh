#!/bin/bash
# map resolution against entries per texel after the texel culling: which map is faster at 512^3 (and 1024^3) for the big meshes
OUT=gpurun_out/r4y; mkdir -p $OUT
for res in 256 512; do
  timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon --reps 9 --set listres=$res > $OUT/q512_res$res.jsonl 2>&1
  timeout 600 python tools/quick_times.py --meshes dragon9,bunny --grid 1024 --reps 3 --set listres=$res > $OUT/q1024_res$res.jsonl 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4y/q*_res*.jsonl")):
    for ln in open(f):
        if ln.startswith("{"):
            d=json.loads(ln); print(f.split("/")[-1], d["mesh"], d["N"], d["lists_ms"], d["entries"], round(d["entries"]/(6*d["res"]**2),2), d["list_ms"])
PY
