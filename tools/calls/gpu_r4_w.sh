#!/bin/bash
# parity rule, round 3's library against this round's on one box (configs: bunny 256^3, torus-1M 512^3)
OUT=gpurun_out/r4w; mkdir -p $OUT
cat > /tmp/par.py <<'PY'
import sys, json, numpy as np
sys.path.insert(0, ".")
import dxrvoxelizer_amd as dxv
from bench import make_mesh
v = dxv.Voxelizer(0)
for mesh, N in (("torus1m", 512), ("bunny", 256), ("dragon9", 512)):
    vb, ib, _ = make_mesh(mesh)
    v.InitFromArrays(vb, ib)
    for _ in range(3): v.Voxelize(N, 1)
    ts = []
    for _ in range(21):
        v.Voxelize(N, 1); ts.append(v.stats()["voxelize_ms"])
    print(json.dumps({"mesh": mesh, "N": N, "parity_ms": round(float(np.median(ts)), 4), "solid": v.CountSolid()}))
PY
for rep in 1 2; do
  (cd .ab_old && python /tmp/par.py) > $OUT/par_r3_$rep.jsonl 2>&1
  python /tmp/par.py > $OUT/par_r4_$rep.jsonl 2>&1
done
tail -n 3 $OUT/par_*.jsonl
