#!/usr/bin/env python3
"""Where the step with nothing carried loses against the kept queue: persistent waves against hardware dispatch on the SAME kept
queue (plan = 1, dispatch 0 / 1), and the step with nothing carried (plan = 2), one process, interleaved."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv
from bench import make_mesh

v = dxv.Voxelizer(0)
vb, ib, _ = make_mesh(sys.argv[1] if len(sys.argv) > 1 else "torus1m")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
v.InitFromArrays(vb, ib)
res = {}
for rnd in range(3):
    for name, opts in (("kept_hardware", {"plan": 1, "dispatch": 1}), ("kept_persistent", {"plan": 1, "dispatch": 0}), ("fresh", {"plan": 2})):
        for k, val in opts.items():
            v.set_option(k, val)
        for _ in range(3):
            v.Voxelize(N)
        ts, ps = [], []
        for _ in range(15):
            v.Voxelize(N)
            st = v.stats()
            ts.append(st["voxelize_ms"]); ps.append(st["plan_ms"])
        res.setdefault(name, []).append((round(float(np.median(ts)), 4), round(float(np.median(ps)), 4)))
print(json.dumps({"mesh": sys.argv[1] if len(sys.argv) > 1 else "torus1m", "N": N, "ms (median of 15, plan part)": res}))
