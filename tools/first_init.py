#!/usr/bin/env python3
"""First-in-process cost of a build: (a) cold, (b) after a tiny mesh has gone through the same kernels (code loaded, nothing of the size allocated)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import dxrvoxelizer_amd as dxv
from dxrvoxelizer_amd import meshes
from bench import make_mesh
vb, ib, _ = make_mesh("torus1m")
v = dxv.Voxelizer(0)
if len(sys.argv) > 1 and sys.argv[1] == "warm":
    v.InitFromArrays(*meshes.torus(60, 30))
    v.Voxelize(64)
t0 = time.perf_counter(); v.InitFromArrays(vb, ib); t1 = time.perf_counter()
st = v.stats()
print(json.dumps({"mode": sys.argv[1] if len(sys.argv) > 1 else "cold", "init_wall_ms": round((t1 - t0) * 1e3, 3), "lbvh_ms": round(st["build_ms"], 3), "list_ms": round(st["list_ms"], 3), "upload_ms": round(st["upload_ms"], 3)}))
