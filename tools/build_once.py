#!/usr/bin/env python3
"""Minimal driver for profilers (no torch): build the LBVH of a mesh K times, then the direction-space lists K times.
    python3 tools/build_once.py [mesh] [K] [key=value ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "torus1m"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
v = dxv.Voxelizer(0)
for kv in sys.argv[3:]:
    k, val = kv.split("=")
    v.set_option(k, int(val))
vb, ib, _ = make_mesh(mesh)
rows = []
for k in range(K):
    v.InitFromArrays(vb, ib)
    st = v.stats()
    t0 = time.perf_counter()
    v.build_lists()
    wall = (time.perf_counter() - t0) * 1e3
    v.Voxelize(64)
    s2 = v.stats()
    rows.append({"build_ms": st["build_ms"], "prep_ms": st["prep_ms"], "sort_ms": st["sort_ms"], "hierarchy_ms": st["hierarchy_ms"],
                 "refit_ms": st["refit_ms"], "list_ms": s2["list_ms"], "list_wall_ms": wall, "list_entries": s2["list_entries"], "list_res": s2["list_res"]})
for r in rows:
    print(json.dumps({"mesh": mesh, **{k: (round(x, 4) if isinstance(x, float) else x) for k, x in r.items()}}))
v.close()
