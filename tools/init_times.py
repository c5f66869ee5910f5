#!/usr/bin/env python3
"""Where the cold call's host time goes: wall clock around each C-ABI call of Init (dxv_set_mesh, dxv_build,
dxv_build_lists_for_grid) and the first Voxelize, on a context that has done the same before (allocations exist, code loaded).
    python tools/init_times.py [mesh] [N] [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "torus1m"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
vb, ib, _ = make_mesh(mesh)
vb = np.ascontiguousarray(vb, np.float32).reshape(-1, 6)
ib = np.ascontiguousarray(ib, np.uint32).reshape(-1)
v = dxv.Voxelizer(0)
lib, ctx = v._lib, v._ctx
v.InitFromArrays(vb, ib)
v.Voxelize(N)
for rep in range(reps):
    t = [time.perf_counter()]
    assert lib.dxv_set_mesh(ctx, vb, len(vb), ib, ib.size // 3) == 0; t.append(time.perf_counter())
    assert lib.dxv_build(ctx) == 0; t.append(time.perf_counter())
    assert lib.dxv_build_lists_for_grid(ctx, 0) == 0; t.append(time.perf_counter())
    assert lib.dxv_voxelize(ctx, N, 0, 0, N) == 0; t.append(time.perf_counter())
    st = v.stats()
    ms = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
    print(json.dumps({"mesh": mesh, "N": N, "set_mesh_wall_ms": round(ms[0], 3), "build_wall_ms": round(ms[1], 3), "lists_wall_ms": round(ms[2], 3),
                      "first_voxelize_wall_ms": round(ms[3], 3), "init_wall_ms": round(sum(ms[:3]), 3),
                      "gpu_ms": {"upload": round(st["upload_ms"], 3), "lbvh": round(st["build_ms"], 3), "lists": round(st["list_ms"], 3), "voxelize": round(st["voxelize_ms"], 3)}}))
# ... and through the Python mirror, as bench.py's config.first_voxelize_after_init does it
for rep in range(3):
    t0 = time.perf_counter()
    v.InitFromArrays(vb, ib)
    t1 = time.perf_counter()
    v.Voxelize(N)
    t2 = time.perf_counter()
    print(json.dumps({"mesh": mesh, "N": N, "InitFromArrays_wall_ms": round((t1 - t0) * 1e3, 3), "first_Voxelize_wall_ms": round((t2 - t1) * 1e3, 3)}))
# the same after the context has used its other frames and kept queues (what bench.py has done by then)
v.set_option("plan", 1)
for f in (0, 1, 2):
    v.Voxelize(N, frameIndex=f)
    v.Voxelize(N, frameIndex=f)
v.set_option("plan", 2)
v.SetFrame(0)
from dxrvoxelizer_amd import meshes  # noqa: E402
other = meshes.torus(900, 500)                  # 900,000 triangles: another mesh of about the size in between, as in bench.py
for rep in range(3):
    v.InitFromArrays(*other)
    v.Voxelize(N)
    t0 = time.perf_counter()
    v.InitFromArrays(vb, ib)
    t1 = time.perf_counter()
    v.Voxelize(N)
    t2 = time.perf_counter()
    print(json.dumps({"mesh": mesh, "N": N, "after": "three frames used, another mesh of 900,000 triangles resident", "InitFromArrays_wall_ms": round((t1 - t0) * 1e3, 3), "first_Voxelize_wall_ms": round((t2 - t1) * 1e3, 3)}))
v.close()
