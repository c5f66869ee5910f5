#!/usr/bin/env python3
"""Diagnostic (library built with -DDXV_PHASE_TIMES: python -c "from dxrvoxelizer_amd import build; build.build(defines=['DXV_PHASE_TIMES'], name='phases')"):
where a brick's TIME goes in the hardware-dispatched lists kernel -- ticks of the wave's own instruction stream between stamps, summed
over all bricks of K prepared launches: ray set-up + the texel's cell, start search, scan rounds, direction / shear at a flush, triangle
rounds, predicate + stores.   usage: DXV_LIBRARY=.../libdxv_phases.so phase_times.py [mesh] [grid] [K] [key=value,...]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "torus1m"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
v = dxv.Voxelizer(0)
for kv in filter(None, (sys.argv[4] if len(sys.argv) > 4 else "").split(",")):
    v.set_option(kv.split("=")[0], int(kv.split("=")[1]))
vb, ib, _ = make_mesh(mesh)
if os.environ.get("PHASE_TEXELS"):
    v.EnableTexels(True)                                   # (the reference's texel image beside the grid: the normal of every inside hit)
v.InitFromArrays(vb, ib, gridDim=N)
v.Voxelize(N)
raw = np.zeros(16, np.uint64)
v._check(v._lib.dxv_debug_download(v._ctx, 102, raw.ctypes.data_as(C.c_void_p), raw.nbytes))      # reset
ms = []
for _ in range(K):
    v.Voxelize(N)
    ms.append(v.stats()["voxelize_ms"])
v._check(v._lib.dxv_debug_download(v._ctx, 101, raw.ctypes.data_as(C.c_void_p), raw.nbytes))
names = ["setup_and_cell", "start_search", "scan_rounds", "direction_and_shear", "triangle_rounds", "predicate_and_stores"]
bricks = int(raw[6]) or 1
us = {n: float(raw[i]) / 100.0 / bricks for i, n in enumerate(names)}
tot = sum(us.values())
print(json.dumps({"mesh": mesh, "N": N, "texels": bool(os.environ.get("PHASE_TEXELS")), "launches": K, "kernel_ms_instrumented": round(float(np.median(ms)), 4), "bricks_per_launch": bricks // K,
                  "us_per_brick": {k: round(x, 2) for k, x in us.items()}, "us_per_brick_total": round(tot, 2),
                  "share": {k: round(x / tot, 3) for k, x in us.items()},
                  "scan_rounds_per_brick": round(float(raw[7]) / bricks, 2), "triangle_rounds_per_brick": round(float(raw[8]) / bricks, 2), "flushes_per_brick": round(float(raw[9]) / bricks, 2)}))
