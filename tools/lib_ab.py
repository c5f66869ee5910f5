#!/usr/bin/env python3
"""One library's back-to-back step times (like bench.py's region: K steps by wall clock, best of 3) for the full grid and one rank's share
at 8 ranks -- run alternately with DXV_LIBRARY set to two builds on ONE box for a same-box A/B of a kernel change.
    DXV_LIBRARY=... python tools/lib_ab.py [mesh] [N] [tag] [key=value,...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "torus1m"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
tag = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(os.environ.get("DXV_LIBRARY", "libdxv.so"))
v = dxv.Voxelizer(0)
opts = sys.argv[4] if len(sys.argv) > 4 else ""
for kv in filter(None, opts.split(",")):
    if kv.split("=")[0] == "texels":                                  # (the reference's texel image beside the grid: dxv_enable_texels)
        v.EnableTexels(bool(int(kv.split("=")[1])))
    else:
        v.set_option(kv.split("=")[0], int(kv.split("=")[1]))
vb, ib, _ = make_mesh(mesh)
v.InitFromArrays(vb, ib, gridDim=N)
v.PrepareLaunchInterleaved(N, 3, 8, 4)
v.set_option("events", 0)
K = 300


def loop(launch):
    launch()
    v.SyncAll()
    for _ in range(2):
        for _ in range(40):
            launch()
        v.SyncAll()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            launch()
        v.SyncAll()
        best = min(best, (time.perf_counter() - t0) / K * 1e3)
    return best


full = loop(lambda: v.Voxelize(N, 0, sync=False))
share = loop(lambda: v.VoxelizeInterleaved(N, 3, 8, 4, 0, sync=False))
v.Voxelize(N, 0)
print(json.dumps({"lib": tag, "options": opts, "mesh": mesh, "N": N, "full_ms": round(full, 4), "share_rank3_of_8_ms": round(share, 4), "solid": v.CountSolid()}))
