#!/usr/bin/env python3
"""Persistent waves per launch against the grid size: the step with nothing carried (plan = 2) for a list of meshes and grids, with
7168 (all the GPU holds), 6144, 5120 and 4096 waves, and with what the library picks by itself (dxv_policy.h: queue_waves_sevenths)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv
from bench import make_mesh

meshes = (sys.argv[1] if len(sys.argv) > 1 else "torus1m,bunny16,dragon9,bunny,dragon").split(",")
grids = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "128,256,512").split(",")]
v = dxv.Voxelizer(0)
for mesh in meshes:
    vb, ib, _ = make_mesh(mesh)
    v.InitFromArrays(vb, ib)
    for N in grids:
        row = {"mesh": mesh, "N": N}
        for rnd in range(2):
            for w in (7168, 6144, 5120, 4096, 0):
                v.set_option("queuewaves", w)
                for _ in range(3):
                    v.Voxelize(N)
                ts = []
                for _ in range(11):
                    v.Voxelize(N)
                    ts.append(v.stats()["voxelize_ms"])
                row.setdefault(f"w{w}" if w else "auto", []).append(round(float(np.median(ts)), 4))
        row["bricks"] = v.stats()["plan_bricks"]
        print(json.dumps(row), flush=True)
v.set_option("queuewaves", 0)
