"""Per-rank kernel times of the block-cyclic Z partition, looped on one GPU: what an N-GPU run can at
best reach (the slowest rank decides).  usage: rank_times.py [mesh] [N] [key=value,...] [noparity]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "torus1m"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
v = dxv.Voxelizer(0)
opts = sys.argv[3] if len(sys.argv) > 3 else ""                      # e.g. plan=0,lists=2
for kv in filter(None, opts.split(",")):
    v.set_option(kv.split("=")[0], int(kv.split("=")[1]))
vb, ib, _ = make_mesh(mesh)
v.InitFromArrays(vb, ib)
for mode in ((0,) if "noparity" in sys.argv else (0, 1)):
    v.Voxelize(N, mode)
    full = float(np.median([(v.Voxelize(N, mode), v.stats()["voxelize_ms"])[1] for _ in range(7)]))
    for world in (2, 4, 8):
        for zb in ((8,) if "zb8" in sys.argv else (2, 4, 8, 16)):
            if N % (world * zb):
                continue
            t = []
            for r in range(world):
                v.VoxelizeInterleaved(N, r, world, zb, mode)
                ts = []
                for _ in range(5):
                    v.VoxelizeInterleaved(N, r, world, zb, mode)
                    ts.append(v.stats()["voxelize_ms"])
                t.append(float(np.median(ts)))
            print(json.dumps({"mesh": mesh, "N": N, "mode": "reference" if mode == 0 else "parity", "options": opts, "full_ms": round(full, 3), "world": world,
                              "zblock": zb, "rank_ms": [round(x, 3) for x in t], "ideal_speedup": round(full / max(t), 2),
                              "sum_over_full": round(sum(t) / full, 2)}))
