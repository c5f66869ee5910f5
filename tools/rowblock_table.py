"""Parity rule: one row per wave vs 2 x 2 rows per wave over meshes and grids, with the mean
triangle extent in voxels that the launcher's choice is based on."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

v = dxv.Voxelizer(0)
for mesh in sys.argv[1:] or ["bunny", "dragon", "dragon9", "torus1m", "bunny16"]:
    vb, ib, _ = make_mesh(mesh)
    v.InitFromArrays(vb, ib)
    for N in (128, 256, 512, 1024):
        ms = {}
        for rb in (1, 2, 4, 0):
            v.set_option("rowblock", rb)
            v.Voxelize(N, dxv.MODE_PARITY)
            t = []
            for _ in range(7):
                v.Voxelize(N, dxv.MODE_PARITY)
                t.append(v.stats()["voxelize_ms"])
            ms[rb] = float(np.median(t))
            chosen = v.stats()["row_block"]
        print(json.dumps({"mesh": mesh, "tris": len(ib) // 3, "N": N, "tri_extent_voxels": round(v.stats()["tri_extent"] * N / 2, 2),
                          "ms_rows": round(ms[1], 4), "ms_2x2": round(ms[2], 4), "ms_4x4": round(ms[4], 4), "auto": chosen, "ms_auto": round(ms[0], 4)}))
