#!/usr/bin/env python3
"""LBVH build and refit timings (HIP events) for the three box-merge variants.
    python tools/build_bench.py [mesh ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

for mesh in (sys.argv[1:] or ["bunny", "torus1m"]):
    vb, ib, label = make_mesh(mesh)
    for refit in (0, 2, 1):
        v = dxv.Voxelizer(0)
        v.set_option("refit", refit)
        rows = []
        for _ in range(4):
            v.InitFromArrays(vb, ib)
            st = v.stats()
            rows.append([st[k] for k in ("prep_ms", "sort_ms", "hierarchy_ms", "refit_ms", "build_ms")])
        med = np.median(np.asarray(rows[1:]), axis=0)
        upd = []
        for _ in range(4):
            v.UpdateVertices(vb)
            upd.append(v.stats()["refit_ms"])
        T = len(ib) // 3
        print(json.dumps({"mesh": mesh, "tris": T, "refit_variant": ("atomic", "pyramid", "sweep")[refit],
                          "prep_ms": med[0], "sort_ms": med[1], "hierarchy_ms": med[2], "refit_ms": med[3],
                          "build_ms": med[4], "build_Mtris_s": T / med[4] / 1e3,
                          "update_refit_ms": float(np.median(upd[1:])), "tree_height": v.stats()["tree_height"]}), flush=True)
        v.close()
