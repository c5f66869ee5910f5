"""Reference rule through the direction-space lists against the tree walk, per mesh and list resolution.
usage: ab_lists.py [--meshes a,b] [--grid 512] [--res auto,128,256,512]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--meshes", default="torus1m,bunny,dragon,bunny16,dragon9")
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--res", default="auto")
ap.add_argument("--reps", type=int, default=7)
a = ap.parse_args()
v = dxv.Voxelizer(0)
for mesh in a.meshes.split(","):
    vb, ib, _ = make_mesh(mesh)
    v.InitFromArrays(vb, ib)
    row = {"mesh": mesh, "N": a.grid}
    solid = set()
    for res in ["tree"] + a.res.split(","):
        v.set_option("lists", 0 if res == "tree" else 2)
        if res != "tree":
            v.set_option("listres", 0 if res == "auto" else int(res))
        v.Voxelize(a.grid)
        ms = []
        for _ in range(a.reps):
            v.Voxelize(a.grid)
            ms.append(v.stats()["voxelize_ms"])
        solid.add(v.CountSolid())
        st = v.stats()
        row["tree_ms" if res == "tree" else f"lists_{res}_ms"] = round(float(np.median(ms)), 3)
        if res != "tree":
            row[f"entries_{res}"] = st["list_entries"]
            row[f"res_{res}"] = st["list_res"]
            row[f"build_{res}_ms"] = round(st["list_ms"], 3)
    assert len(solid) == 1, solid
    print(json.dumps(row), flush=True)
