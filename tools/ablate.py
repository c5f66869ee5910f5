#!/usr/bin/env python3
"""Timing-only ablations of the lists kernel (option `ablate`: the grids are wrong by design): where the launch's time goes.
   0 = the real kernel, 8 = nothing but launch + early-outs + stores, 1 = + texel lookup, 2 = + entry scan (no triangle test),
   6 = the same without normals, 4 = everything but the normal fetch + predicate.
usage: ablate.py [--meshes torus1m,bunny16] [--grid 512] [--reps 7]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
# the ablation kernels are not in the product library: this tool loads libdxv_ablate.so (python -m dxrvoxelizer_amd.build
# --ablate, cross-compiled in the build container; it travels to the GPU box with the snapshot)
os.environ.setdefault("DXV_LIBRARY", os.path.abspath(os.path.join(ROOT, "dxrvoxelizer_amd", "libdxv_ablate.so")))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--meshes", default="torus1m,bunny16,dragon9")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--values", default="0,8,1,2,4")
    a = ap.parse_args()
    v = dxv.Voxelizer(0)
    v.set_option("lists", 2)
    v.set_option("plan", 0)                      # (the ablations are variants of the plain launch)
    for mesh in a.meshes.split(","):
        vb, ib, _ = make_mesh(mesh)
        v.InitFromArrays(vb, ib)
        out = {"mesh": mesh, "N": a.grid}
        for val in [int(x) for x in a.values.split(",")]:
            v.set_option("ablate", val)
            v.Voxelize(a.grid, 0)
            ts = []
            for _ in range(a.reps):
                v.Voxelize(a.grid, 0)
                ts.append(v.stats()["voxelize_ms"])
            out[f"ms[{val}]"] = round(float(np.median(ts)), 4)
        v.set_option("ablate", 0)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
