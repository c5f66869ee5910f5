"""A/B one option of libdxv.so inside one process: interleaved rounds of both settings.
usage: ab_option.py OPTION V0,V1 [--meshes torus1m,bunny] [--grid 512] [--mode reference] [--reps 9] [--set lists=0]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("option")
    ap.add_argument("values")
    ap.add_argument("--meshes", default="torus1m,bunny,dragon")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--mode", default="reference")
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--set", default="", help="other options held fixed, e.g. lists=0,brick=4")
    a = ap.parse_args()
    values = [int(v) for v in a.values.split(",")]
    mode = dxv.MODE_REFERENCE if a.mode == "reference" else dxv.MODE_PARITY
    v = dxv.Voxelizer(0)
    for kv in filter(None, a.set.split(",")):
        v.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    for mesh in a.meshes.split(","):
        vb, ib, _ = make_mesh(mesh)
        v.InitFromArrays(vb, ib)
        ms = {val: [] for val in values}
        solid = {}
        for _ in range(a.rounds):
            for val in values:
                v.set_option(a.option, val)
                v.Voxelize(a.grid, mode)                      # warm-up (and stack adaptation)
                for _ in range(a.reps):
                    v.Voxelize(a.grid, mode)
                    ms[val].append(v.stats()["voxelize_ms"])
                solid[val] = v.CountSolid()
        assert len(set(solid.values())) == 1, solid
        print(json.dumps({"mesh": mesh, "N": a.grid, "mode": a.mode, "option": a.option, "stack": v.stats()["stack_entries"],
                          **{f"median_ms[{val}]": float(np.median(ms[val])) for val in values},
                          **{f"min_ms[{val}]": float(np.min(ms[val])) for val in values}, "solid": solid[values[0]]}))


if __name__ == "__main__":
    main()
