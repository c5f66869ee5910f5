#!/usr/bin/env python3
"""Standalone launches of the reference rule for a few meshes, median of the library's own events around each launch:
    prepared   queue from Init (dxv_prepare_launch), grid cleared inside the launch, hardware dispatch -- the default of a scene whose Init
               was told the grid
    fresh      queue built inside the launch (plan = 2, persistent waves)   kept   queue and zeros kept (plan = 1)
    box        lists over the brick box (plan = 0)                          tree   LBVH walk (lists = 0)
with the solid count of each (all equal) and the queue's exhaustive check.
usage: quick_times.py [--meshes torus1m,bunny16,dragon9,bunny,dragon,soup10m] [--grid 512] [--reps 7] [--set key=value,...] [--no-tree]"""
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
    ap.add_argument("--meshes", default="torus1m,bunny16,dragon9,bunny,dragon")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--set", default="")
    ap.add_argument("--no-tree", action="store_true")
    a = ap.parse_args()
    v = dxv.Voxelizer(0)
    for kv in filter(None, a.set.split(",")):
        v.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    N = a.grid

    def median(warm=2):
        for _ in range(warm):
            v.Voxelize(N, 0)
        ts = []
        for _ in range(a.reps):
            v.Voxelize(N, 0)
            ts.append(v.stats()["voxelize_ms"])
        return round(float(np.median(ts)), 4)

    for mesh in a.meshes.split(","):
        vb, ib, _ = make_mesh(mesh)
        v.InitFromArrays(vb, ib, gridDim=N)
        st0 = v.stats()
        out = {"mesh": mesh, "N": N, "build_ms": round(st0["build_ms"], 3), "list_ms": round(st0["list_ms"], 3), "prepare_ms": round(st0["prepare_ms"], 4)}
        solids = {}
        for name, opts in (("prepared", {"prepared": 1, "plan": 2}), ("fresh", {"prepared": 0, "plan": 2}), ("kept", {"prepared": 0, "plan": 1}),
                           ("box", {"prepared": 0, "plan": 0})):
            for k, val in opts.items():
                v.set_option(k, val)
            out[f"{name}_ms"] = median(3 if name == "kept" else 2)
            st = v.stats()
            solids[name] = v.CountSolid()
            if name == "prepared":
                out.update({"entries": st["list_entries"], "res": st["list_res"], "queued_bricks": st["plan_bricks"], "launched_prepared": bool(st["plan_prepared"])})
                if st["plan_bricks"]:
                    chk = v.plan_check()
                    out.update({"live_bricks": chk["live_bricks"], "live_voxels": chk["live_voxels"], "queue_violations": chk["violations"] + chk["duplicates"]})
            if name == "fresh":
                out["queue_build_ms"] = round(st["plan_ms"], 4)
        v.set_option("plan", 2)
        v.set_option("prepared", 1)
        if not a.no_tree:
            v.set_option("lists", 0)
            out["tree_ms"] = median()
            solids["tree"] = v.CountSolid()
            v.set_option("lists", 1)
        out["solid"] = solids["prepared"]
        out["all_equal"] = len(set(solids.values())) == 1
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
