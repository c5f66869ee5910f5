#!/usr/bin/env python3
"""Median kernel times of the reference rule for a few meshes: lists (as the library chooses them) and tree walk.
usage: quick_times.py [--meshes torus1m,bunny16,dragon9,bunny,dragon,soup10m] [--grid 512] [--set key=value,...]"""
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
    ap.add_argument("--tree", action="store_true")
    ap.add_argument("--frames", type=int, default=0, help="also the throughput with this many voxelizations in flight (frames of the one context)")
    ap.add_argument("--fresh", action="store_true", help="also with queue and grid rebuilt on every launch (plan=2) and over the brick box (plan=0), and the queue's exhaustive check")
    a = ap.parse_args()
    v = dxv.Voxelizer(0)
    v.set_option("lists", 2)
    for kv in filter(None, a.set.split(",")):
        v.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    for mesh in a.meshes.split(","):
        vb, ib, _ = make_mesh(mesh)
        v.InitFromArrays(vb, ib)
        out = {"mesh": mesh, "N": a.grid, "build_ms": round(v.stats()["build_ms"], 3)}
        for lists in ((2, 0) if a.tree else (2,)):
            v.set_option("lists", lists)
            v.set_option("plan", 1)                        # lists_ms: the kept queue (from the third launch on dealt out by the hardware); fresh_ms: nothing carried
            v.Voxelize(a.grid, 0)
            v.Voxelize(a.grid, 0)
            ts = []
            for _ in range(a.reps):
                v.Voxelize(a.grid, 0)
                ts.append(v.stats()["voxelize_ms"])
            st = v.stats()
            tag = "lists" if lists else "tree"
            out[f"{tag}_ms"] = round(float(np.median(ts)), 4)
            if lists:
                out.update({"entries": st["list_entries"], "res": st["list_res"], "list_ms": round(st["list_ms"], 3),
                            "plan_bricks": st["plan_bricks"], "plan_waves": st["plan_waves"]})
            out[f"{tag}_solid"] = v.CountSolid()
            if lists and a.fresh:
                chk = v.plan_check()
                out.update({"live_bricks": chk["live_bricks"], "live_voxels": chk["live_voxels"], "queue_violations": chk["violations"] + chk["duplicates"]})
                for plan, name in ((2, "fresh"), (0, "box")):
                    v.set_option("plan", plan)
                    v.Voxelize(a.grid, 0)
                    ts, ps = [], []
                    for _ in range(a.reps):
                        v.Voxelize(a.grid, 0)
                        ts.append(v.stats()["voxelize_ms"]); ps.append(v.stats()["plan_ms"])
                    out[f"{name}_ms"] = round(float(np.median(ts)), 4)
                    if plan == 2:
                        out["plan_ms"] = round(float(np.median(ps)), 4)
                    out[f"{name}_solid"] = v.CountSolid()
        v.set_option("plan", 2)
        v.set_option("lists", 2)
        if a.frames > 1:
            import time
            for f in range(a.frames):
                v.Voxelize(a.grid, 0, sync=False, frameIndex=f)
            v.SyncAll()
            steps = 12
            t0 = time.perf_counter()
            for k in range(steps):
                v.Voxelize(a.grid, 0, sync=False, frameIndex=k % a.frames)
            v.SyncAll()
            out[f"ms_per_voxelize_{a.frames}_in_flight"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
            v.SetFrame(0)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
