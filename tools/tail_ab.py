#!/usr/bin/env python3
"""The end of a short launch, measured: a rank's share of the block-cyclic Z partition looped on one GPU, the 256^3 grid and the
full grid, with the queue kept (plan = 1) and with nothing carried (plan = 2), for a list of option sets on ONE box in ONE process
(the sets alternate inside every repetition, so clock drift hits all of them alike).

usage: tail_ab.py [--meshes torus1m,bunny16] [--sets "a:planregion=8,fuse=0;b:planregion=6,fuse=1"] [--reps 9] [--world 8] [--zblock 4]
Each figure is the median over reps of the library's own events around one launch (everything the launch puts into the stream)."""
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
    ap.add_argument("--meshes", default="torus1m,bunny16")
    ap.add_argument("--sets", default="base:planregion=8,fuse=0;new:planregion=0,fuse=1")
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--zblock", type=int, default=4)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--check", action="store_true", help="whole-grid equality of every set against the first, and the queue's exhaustive check")
    a = ap.parse_args()
    sets = []
    for item in filter(None, a.sets.split(";")):
        name, _, kv = item.partition(":")
        sets.append((name, [(k.split("=")[0], int(k.split("=")[1])) for k in filter(None, kv.split(","))]))
    v = dxv.Voxelizer(0)
    v.set_option("lists", 2)
    N = a.grid
    for mesh in a.meshes.split(","):
        vb, ib, _ = make_mesh(mesh)
        v.InitFromArrays(vb, ib)
        v.build_lists(grid=N)
        acc = {name: {"full_kept": [], "full_fresh": [], "g256_kept": [], "g256_fresh": [], "rank_kept": [[] for _ in range(a.world)],
                      "rank_fresh": [[] for _ in range(a.world)], "plan_full": [], "plan_rank0": []} for name, _ in sets}
        ref_grids = {}

        def apply(opts, plan):
            for k, val in opts:
                v.set_option(k, val)
            v.set_option("plan", plan)

        def one(fn):
            fn()
            return v.stats()["voxelize_ms"]

        for rep in range(a.reps):
            for name, opts in sets:
                for plan, tag in ((1, "kept"), (2, "fresh")):
                    apply(opts, plan)
                    v.Voxelize(N, 0); v.Voxelize(N, 0)                    # (kept: the second one has the lengths; both: warm)
                    acc[name]["full_" + tag].append(one(lambda: v.Voxelize(N, 0)))
                    if plan == 2:
                        acc[name]["plan_full"].append(v.stats()["plan_ms"])
                    if a.check and rep == 0:
                        g = v.Grid()
                        key = ("full", 0)
                        if key not in ref_grids:
                            ref_grids[key] = g
                        assert np.array_equal(g, ref_grids[key]), (mesh, name, tag, "full grid differs")
                        chk = v.plan_check()
                        assert not chk["violations"] and not chk["duplicates"], (mesh, name, tag, chk)
                    v.Voxelize(N // 2, 0); v.Voxelize(N // 2, 0)
                    acc[name]["g256_" + tag].append(one(lambda: v.Voxelize(N // 2, 0)))
                    for r in range(a.world):
                        f = lambda: v.VoxelizeInterleaved(N, r, a.world, a.zblock, 0)        # noqa: E731
                        f(); f()
                        acc[name]["rank_" + tag][r].append(one(f))
                        if plan == 2 and r == 0:
                            acc[name]["plan_rank0"].append(v.stats()["plan_ms"])
                        if a.check and rep == 0:
                            g = v.Grid()
                            key = ("rank", r)
                            if key not in ref_grids:
                                ref_grids[key] = g
                            assert np.array_equal(g, ref_grids[key]), (mesh, name, tag, r, "share differs")
                            chk = v.plan_check()
                            assert not chk["violations"] and not chk["duplicates"], (mesh, name, tag, r, chk)
        for name, opts in sets:
            d = acc[name]
            med = lambda xs: float(np.median(xs))                         # noqa: E731
            out = {"mesh": mesh, "N": N, "set": name, "options": dict(opts), "world": a.world, "zblock": a.zblock, "reps": a.reps}
            for tag in ("kept", "fresh"):
                full, g256 = med(d["full_" + tag]), med(d["g256_" + tag])
                ranks = [med(x) for x in d["rank_" + tag]]
                out[tag] = {"full_ms": round(full, 4), "rank_ms": [round(x, 4) for x in ranks], "slowest_rank_ms": round(max(ranks), 4),
                            "ideal_speedup": round(full / max(ranks), 2), "sum_over_full": round(sum(ranks) / full, 3),
                            "g256_ms": round(g256, 4), "g256_gvoxels_s": round((N // 2) ** 3 / g256 / 1e6, 1)}
            out["fresh"]["queue_build_ms"] = {"full": round(med(d["plan_full"]), 4), "rank0": round(med(d["plan_rank0"]), 4)}
            print(json.dumps(out), flush=True)
    v.close()


if __name__ == "__main__":
    main()
