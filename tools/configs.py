#!/usr/bin/env python3
"""BASELINE.json configs 2-5 on one GPU: timings (HIP events, median of R), solid counts and
bit-exact spot checks against the CPU oracle.  One JSON line per configuration.

    python tools/configs.py [--quick] [--no-verify]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import algorithmic_bytes, make_mesh  # noqa: E402
from dxrvoxelizer_amd import meshes  # noqa: E402
from dxrvoxelizer_amd.slabs import slab_range  # noqa: E402


def timed(v, N, mode, z0=0, nz=None, reps=5):
    v.Voxelize(N, mode, z0, nz)
    ts = []
    for _ in range(reps):
        v.Voxelize(N, mode, z0, nz)
        ts.append(v.stats()["voxelize_ms"])
    return float(np.median(ts)), float(min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()
    verify = not args.no_verify
    if verify:
        from oracle import orc
    v = dxv.Voxelizer(0)
    gold = lambda n: np.load(os.path.join(ROOT, "tests", "golden", "meshes", n + ".npz"))

    def run(name, vb, ib, N, mode=dxv.MODE_REFERENCE, slices=(), scene=None, extra=None):
        t0 = time.perf_counter()
        v.InitFromArrays(vb, ib)
        init_s = time.perf_counter() - t0
        st = v.stats()
        ms, mn = timed(v, N, mode)
        rec = {"config": name, "tris": st["num_tris"], "verts": st["num_verts"], "N": N,
               "mode": "reference" if mode == 0 else "parity", "ms": ms, "min_ms": mn, "mvox_s": N ** 3 / ms / 1e3,
               "solid": v.CountSolid(), "tree_height": st["tree_height"], "stack": v.stats()["stack_entries"],
               "build_ms": st["build_ms"], "build_stages_ms": [st["prep_ms"], st["sort_ms"], st["hierarchy_ms"], st["refit_ms"]],
               "upload_ms": st["upload_ms"], "init_wall_s": init_s,
               "algorithmic_GBps": algorithmic_bytes(N, N, st["num_tris"], st["num_verts"]) / ms / 1e6}
        if verify and slices:
            sc = scene or orc.Scene(vb, ib)
            g = v.Grid()
            want = orc.voxelize_slices(sc, N, list(slices), mode=mode)
            rec["oracle_slices"] = list(slices)
            rec["oracle_equal"] = bool(all(np.array_equal(g[z], want[i]) for i, z in enumerate(slices)))
        if extra:
            rec.update(extra(v))
        print(json.dumps(rec), flush=True)
        return rec

    b = gold("bunny")
    run("2: bunny 256^3", b["vb"], b["ib"], 256, slices=(0, 100, 128, 200))
    run("2p: bunny 256^3 parity", b["vb"], b["ib"], 256, dxv.MODE_PARITY, slices=(100, 128))
    d = gold("dragon")
    run("3a: dragon (100k) 512^3", d["vb"], d["ib"], 512, slices=(256, 300))
    g100 = v.Grid()
    vb9, ib9 = meshes.trisect(d["vb"], d["ib"])
    run("3: dragon x9 (900k) 512^3", vb9, ib9, 512, slices=(256, 300),
        extra=lambda vv: {"voxels_differing_from_100k_mesh": int((vv.Grid() != g100).sum())})
    del g100
    vb, ib, _ = make_mesh("torus1m")
    run("metric: torus-1M 256^3", vb, ib, 256, slices=(64, 128))
    run("metric: torus-1M 512^3", vb, ib, 512, slices=(90, 256))
    run("metric-p: torus-1M 512^3 parity", vb, ib, 512, dxv.MODE_PARITY, slices=(256,))
    vb16, ib16 = meshes.midpoint_subdivide(b["vb"], b["ib"], 2)
    run("metric: bunny x16 (1.11M) 512^3", vb16, ib16, 512, slices=(256,))
    if not args.quick:
        # config 4: 1024^3 in 8 Z-slabs of 128 slices, looped on this one GPU, vs a single launch
        v.InitFromArrays(vb9, ib9)
        N = 1024
        t_full, _ = timed(v, N, 0, reps=3)
        full = v.Grid()
        h_full = hashlib.sha256(full.tobytes()).hexdigest()
        solid = int(full.sum(dtype=np.uint64))
        del full
        slab_ms, hs = [], hashlib.sha256()
        for r in range(8):
            z0, nz = slab_range(N, r, 8)
            ms, _ = timed(v, N, 0, z0, nz, reps=3)
            slab_ms.append(ms)
            hs.update(v.Grid().tobytes())
        il_ms = {}
        for blk in (4, 8, 16):
            ms_r = []
            for r in range(8):
                v.VoxelizeInterleaved(N, r, 8, blk)
                ts = []
                for _ in range(3):
                    v.VoxelizeInterleaved(N, r, 8, blk)
                    ts.append(v.stats()["voxelize_ms"])
                ms_r.append(float(np.median(ts)))
            il_ms[blk] = ms_r
        print(json.dumps({"config": "4i: dragon x9 1024^3, block-cyclic Z partition over 8 ranks looped on one GPU",
                          "N": N, "full_ms": t_full, "rank_ms_by_block": il_ms,
                          "ideal_8gpu_speedup": {b: t_full / max(m) for b, m in il_ms.items()}}), flush=True)
        print(json.dumps({"config": "4: dragon x9 1024^3, 8 Z-slabs looped on one GPU", "N": N, "full_ms": t_full,
                          "full_mvox_s": N ** 3 / t_full / 1e3, "slab_ms": slab_ms, "max_slab_ms": max(slab_ms),
                          "ideal_8gpu_speedup_static_slabs": t_full / max(slab_ms), "solid": solid,
                          "slabs_concat_equal_full": hs.hexdigest() == h_full}), flush=True)
        # config 5: 10 M-triangle soup
        t0 = time.perf_counter()
        svb, sib = meshes.soup()
        gen_s = time.perf_counter() - t0
        run("5: soup-10M 512^3", svb, sib, 512, slices=(256,) if verify else (), extra=lambda vv: {"mesh_gen_s": gen_s})
    v.close()


if __name__ == "__main__":
    main()
