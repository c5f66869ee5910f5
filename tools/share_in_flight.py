#!/usr/bin/env python3
"""A rank's share of the block-cyclic partition with 1, 2 and 3 voxelizations in flight (frames of the one context), looped on one
GPU like bench.py's timed region: ms per step of every rank's share against the full grid's step -- what two frames in flight buy
a multi-GPU run (the reference keeps FrameCount = 3 grids in flight, Content/Voxelizer.h:24).
    python tools/share_in_flight.py [mesh] [N] [world] [zblock] [key=value,...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "torus1m"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
zb = int(sys.argv[4]) if len(sys.argv) > 4 else 4
v = dxv.Voxelizer(0)
for kv in filter(None, (sys.argv[5] if len(sys.argv) > 5 else "").split(",")):
    v.set_option(kv.split("=")[0], int(kv.split("=")[1]))
vb, ib, _ = make_mesh(mesh)
v.InitFromArrays(vb, ib)
v.set_option("events", 0)
K = 300


def loop(frames, launch):
    for f in range(frames):
        launch(f)
    v.SyncAll()
    for _ in range(3):                               # (clocks up)
        for k in range(40):
            launch(k % frames)
        v.SyncAll()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for k in range(K):
            launch(k % frames)
        v.SyncAll()
        best = min(best, (time.perf_counter() - t0) / K * 1e3)
    return best


out = {"mesh": mesh, "N": N, "world": world, "zblock": zb, "steps": K}
for frames in (1, 2, 3):
    full = loop(frames, lambda f: v.Voxelize(N, 0, sync=False, frameIndex=f))
    shares = [loop(frames, lambda f, r=r: v.VoxelizeInterleaved(N, r, world, zb, 0, sync=False, frameIndex=f)) for r in range(world)]
    out[f"frames_{frames}"] = {"full_ms": round(full, 4), "slowest_share_ms": round(max(shares), 4), "shares_ms": [round(x, 4) for x in shares]}
one = out["frames_1"]["full_ms"]
for frames in (1, 2, 3):
    out[f"frames_{frames}"]["speedup_against_one_gpu_one_in_flight"] = round(one / out[f"frames_{frames}"]["slowest_share_ms"], 2)
print(json.dumps(out))
