#!/usr/bin/env python3
"""Launch disciplines side by side, in one process on one GPU, timed like bench.py's region (K back-to-back steps by wall clock,
best of 3): the full grid and every rank's share of the block-cyclic partition at `world` ranks, for
    prepared                   queue from Init (dxv_prepare_launch), the unqueued bricks zeroed by workgroups behind the bricks' in the same dispatch
                               (the default); ..._every_lane_scans_alone: option coop = 0; ..._clear_in_front / _clear_kernel: prepclear = 1 / 0
    unprepared                 queue built inside every launch (plan = 2, persistent waves)
    kept                       queue and zeros kept (plan = 1, hardware dispatch once a sync has read the lengths)
with 1, 2 and 3 voxelizations in flight.  Slowest share against the full grid's step of the SAME discipline with ONE in flight is
the looped estimate of the speed-up at `world` GPUs.
    python tools/prepared_ab.py [mesh] [N] [world] [zblock] [frames,...] [key=value,...]"""
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
frames_list = [int(f) for f in (sys.argv[5] if len(sys.argv) > 5 else "1,2").split(",")]
v = dxv.Voxelizer(0)
for kv in filter(None, (sys.argv[6] if len(sys.argv) > 6 else "").split(",")):
    v.set_option(kv.split("=")[0], int(kv.split("=")[1]))
vb, ib, _ = make_mesh(mesh)
v.InitFromArrays(vb, ib, gridDim=N)
for r in range(world):
    v.PrepareLaunchInterleaved(N, r, world, zb)
v.set_option("events", 0)
K = 300

DISC = {"prepared": {"prepared": 1, "prepclear": 2, "plan": 2, "coop": 1}, "prepared_every_lane_scans_alone": {"prepared": 1, "prepclear": 2, "plan": 2, "coop": 0},
        "prepared_clear_in_front": {"prepared": 1, "prepclear": 1, "plan": 2, "coop": 1}, "prepared_clear_kernel": {"prepared": 1, "prepclear": 0, "plan": 2},
        "unprepared": {"prepared": 0, "plan": 2}, "kept": {"prepared": 0, "plan": 1}}


def loop(frames, launch):
    for f in range(frames):
        launch(f)
    v.SyncAll()
    for _ in range(2):                               # (clocks up; a kept queue's lengths read)
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
for name, opts in DISC.items():
    for k, val in opts.items():
        v.set_option(k, val)
    rec = {}
    for frames in frames_list:
        full = loop(frames, lambda f: v.Voxelize(N, 0, sync=False, frameIndex=f))
        shares = [loop(frames, lambda f, r=r: v.VoxelizeInterleaved(N, r, world, zb, 0, sync=False, frameIndex=f)) for r in range(world)]
        rec[f"frames_{frames}"] = {"full_ms": round(full, 4), "slowest_share_ms": round(max(shares), 4), "sum_over_full": round(sum(shares) / full, 3),
                                   "shares_ms": [round(x, 4) for x in shares]}
    one = rec[f"frames_{frames_list[0]}"]["full_ms"]
    for frames in frames_list:
        rec[f"frames_{frames}"]["speedup_against_full_one_in_flight"] = round(one / rec[f"frames_{frames}"]["slowest_share_ms"], 2)
    out[name] = rec
print(json.dumps(out))
