"""Wall-clock per step of back-to-back asynchronous launches of one rank's share (what bench.py times at N ranks), with and
without the library's two HIP events per launch.  usage: step_loop.py [mesh] [N] [world] [zblock] [steps]"""
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
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 200
v = dxv.Voxelizer(0)
v.set_option("lists", 2)
vb, ib, _ = make_mesh(mesh)
v.InitFromArrays(vb, ib)
for rounds in range(3):
    for ev in (1, 0):
        v.set_option("events", ev)
        out = {"mesh": mesh, "N": N, "world": world, "zblock": zb, "events": ev, "steps": steps}
        per = []
        for rank in (0, world // 2):
            for _ in range(4):
                v.VoxelizeInterleaved(N, rank, world, zb, sync=False) if world > 1 else v.Voxelize(N, sync=False)
            v.Sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                v.VoxelizeInterleaved(N, rank, world, zb, sync=False) if world > 1 else v.Voxelize(N, sync=False)
            v.Sync()
            per.append((time.perf_counter() - t0) / steps * 1e3)
        out["ms_per_step_rank0_rankmid"] = [round(x, 4) for x in per]
        print(json.dumps(out), flush=True)
v.set_option("events", 1)
