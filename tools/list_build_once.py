"""Build a mesh's scene and direction-space lists a few times (stats.list_ms of each; for rocprofv3 --kernel-trace --stats).
usage: list_build_once.py mesh [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

mesh = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
vb, ib, _ = make_mesh(mesh)
v = dxv.Voxelizer(0)
v.set_option("lists", 2)
ms = []
for _ in range(reps):
    v.InitFromArrays(vb, ib)           # (a new scene: the first launch builds its lists)
    v.Voxelize(128)
    ms.append(round(v.stats()["list_ms"], 4))
st = v.stats()
counts = v.debug(7)[:, 1] & 0xffff     # DirCell.count of every texel
print(json.dumps({"mesh": mesh, "entries": st["list_entries"], "res": st["list_res"], "list_ms": ms, "longest": int(counts.max()),
                  "texels_over_32": int((counts > 32).sum()), "texels_nonempty": int((counts > 0).sum())}))
