import sys, json, numpy as np
sys.path.insert(0, '/root/repo')
import dxrvoxelizer_amd as dxv
from bench import make_mesh
v = dxv.Voxelizer(0)
for mesh in ("torus1m", "bunny"):
    vb, ib, _ = make_mesh(mesh)
    v.InitFromArrays(vb, ib)
    out = {"mesh": mesh, "N": 512}
    for tex in (0, 1):
        v.EnableTexels(bool(tex))
        v.Voxelize(512)
        t = []
        for _ in range(9):
            v.Voxelize(512); t.append(v.stats()["voxelize_ms"])
        out["texels_on" if tex else "texels_off"] = round(float(np.median(t)), 3)
    v.EnableTexels(False)
    print(json.dumps(out))
