import sys, json, hashlib
sys.path.insert(0, '.')
import numpy as np
import dxrvoxelizer_amd as dxv
from bench import make_mesh
vb, ib, _ = make_mesh("dragon")
v = dxv.Voxelizer(0)
v.InitFromArrays(vb, ib, gridDim=2048)
out = {}
for name, opts in (("prepared", {"prepared": 1}), ("unprepared", {"prepared": 0}), ("box", {"prepared": 0, "plan": 0})):
    for k, val in opts.items():
        v.set_option(k, val)
    v.Voxelize(2048)
    v.Voxelize(2048)
    st = v.stats()
    g = v.Grid()
    out[name] = {"ms": round(st["voxelize_ms"], 3), "prepared": st["plan_prepared"], "bricks": st["plan_bricks"], "solid": int(v.CountSolid()), "sha": hashlib.sha256(g.tobytes()).hexdigest()[:16]}
    del g
print(json.dumps(out))
assert len({o["sha"] for o in out.values()}) == 1
