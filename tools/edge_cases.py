"""Edge cases of the direction-space lists against the oracle's brute force (test infrastructure, like tests/):
single triangle, triangles through / clustered at the grid centre, near-collinear slivers.  Prints BAD <n>."""
import sys, numpy as np, json
import os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dxrvoxelizer_amd as dxv
from oracle import orc
rng = np.random.default_rng(11)
v = dxv.Voxelizer(0)
v.set_option("lists", 2)
bad = 0
def one(vb, ib, Ns, label, res=(0, 16, 4096)):
    global bad
    vb = np.ascontiguousarray(vb, np.float32); ib = np.ascontiguousarray(ib, np.uint32)
    s = orc.Scene(vb, ib)
    v.InitFromArrays(vb, ib)
    for N in Ns:
        want = s.voxelize(N, algo=orc.ALGO_BRUTE)
        for r in res:
            v.set_option("listres", r)
            v.Voxelize(N)
            st = v.stats()
            ok = np.array_equal(v.Grid(), want)
            bad += not ok
            print(json.dumps({"case": label, "N": N, "listres": r, "entries": st["list_entries"], "res": st["list_res"], "ok": bool(ok)}))
    v.set_option("listres", 0)
n = np.array([[0, 0, 1]], np.float32)
tri = np.hstack([np.array([[0.3, -0.2, 0.1], [0.9, 0.4, 0.2], [0.1, 0.8, -0.5]], np.float32), np.repeat(n, 3, 0)])
one(tri, np.arange(3), (2, 4, 6, 16), "one triangle")
c = np.hstack([np.array([[-0.01, -0.01, 0.0], [0.02, -0.01, 0.0], [0.0, 0.02, 0.0], [-1, -1, -1], [1, 1, 1], [1, -1, 0.5]], np.float32), np.repeat(n, 6, 0)])
one(c, np.arange(6), (2, 8, 32), "triangle through the centre + big one")
pts = rng.uniform(-1e-3, 1e-3, (300, 3)).astype(np.float32); pts[0] = [-1, -1, -1]; pts[1] = [1, 1, 1]; pts[2] = [1, -1, 0]
one(np.hstack([pts, np.repeat(n, 300, 0)]), np.arange(300), (8, 32), "cluster at the centre", res=(0, 64))
# slivers along diagonals
sl = []
for k in range(60):
    a = rng.uniform(-0.9, 0.9, 3); d = rng.uniform(-1, 1, 3); d /= np.linalg.norm(d)
    L = rng.choice([1e-3, 0.05, 0.8]); e = rng.uniform(-1, 1, 3) * 1e-7
    sl += [a, a + d * L, a + d * L * 0.5 + e]
sl = np.array(sl, np.float32)
one(np.hstack([sl, np.repeat(n, len(sl), 0)]), np.arange(len(sl)), (16, 64), "slivers", res=(0, 256))
print("BAD", bad)
