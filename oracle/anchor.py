"""TEST INFRASTRUCTURE: compare the oracle's reference-rule grid with the independent FP64 all-triangles tracer
(oracle/indep_fp64.c) and classify every voxel on which they differ.

DXR leaves three things implementation-defined that a float32 tracer and a float64 tracer resolve differently:
  tie        two triangles are hit at the same point (shared edge or vertex) and carry different normals there
             (split vertices, XUSGObjLoader.cpp:300-335): whichever wins decides the predicate (hlsl:137-138)
  edge       the hit lies within rounding of an edge of the triangle: one arithmetic is just inside, the other just
             outside, and the ray goes on to the next surface
  threshold  |dot(normalize(n), dir) - 0.12| is at rounding level (hlsl:5, :138)
  origin     the surface passes within rounding of the voxel centre (0 < t, hlsl:76)
Anything else is UNEXPLAINED and fails tests/test_oracle_anchor.py.

    python oracle/anchor.py [mesh ...]      # prints the classification, rewrites tests/golden/anchor.json
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

_HERE = os.path.dirname(os.path.abspath(__file__))
EPS_LEN = 4e-6          # "within rounding": distances in the normalised [-1, 1]^3 space; float32 spacing near 1 is 1.2e-7
EPS_DOT = 2e-6
_L = None


def lib():
    global _L
    if _L is None:
        so, src = os.path.join(_HERE, "libindep64.so"), os.path.join(_HERE, "indep_fp64.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "libindep64.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
        u32p = np.ctypeslib.ndpointer(np.uint32, flags="C")
        L.i64_create.restype = C.c_void_p
        L.i64_create.argtypes = [f32p, u32p, C.c_uint32, f32p]
        L.i64_destroy.argtypes = [C.c_void_p]
        L.i64_voxelize.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 5
        L.i64_probe.argtypes = [C.c_void_p] + [C.c_uint32] * 5 + [np.ctypeslib.ndpointer(np.float64, flags="C")]
        L.i64_voxels.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6
        L.i64_parity_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u32p, C.c_double, C.c_void_p, C.c_void_p]
        _L = L
    return _L


class Indep:
    def __init__(self, vb, ib, bound):
        self.vb = np.ascontiguousarray(vb, np.float32)
        self.ib = np.ascontiguousarray(ib, np.uint32)
        self.bound = np.ascontiguousarray(bound, np.float32)
        self.h = lib().i64_create(self.vb, self.ib, len(self.ib) // 3, self.bound)

    def __del__(self):
        if getattr(self, "h", None):
            lib().i64_destroy(self.h)
            self.h = None

    def voxelize(self, N):
        n = N ** 3
        occ, t, k, d, t2 = np.zeros(n, np.uint8), np.zeros(n), np.zeros(n, np.uint32), np.zeros(n), np.zeros(n)
        lib().i64_voxelize(self.h, N, *[a.ctypes.data_as(C.c_void_p) for a in (occ, t, k, d, t2)])
        return occ.reshape(N, N, N), t.reshape(N, N, N), k.reshape(N, N, N), d.reshape(N, N, N), t2.reshape(N, N, N)

    def voxels(self, N, ids):
        """the same for a list of voxel ids ((iz * N + iy) * N + ix): samples of grids too large to trace whole"""
        ids = np.ascontiguousarray(ids, np.uint64)
        n = len(ids)
        occ, t, k, d, t2 = np.zeros(n, np.uint8), np.zeros(n), np.zeros(n, np.uint32), np.zeros(n), np.zeros(n)
        lib().i64_voxels(self.h, N, n, ids.ctypes.data_as(C.c_void_p), *[a.ctypes.data_as(C.c_void_p) for a in (occ, t, k, d, t2)])
        return occ, t, k, d, t2

    def parity_rows(self, N, rows, eps=EPS_LEN):
        """+X crossing parity of whole grid rows, rows = [(iy, iz), ...]: (occ [nrows, N], near [nrows, N]); near = the count
        hinges on a crossing within eps of a triangle edge or of the voxel centre (the fill rule's business)"""
        rows = np.ascontiguousarray(rows, np.uint32).reshape(-1, 2)
        occ, near = np.zeros((len(rows), N), np.uint8), np.zeros((len(rows), N), np.uint8)
        lib().i64_parity_rows(self.h, N, len(rows), rows.reshape(-1), float(eps), occ.ctypes.data_as(C.c_void_p), near.ctypes.data_as(C.c_void_p))
        return occ, near

    def probe(self, N, ix, iy, iz, k):
        out = np.zeros(6)
        lib().i64_probe(self.h, N, ix, iy, iz, int(k), out)
        return {"t": out[0], "b": out[1:4].copy(), "dot": out[4], "parallel": bool(out[5])}

    def edge_distance(self, N, ix, iy, iz, k):
        """Signed distance (float64, numpy -- a third arithmetic) from the point where the ray's line meets the plane of
        triangle k to the nearest edge of k: >= 0 inside.  Also |t| of that point."""
        c, w = self.bound[:3].astype(np.float64), float(self.bound[3])
        p = (self.vb[self.ib[3 * k:3 * k + 3], :3].astype(np.float64) - c) / w
        o = np.array([(ix + .5) / N * 2 - 1, -((iy + .5) / N * 2 - 1), (iz + .5) / N * 2 - 1])
        d = o / np.linalg.norm(o)
        n = np.cross(p[1] - p[0], p[2] - p[0])
        den = n @ d
        if den == 0:
            return -np.inf, np.inf
        t = n @ (p[0] - o) / den
        x = o + t * d
        nn = n / np.linalg.norm(n)
        dist = []
        for i in range(3):
            a, b = p[i], p[(i + 1) % 3]
            inward = np.cross(nn, b - a)
            inward /= np.linalg.norm(inward)
            dist.append((x - a) @ inward)
        return float(min(dist)), float(t)


def classify(ind, scene, N, ix, iy, iz, fp64):
    """fp64 = (occ, t, k, dot) of the independent tracer at the voxel; returns (label, detail)."""
    occ32, t32, k32, b32, _ = scene.voxel(N, ix, iy, iz, algo=orc.ALGO_BRUTE)
    occ64, t64, k64, dot64 = fp64
    hit32, hit64 = k32 != 0xffffffff, k64 != 0xffffffff
    if hit32 and hit64 and k32 == k64:
        return ("threshold", {"dot64": dot64}) if abs(dot64 - 0.12) < EPS_DOT else ("unexplained", {"why": "same triangle, dot far from 0.12", "dot64": dot64})
    d32 = ind.edge_distance(N, ix, iy, iz, k32) if hit32 else None      # (edge margin, t) of the oracle's triangle, in float64
    d64 = ind.edge_distance(N, ix, iy, iz, k64) if hit64 else None
    if hit32 and hit64 and abs(d32[1] - d64[1]) <= EPS_LEN and d32[0] >= -EPS_LEN and d64[0] >= -EPS_LEN:
        return "tie", {"k32": int(k32), "k64": int(k64), "dt": d32[1] - d64[1], "edge32": d32[0], "edge64": d64[0]}
    if hit64 and 0 <= d64[0] <= EPS_LEN:
        return "edge", {"who": "fp64 hit within rounding of its triangle's edge", "k64": int(k64), "edge64": d64[0]}
    if hit32 and -EPS_LEN <= d32[0] <= EPS_LEN:
        return "edge", {"who": "oracle hit within rounding of its triangle's edge", "k32": int(k32), "edge32": d32[0]}
    if (hit64 and abs(d64[1]) <= EPS_LEN) or (hit32 and abs(d32[1]) <= EPS_LEN):
        return "origin", {}
    return "unexplained", {"k32": int(k32), "k64": int(k64), "t32": float(t32), "t64": float(t64), "d32": d32, "d64": d64}


def compare(name, N=64, golden=None):
    d = np.load(os.path.join(ROOT, "tests", "golden", "meshes", name + ".npz"))
    vb, ib = d["vb"], d["ib"]
    scene = orc.Scene(vb, ib)
    ind = Indep(vb, ib, scene.bound)
    occ, t, k, dot, _ = ind.voxelize(N)
    want = golden if golden is not None else scene.voxelize(N, algo=orc.ALGO_BRUTE)
    diff = np.argwhere(occ != want)
    rec = {"N": N, "solid_oracle": int(want.sum()), "solid_fp64": int(occ.sum()), "differ": len(diff), "voxels": []}
    for iz, iy, ix in diff:
        label, detail = classify(ind, scene, N, int(ix), int(iy), int(iz), (int(occ[iz, iy, ix]), float(t[iz, iy, ix]), int(k[iz, iy, ix]), float(dot[iz, iy, ix])))
        rec["voxels"].append({"ix": int(ix), "iy": int(iy), "iz": int(iz), "oracle": int(want[iz, iy, ix]), "fp64": int(occ[iz, iy, ix]), "class": label})
        if label == "unexplained":
            rec["voxels"][-1]["detail"] = json.loads(json.dumps(detail, default=str))
    rec["classes"] = {c: sum(1 for v in rec["voxels"] if v["class"] == c) for c in ("tie", "edge", "threshold", "origin", "unexplained")}
    return rec


def load_mesh(name):
    """the three assets (committed loader output) and the synthetic meshes of BASELINE.md section 4, as the fixtures make them"""
    gold = lambda n: np.load(os.path.join(ROOT, "tests", "golden", "meshes", n + ".npz"))
    if name in ("bunny", "dragon", "turingbowl"):
        d = gold(name)
        return d["vb"], d["ib"]
    from dxrvoxelizer_amd import meshes                      # (pure numpy generators; no library, no GPU)
    if name == "dragon9":
        d = gold("dragon")
        return meshes.trisect(d["vb"], d["ib"])
    if name == "bunny16":
        b = gold("bunny")
        return meshes.midpoint_subdivide(b["vb"], b["ib"], 2)
    if name == "torus1m":
        return meshes.torus()
    if name == "soup1m":
        return meshes.soup(1_000_000)
    if name == "soup10m":
        return meshes.soup()                                 # BASELINE config 5's own mesh (10,000,000 triangles)
    raise KeyError(name)


def compare_sampled(name, N, n, seed=1):
    """Reference rule on `n` random voxels of an N^3 grid (grids too large to trace whole with every triangle tested for
    every ray): half of the sample uniform over the grid, half from the shell of voxels within two voxels of a change of
    the oracle's grid along x -- where surfaces are."""
    vb, ib = load_mesh(name)
    scene = orc.Scene(vb, ib)
    ind = Indep(vb, ib, scene.bound)
    rng = np.random.default_rng(seed)
    zs = np.sort(rng.choice(N, size=min(N, 24), replace=False)).astype(np.uint32)
    slices = orc.voxelize_slices(scene, N, zs)                                   # the oracle's grid on a few slices (BVH tracer)
    edge = np.zeros_like(slices, bool)
    chg = slices[:, :, 1:] != slices[:, :, :-1]
    for sh in range(-2, 3):
        lo, hi = max(0, -sh), min(N - 1, N - 1 - sh)
        edge[:, :, lo + max(sh, 0):hi + max(sh, 0) + 0][:, :, : hi - lo] |= chg[:, :, lo:hi]
    near_ids = np.argwhere(edge)
    pick = near_ids[rng.choice(len(near_ids), size=min(n // 2, len(near_ids)), replace=False)]
    ids_a = (zs[pick[:, 0]].astype(np.uint64) * N + pick[:, 1].astype(np.uint64)) * N + pick[:, 2].astype(np.uint64)
    ids_b = rng.integers(0, N ** 3, size=n - len(ids_a), dtype=np.uint64)
    ids = np.unique(np.concatenate([ids_a, ids_b]))
    occ, t, k, dot, _ = ind.voxels(N, ids)
    rec = {"N": N, "rule": "reference", "sampled_voxels": int(len(ids)), "near_surface": int(len(ids_a)), "seed": seed, "differ": 0, "voxels": []}
    solid32 = 0
    for j, vid in enumerate(ids):
        ix, iy, iz = int(vid % N), int((vid // N) % N), int(vid // (N * N))
        occ32 = scene.voxel(N, ix, iy, iz)[0]                                   # the oracle (BVH tracer) at this voxel
        solid32 += occ32
        if occ32 == occ[j]:
            continue
        label, detail = classify(ind, scene, N, ix, iy, iz, (int(occ[j]), float(t[j]), int(k[j]), float(dot[j])))
        rec["voxels"].append({"ix": ix, "iy": iy, "iz": iz, "oracle": int(occ32), "fp64": int(occ[j]), "class": label})
        if label == "unexplained":
            rec["voxels"][-1]["detail"] = json.loads(json.dumps(detail, default=str))
    rec["differ"] = len(rec["voxels"])
    rec["solid_oracle"], rec["solid_fp64"] = int(solid32), int(occ.sum())
    rec["classes"] = {c: sum(1 for v in rec["voxels"] if v["class"] == c) for c in ("tie", "edge", "threshold", "origin", "unexplained")}
    return rec


def compare_parity(name, N, nslices=None, seed=1):
    """Parity rule: the oracle's grid against the independent float64 crossing count, whole grid rows (every voxel of the
    chosen slices; nslices = None: the whole grid).  A voxel whose count hinges on a crossing within EPS_LEN of a triangle
    edge or of the voxel centre is the fill rule's business (class "edge" when the two differ there); every other voxel
    must agree."""
    vb, ib = load_mesh(name)
    scene = orc.Scene(vb, ib)
    ind = Indep(vb, ib, scene.bound)
    rng = np.random.default_rng(seed)
    zs = np.arange(N, dtype=np.uint32) if nslices is None else np.sort(rng.choice(N, size=nslices, replace=False)).astype(np.uint32)
    want = orc.voxelize_slices(scene, N, zs, mode=orc.MODE_PARITY)               # [len(zs), N, N] (iy, ix)
    rows = np.array([(iy, iz) for iz in zs for iy in range(N)], np.uint32)
    occ, near = ind.parity_rows(N, rows)
    occ, near = occ.reshape(len(zs), N, N), near.reshape(len(zs), N, N)
    diff = occ != want
    rec = {"N": N, "rule": "parity", "slices": [int(z) for z in zs] if nslices is not None else "all", "voxels_compared": int(occ.size),
           "solid_oracle": int(want.sum()), "solid_fp64": int(occ.sum()), "near_edge_voxels": int(near.sum()), "differ": int(diff.sum()),
           "classes": {"edge": int((diff & (near != 0)).sum()), "unexplained": int((diff & (near == 0)).sum())}, "voxels": []}
    for sz, iy, ix in np.argwhere(diff)[:64]:
        rec["voxels"].append({"ix": int(ix), "iy": int(iy), "iz": int(zs[sz]), "oracle": int(want[sz, iy, ix]), "fp64": int(occ[sz, iy, ix]),
                              "class": "edge" if near[sz, iy, ix] else "unexplained"})
    return rec


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "wide":
        return main_wide()
    names = sys.argv[1:] or ["turingbowl", "bunny", "dragon"]
    path = os.path.join(ROOT, "tests", "golden", "anchor.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    g64 = np.load(os.path.join(ROOT, "tests", "golden", "grids64.npz"))
    for name in names:
        golden = np.unpackbits(g64[f"{name}_64_reference"])[: 64 ** 3].reshape(64, 64, 64)      # the oracle's brute-force grid
        rec = compare(name, 64, golden)
        # the two rules this restatement adds to a plain watertight tracer (padded-box candidacy, tn <= t): how many voxels do they change?
        d = np.load(os.path.join(ROOT, "tests", "golden", "meshes", name + ".npz"))
        plain = orc.Scene(d["vb"], d["ib"]).voxelize(64, algo=orc.ALGO_PLAIN)
        rec["plain_vs_canonical_differ"] = int((plain != golden).sum())
        out[name] = rec
        print(name, {k: v for k, v in rec.items() if k != "voxels"}, flush=True)
        with open(path, "w") as fh:
            json.dump(out, fh, indent=1, sort_keys=True)


def main_wide():
    """Round 3: the anchor beyond 64^3 of the three assets -- their 128^3 grids whole, samples of the BASELINE configurations,
    and the parity rule.  ~10 minutes of CPU; writes tests/golden/anchor_wide.json."""
    import time
    path = os.path.join(ROOT, "tests", "golden", "anchor_wide.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    jobs = [(f"{n}/128/reference", lambda n=n: compare(n, 128)) for n in ("turingbowl", "bunny", "dragon")]
    jobs += [("dragon9/512/reference/sample", lambda: compare_sampled("dragon9", 512, 60000)),
             ("torus1m/512/reference/sample", lambda: compare_sampled("torus1m", 512, 60000)),
             ("soup1m/256/reference/sample", lambda: compare_sampled("soup1m", 256, 60000)),
             ("bunny16/512/reference/sample", lambda: compare_sampled("bunny16", 512, 60000)),
             ("soup10m/512/reference/sample", lambda: compare_sampled("soup10m", 512, 20000)),      # BASELINE config 5 itself
             ("dragon9/1024/reference/sample", lambda: compare_sampled("dragon9", 1024, 20000))]    # ... config 4's own grid
    jobs += [(f"{n}/64/parity", lambda n=n: compare_parity(n, 64)) for n in ("bunny", "dragon")]
    jobs += [(f"{n}/128/parity", lambda n=n: compare_parity(n, 128)) for n in ("bunny", "dragon")]
    jobs += [("torus1m/512/parity/slices", lambda: compare_parity("torus1m", 512, 6)),
             ("dragon9/512/parity/slices", lambda: compare_parity("dragon9", 512, 6)),
             ("bunny16/512/parity/slices", lambda: compare_parity("bunny16", 512, 6))]      # (closed surfaces only: a crossing count
                                                                                               # has no inside to find in a soup)
    jobs += [(f"turingbowl/{n}/parity", lambda n=n: compare_parity("turingbowl", n)) for n in (64, 128)]
    only = sys.argv[2:]
    for key, job in jobs:
        if only and not any(o in key for o in only):
            continue
        t0 = time.time()
        rec = job()
        rec["seconds"] = round(time.time() - t0, 1)
        out[key] = rec
        print(key, {k: v for k, v in rec.items() if k not in ("voxels", "slices")}, flush=True)
        with open(path, "w") as fh:
            json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
