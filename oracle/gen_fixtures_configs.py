"""Generate tests/golden/configs.json: whole-grid digests of the CPU oracle for the BASELINE.json
configurations that are too large to commit as grids (configs 3, 4, 5 and the metric's meshes).

Run here (build container; minutes of CPU):  python oracle/gen_fixtures_configs.py [key-substring ...]

Per (mesh, N, rule): solid count, SHA-256 of the whole uint8 grid, per-slice popcounts, SHA-256 of the
mesh arrays the grid was made from (the GPU tests regenerate the synthetic meshes and must get the
same bytes) and, for the multi-GPU configuration, the SHA-256 of every rank's part under both
partitions of dxrvoxelizer_amd/slabs.py (8 contiguous Z slabs; Z blocks of 8 slices dealt round-robin).
For torus1m/512, dragon9/512 and bunny/256 also the reference's own per-frame output, the R10G10B10A2_UNORM texel image
(hlsl:83-84, Content/Voxelizer.cpp:65): SHA-256 over the whole uint32 image and every slice's wrapping 64-bit sum of its texels.
Existing entries of the file are kept unless regenerated.
Indexing follows Content/Voxelizer.cpp:366-368 and DXRVoxelizer.hlsl:64-67 (x fastest, then y, then z).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from dxrvoxelizer_amd import meshes  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(GOLD, "configs.json")


def mesh_sha(vb, ib):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(vb, np.float32).tobytes())
    h.update(np.ascontiguousarray(ib, np.uint32).tobytes())
    return h.hexdigest()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make(name):
    gold = lambda n: np.load(os.path.join(GOLD, "meshes", n + ".npz"))
    if name == "dragon9":
        d = gold("dragon")
        return meshes.trisect(d["vb"], d["ib"])
    if name == "bunny16":
        b = gold("bunny")
        return meshes.midpoint_subdivide(b["vb"], b["ib"], 2)
    if name == "torus1m":
        return meshes.torus()
    if name == "soup10m":
        return meshes.soup()
    if name == "soup1m":
        return meshes.soup(1_000_000)
    if name in ("bunny", "dragon"):
        d = gold(name)
        return d["vb"], d["ib"]
    raise KeyError(name)


# (mesh, N, mode, partitions?)
JOBS = [("dragon9", 512, 0, False), ("torus1m", 256, 0, False), ("torus1m", 512, 0, False), ("torus1m", 512, 1, False),
        ("bunny16", 512, 0, False), ("soup1m", 256, 0, False), ("dragon9", 1024, 0, True), ("soup10m", 512, 0, False), ("bunny", 256, 0, False)]
TEXELS = {"torus1m/512/reference", "dragon9/512/reference", "bunny/256/reference"}


def digest(scene, N, mode, partitions, texels=False):
    hw, ht = hashlib.sha256(), hashlib.sha256()
    slices, tsums, chunk = [], [], 16
    grid = np.empty((N, N, N), np.uint8) if partitions else None
    for z0 in range(0, N, chunk):
        if texels:
            g, t = scene.voxelize(N, mode=mode, z0=z0, nz=chunk, texels=True)
            ht.update(t.tobytes())
            tsums += [int(x) for x in t.reshape(chunk, -1).sum(1, dtype=np.uint64)]
        else:
            g = scene.voxelize(N, mode=mode, z0=z0, nz=chunk)
        hw.update(g.tobytes())
        slices += [int(x) for x in g.reshape(chunk, -1).sum(1)]
        if partitions:
            grid[z0:z0 + chunk] = g
    rec = {"solid": int(sum(slices)), "sha256": hw.hexdigest(), "slices": slices}
    if texels:
        rec["texels_sha256"], rec["texel_slice_sums"] = ht.hexdigest(), tsums
    if partitions:
        W, blk = 8, 8
        rec["slabs8_sha256"] = [sha(grid[r * N // W:(r + 1) * N // W]) for r in range(W)]
        g4 = grid.reshape(N // (blk * W), W, blk, N, N)       # [round, rank, slice in block, y, x]
        rec["cyclic8x8_sha256"] = [sha(g4[:, r]) for r in range(W)]
    return rec


def main():
    want = sys.argv[1:]
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    cache = {}
    for name, N, mode, parts in JOBS:
        key = f"{name}/{N}/{'reference' if mode == 0 else 'parity'}"
        if want and not any(w in key for w in want):
            continue
        if name not in cache:
            cache.clear()
            vb, ib = make(name)
            cache[name] = (orc.Scene(vb, ib), mesh_sha(vb, ib), len(ib) // 3, len(vb))
        scene, msha, T, V = cache[name]
        t0 = time.time()
        rec = digest(scene, N, mode, parts, texels=key in TEXELS)
        rec.update({"mesh_sha256": msha, "tris": T, "verts": V, "oracle_algo": "bvh", "oracle_s": round(time.time() - t0, 1)})
        out[key] = rec
        print(key, rec["solid"], rec["sha256"][:16], f"{rec['oracle_s']} s", flush=True)
        with open(OUT, "w") as fh:
            json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
