"""Generate tests/golden/* from the REFERENCE loader build (oracle/_ref) and the oracle.

Run here (where /root/reference exists):  python oracle/gen_fixtures.py [--brute]
  meshes/<name>.npz     vb/ib/aabb exactly as the reference's ObjLoader::Import produces them
                        for the three shipped assets (inputs of every parity test)
  obj/<name>.npz        the same for the hand-written OBJ files in tests/golden/obj
  grids.json            per (mesh, N, mode): solid count, SHA-256 and per-slice popcounts of the
                        oracle grid; 64^3 grids also bit-packed in grids64.npz
With --brute the 64^3 reference-mode grids come from the brute-force oracle (minutes); otherwise
from the oracle's BVH path (tests assert brute == BVH at 32^3 every run).
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

ASSETS = "/root/reference/Bin/Assets"
GOLD = os.path.join(ROOT, "tests", "golden")


def digest(grid):
    return {"solid": int(grid.sum()), "sha256": hashlib.sha256(np.ascontiguousarray(grid).tobytes()).hexdigest(),
            "slices": [int(x) for x in grid.reshape(grid.shape[0], -1).sum(1)]}


def main():
    brute = "--brute" in sys.argv
    os.makedirs(os.path.join(GOLD, "meshes"), exist_ok=True)
    grids, packed = {}, {}
    for name, file in (("bunny", "bunny.obj"), ("dragon", "dragon.obj"), ("turingbowl", "TuringBowl.obj")):
        vb, ib, aabb = orc.ref_objloader(os.path.join(ASSETS, file))
        assert vb.shape[1] == 6
        # positions have <= 4 decimals and normals are recomputed, but keep the reference's exact bits
        np.savez_compressed(os.path.join(GOLD, "meshes", name + ".npz"), vb=vb, ib=ib, aabb=aabb)
        scene = orc.Scene(vb, ib)
        for N, mode in ((32, 0), (32, 1), (64, 0), (64, 1), (128, 0), (256, 0), (256, 1)):
            if N == 256 and name == "turingbowl":
                continue
            algo = orc.ALGO_BRUTE if (brute and N == 64 and mode == 0) else orc.ALGO_BVH
            g = scene.voxelize(N, mode=mode, algo=algo)
            key = f"{name}/{N}/{'reference' if mode == 0 else 'parity'}"
            grids[key] = digest(g)
            grids[key]["oracle_algo"] = "brute" if algo == orc.ALGO_BRUTE else "bvh"
            if N == 64:
                packed[key.replace("/", "_")] = np.packbits(g.reshape(-1))
            print(key, grids[key]["solid"], grids[key]["oracle_algo"], flush=True)
        if True:
            g, tex = scene.voxelize(64, mode=0, texels=True)
            grids[f"{name}/64/texels"] = {"sha256": hashlib.sha256(tex.tobytes()).hexdigest(),
                                          "nonzero": int((tex != 0).sum())}
    for f in sorted(os.listdir(os.path.join(GOLD, "obj"))):
        if f.endswith(".obj"):
            vb, ib, aabb = orc.ref_objloader(os.path.join(GOLD, "obj", f))
            # a file with vt makes the reference's stride 32 (XUSGObjLoader.cpp:160): the extra 8
            # bytes per vertex are never written; the voxelizer's vertex is the first 24 bytes
            stride = vb.shape[1] * 4
            vb = np.ascontiguousarray(vb[:, :6])
            np.savez_compressed(os.path.join(GOLD, "obj", f[:-4] + ".npz"), vb=vb, ib=ib, aabb=aabb, stride=stride)
            print("obj", f, vb.shape, ib.shape)
    with open(os.path.join(GOLD, "grids.json"), "w") as fh:
        json.dump(grids, fh, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(GOLD, "grids64.npz"), **packed)


if __name__ == "__main__":
    main()
