"""Randomised soak of the HIP path against the oracle (test infrastructure, like tests/): random meshes
(lattice-snapped adversarial triangles, soups of various densities, closed shapes), random even grid
sizes, random slabs / block-cyclic partitions, every walk variant and option, both rules.  Every grid
must equal the oracle's bit for bit.  usage: gpu_soak.py [seconds] [seed]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dxrvoxelizer_amd as dxv  # noqa: E402
from dxrvoxelizer_amd import camera, meshes  # noqa: E402
from oracle import orc  # noqa: E402
from test_fuzz import lattice_mesh, needle_mesh  # noqa: E402


def random_mesh(rng):
    kind = rng.integers(0, 7)
    if kind == 6:
        n = int(rng.choice([5, 60, 400, 2500]))
        return needle_mesh(rng, n, int(rng.choice([8, 16, 32]))), f"needles{n}"
    if kind == 0:
        n = int(rng.choice([1, 2, 3, 7, 30, 200, 1500]))
        return lattice_mesh(rng, n, int(rng.choice([8, 16, 32]))), f"lattice{n}"
    if kind == 1:
        n = int(rng.choice([50, 1000, 20000, 150000]))
        return meshes.soup(n, seed=int(rng.integers(1, 1 << 30)), edge=float(rng.choice([0.01, 0.05, 0.3]))), f"soup{n}"
    if kind == 2:
        a, b = int(rng.integers(3, 200)), int(rng.integers(3, 100))
        return meshes.torus(a, b), f"torus{a}x{b}"
    if kind == 3:
        a, b = int(rng.integers(3, 120)), int(rng.integers(2, 60))
        c = tuple(float(x) for x in rng.uniform(-0.3, 0.3, 3))
        return meshes.uv_sphere(a, b, float(rng.uniform(0.2, 0.9)), c), f"sphere{a}x{b}"
    if kind == 4:
        vb, ib = meshes.cube() if rng.integers(0, 2) else meshes.tetrahedron()
        return (vb, ib), "solid"
    d = np.load(os.path.join(ROOT, "tests", "golden", "meshes", str(rng.choice(["bunny", "dragon"])) + ".npz"))
    return (d["vb"], d["ib"]), "asset"


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    rng = np.random.default_rng(seed)
    v = dxv.Voxelizer(0)
    t0, cases, grids, checked, classes, queued = time.time(), 0, 0, 0, 0, 0
    while time.time() - t0 < budget:
        (vb, ib), label = random_mesh(rng)
        T = len(ib) // 3
        try:
            s = orc.Scene(vb, ib)
        except Exception:
            continue
        v.set_option("wide", int(rng.integers(0, 3)))
        v.set_option("refit", int(rng.choice([1, 1, 2, 0])) if T < 30000 else int(rng.choice([1, 2])))   # box merge of the build
        (v.InitFromArrays if rng.integers(0, 2) else v.InitDynamic)(vb, ib)      # the mirrors' Init (lists built) or the C-ABI's own rules
        if rng.integers(0, 3) == 0:                       # same vertices again: the refit must reproduce the build's boxes
            v.set_option("deferboxes", int(rng.integers(0, 2)))       # ... at once, or when the first tree walk asks for them
            v.UpdateVertices(np.ascontiguousarray(vb, np.float32))
        cases += 1
        for _ in range(4):
            big = T > 5000
            N = int(rng.choice([8, 16, 30, 32, 64, 96] if not big else [32, 64]))
            mode = int(rng.integers(0, 2))
            want = s.voxelize(N, mode=mode, algo=orc.ALGO_BRUTE if T * N ** 3 < 3e8 else orc.ALGO_BVH)
            opts = {"queue": int(rng.integers(0, 2)), "rows": int(rng.integers(0, 2)), "rowblock": int(rng.choice([0, 1, 2, 4])),
                    "wide": int(rng.integers(0, 3)), "brick": int(rng.integers(0, 8)), "stack": int(rng.choice([0, 0, 12, 16, 32])),
                    "subbox": int(rng.integers(0, 2)), "morton": int(rng.integers(0, 2)),
                    "lists": int(rng.integers(0, 3)), "listres": int(rng.choice([0, 0, 16, 64, 512, 2048])),
                    "plists": int(rng.integers(0, 3)), "plistres": int(rng.choice([0, 0, 16, 128, 1024])),
                    "plan": int(rng.integers(0, 3)), "queuewaves": int(rng.choice([0, 0, 8, 64, 1000])),
                    "dispatch": int(rng.choice([1, 1, 0, 2])), "planregion": int(rng.choice([0, 0, 6, 7, 8])),
                    "planheavy": int(rng.choice([0, 0, 0, 3, 65535])), "fuse": int(rng.choice([1, 1, 0])), "queueheads": int(rng.choice([8, 8, 8, 1, 2, 4])),
                    "prepared": int(rng.choice([1, 1, 1, 0])), "prepclear": int(rng.integers(0, 4)), "coop": int(rng.choice([1, 1, 0])),
                    "farmap": int(rng.choice([1, 1, 0])), "listedwaves": int(rng.choice([0, 0, 32, 28, 16, 8]))}
            for k, val in opts.items():
                v.set_option(k, val)
            part = int(rng.integers(0, 3))
            prep = bool(rng.integers(0, 2))                   # the partition's work queue prepared before the launches (Init-time structure) or not
            v.SetFrame(int(rng.integers(0, 3)))               # any of the context's frames in flight
            again = int(rng.choice([1, 1, 2, 3]))             # the same launch again: kept memsets and work queues from the second launch on
            try:
                if part == 0:
                    if prep:
                        v.PrepareLaunch(N)
                    for _ in range(again):
                        v.Voxelize(N, mode)
                    got, ref = v.Grid(), want
                elif part == 1:
                    z0 = int(rng.integers(0, N)); nz = int(rng.integers(1, N - z0 + 1))
                    if prep:
                        v.PrepareLaunch(N, z0, nz)
                    for _ in range(again):
                        v.Voxelize(N, mode, z0, nz)
                    got, ref = v.Grid(), want[z0:z0 + nz]
                else:
                    world = int(rng.choice([1, 2, 4])); zb = int(rng.choice([1, 2, 4, 8]))
                    if N % (world * zb):
                        continue
                    rank = int(rng.integers(0, world))
                    if prep:
                        v.PrepareLaunchInterleaved(N, rank, world, zb)
                    for _ in range(again):
                        v.VoxelizeInterleaved(N, rank, world, zb, mode)
                    zs = np.concatenate([np.arange(b, b + zb) for b in range(rank * zb, N, world * zb)])
                    got, ref = v.Grid(), want[zs]
            except dxv.DxvError as e:
                if opts["stack"] and "stack" in str(e):     # a forced shallow column may legitimately run out
                    continue
                raise
            grids += 1
            if not np.array_equal(got, ref):
                print(json.dumps({"FAIL": label, "T": T, "N": N, "mode": mode, "part": part, "opts": opts, "seed": seed,
                                  "differ": int((got != ref).sum())}))
                sys.exit(1)
            assert np.array_equal(v.GridBits(), np.packbits(got.reshape(-1), bitorder="little"))
            if mode == 0 and v.stats()["plan_bricks"] > 0:
                # the launch went through a work queue: no live ray in a brick that was not queued, no brick queued twice
                chk = v.plan_check()
                if chk["violations"] or chk["duplicates"] or chk["queued_bricks"] != v.stats()["plan_bricks"]:
                    print(json.dumps({"FAIL": label, "plan_check": chk, "T": T, "N": N, "part": part, "opts": opts, "seed": seed}))
                    sys.exit(1)
                queued += chk["queued_bricks"]
            if mode == 0 and T * N ** 3 < 2e9 and rng.integers(0, 4) == 0:
                # the brick test of the launches over the brick box: no ray of a brick it calls dead hits anything (the triangles' own far-radius
                # map, and the lists' mip when the scene has lists)
                for lists_mip in ((False, True) if v.stats()["list_entries"] > 0 else (False,)):
                    fc = v.far_check(N, lists_mip=lists_mip)
                    if fc["violations"]:
                        print(json.dumps({"FAIL": label, "far_check": fc, "lists_mip": lists_mip, "T": T, "N": N, "opts": opts, "seed": seed}))
                        sys.exit(1)
            if mode == 0 and v.stats()["list_entries"] > 0 and T * N ** 3 < 2e9:
                # the lists in use, exhaustively: every triangle the canonical step accepts for a ray is selectable from its list
                accepted, violations, first = v.list_check(N)
                if violations:
                    print(json.dumps({"FAIL": label, "list_check": int(violations), "first": [list(map(int, x)) for x in first[:4]], "T": T, "N": N,
                                      "opts": opts, "seed": seed}))
                    sys.exit(1)
                checked += accepted
            if mode == 0 and T * N ** 3 < 2e9 and rng.integers(0, 3) == 0:
                classified, wrong, hits, first = v.class_check(N)      # the per-triangle class of the normal test against the predicate
                if wrong:
                    print(json.dumps({"FAIL": label, "class_check": int(wrong), "first": [list(map(int, x)) for x in first[:4]], "T": T, "N": N, "seed": seed}))
                    sys.exit(1)
                classes += classified
            if part == 0 and rng.integers(0, 4) == 0:      # display pass: the empty-brick skip changes no pixel
                eye, vp = camera.default_view_proj(96, 64, eye=tuple(float(x) for x in rng.uniform(-12, 12, 3) + np.array([0, 0, 14.0])))
                v.set_option("skipempty", 1)
                a = v.Render(eye, vp, 96, 64)
                v.set_option("skipempty", 0)
                b = v.Render(eye, vp, 96, 64)
                v.set_option("skipempty", 1)
                assert np.array_equal(a, b), (label, N)
    print(json.dumps({"soak": "ok", "seconds": round(time.time() - t0, 1), "meshes": cases, "grids": grids, "list_pairs_checked": int(checked), "class_hits_checked": int(classes), "queued_bricks_checked": int(queued), "seed": seed}))


if __name__ == "__main__":
    main()
