"""Fuzz: small adversarial meshes (vertices snapped to the voxel-centre lattice so rays run exactly
through vertices, edges and coplanar duplicates; zero-area triangles; axis-aligned slivers) against
the BRUTE-FORCE oracle, both occupancy rules.  CPU: the product's host-compiled code; GPU: kernels."""
import numpy as np
import pytest

from dxrvoxelizer_amd import meshes


def lattice_mesh(rng, n_tris, N):
    """Triangles whose normalised coordinates fall on multiples of 1/N: voxel centres sit at odd
    multiples of 1/N, so hits through vertices/edges and equal-t ties are common."""
    q = rng.integers(-N, N + 1, size=(n_tris, 3, 3)).astype(np.float32) / np.float32(N)
    kind = rng.integers(0, 6, size=n_tris)
    for t in range(n_tris):
        if kind[t] == 0:
            q[t, 2] = q[t, 1]                                   # zero-area
        elif kind[t] == 1:
            q[t, :, rng.integers(0, 3)] = q[t, 0, 0]            # axis-aligned plane
        elif kind[t] == 2 and t > 0:
            q[t] = q[t - 1]                                     # exact duplicate (equal t, tie -> smaller index)
        elif kind[t] == 3 and t > 0:
            q[t, 0], q[t, 1] = q[t - 1, 1], q[t - 1, 0]         # shares an edge with the previous triangle
    pos = q.reshape(-1, 3)
    pos = np.concatenate([pos, [[-1, -1, -1], [1, 1, 1]]]).astype(np.float32)   # pin the bound to the unit cube
    nrm = rng.normal(size=pos.shape).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    vb = np.concatenate([pos, nrm], axis=1).astype(np.float32)
    return vb, np.arange(3 * n_tris, dtype=np.uint32)


CASES = [(seed, n) for seed, n in zip(range(12), (1, 2, 3, 5, 8, 13, 21, 40, 80, 150, 300, 600))]


@pytest.mark.parametrize("seed,n_tris", CASES)
def test_fuzz_host_code_vs_brute_force(orc, hostcheck, seed, n_tris):
    rng = np.random.default_rng(1000 + seed)
    N = 16
    vb, ib = lattice_mesh(rng, n_tris, N)
    s = orc.Scene(vb, ib)
    assert np.allclose(s.bound, [0, 0, 0, 1])
    h = hostcheck(vb, ib, s.bound)
    want = {0: s.voxelize(N, algo=orc.ALGO_BRUTE), 1: s.voxelize(N, mode=1, algo=orc.ALGO_BRUTE)}
    assert np.array_equal(s.voxelize(N), want[0]) and np.array_equal(s.voxelize(N, mode=1), want[1])   # oracle BVH too
    # 6 = postponed-leaf walk over the wide nodes; 4, 7, 8 = parity rows, one walk per 1, 2 x 2, 4 x 4 rows
    for mode in (0, 1, 2, 3, 4, 6, 7, 8, 10, 11):           # 10, 11: parity rows over the four-box nodes
        g, ovf = h.voxelize(N, mode)
        assert not ovf
        assert np.array_equal(g, want[0 if mode in (0, 2, 6) else 1]), (seed, mode)
    for z0, nz in ((3, 5), (10, 1)):                     # odd slabs: a block repeats its last slice
        for mode in (7, 8, 11):
            g, _ = h.voxelize(N, mode, z0, nz)
            assert np.array_equal(g, want[1][z0:z0 + nz]), (seed, mode, z0, nz)


@pytest.mark.gpu
def test_fuzz_gpu_vs_brute_force(dxvlib, orc):
    import dxrvoxelizer_amd as dxv
    v = dxv.Voxelizer(0)
    for seed, n_tris in CASES:
        rng = np.random.default_rng(1000 + seed)
        for N in (16, 32):
            vb, ib = lattice_mesh(rng, n_tris, N)
            s = orc.Scene(vb, ib)
            v.InitFromArrays(vb, ib)
            for mode in (0, 1):
                want = s.voxelize(N, mode=mode, algo=orc.ALGO_BRUTE)
                if mode == 0:                             # the direction-space lists (default), a coarse and a fine map
                    v.set_option("lists", 2)              # (2: from the first launch of a scene on)
                    for res in (0, 16, 512):
                        v.set_option("listres", res)
                        v.Voxelize(N, mode)
                        assert np.array_equal(v.Grid(), want), (seed, N, "lists", res)
                    v.set_option("listres", 0)
                v.set_option("lists", 0)                  # ... and every tree walk
                for rows, queue, wide in (((1, 1, 1), (1, 1, 2), (1, 1, 0), (1, 0, 0)) if mode == 0 else ((1, 1, 1), (0, 1, 1), (0, 0, 1))):
                    v.set_option("rows", rows)
                    v.set_option("queue", queue)
                    v.set_option("wide", wide)
                    for rowblock in ((1, 2, 4) if mode == 1 and rows == 1 else (0,)):
                        v.set_option("rowblock", rowblock)
                        v.Voxelize(N, mode)
                        assert np.array_equal(v.Grid(), want), (seed, N, mode, rows, queue, wide, rowblock)
                    v.set_option("rowblock", 0)
                v.set_option("lists", 1)
            v.set_option("rows", 1)
            v.set_option("queue", 1)
            v.set_option("wide", 2)
    v.close()


@pytest.mark.gpu
def test_randomised_soak_short(dxvlib, orc):
    """tools/gpu_soak.py for a few seconds: random meshes x grids x partitions x every option."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_soak.py"), "12", "777"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and '"soak": "ok"' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_lists_edge_cases(dxvlib, orc):
    """tools/edge_cases.py: the direction-space lists on a single triangle at grids 2..16, triangles through
    and clustered at the grid centre (whole-face footprints, the fall-back to the tree), near-collinear
    slivers of three lengths -- coarsest, automatic and finest maps, each grid against brute force."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "edge_cases.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("BAD 0"), r.stdout[-2000:] + r.stderr[-2000:]
