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


def needle_mesh(rng, n_tris, N):
    """Off-lattice needles and slivers aimed at the rays: short edge 1e-7 .. 1e-4, long edge 0.003 .. 0.5, centred ON the
    radial ray of a random voxel centre at radius 0.5 .. 1.7, the long edge within 1e-4 .. 0.3 rad of that ray (grazing
    to face-on), a third of them split into two triangles sharing the long edge.  What the lists' margins must survive:
    edge functions that all but vanish, footprints a fraction of a texel wide, radial extents of half the scene."""
    tris = []
    while len(tris) < n_tris:
        c = (rng.integers(0, N, 3) + 0.5) / N * 2 - 1
        c[1] = -c[1]
        d = c / np.linalg.norm(c)
        rho = rng.uniform(0.5, 1.7)
        p = d * rho
        if np.abs(p).max() > 0.98 or rho <= np.linalg.norm(c):
            continue                                            # keep the needle inside the cube and in front of the ray
        a = rng.normal(size=3)
        a -= a.dot(d) * d
        a /= np.linalg.norm(a)                                  # unit vector across the ray
        b = np.cross(d, a)
        ang = 10.0 ** rng.uniform(-4, -0.5)
        long_dir = np.cos(ang) * d + np.sin(ang) * (np.cos(1.7) * a + np.sin(1.7) * b)
        L, w = 10.0 ** rng.uniform(-2.5, -0.3), 10.0 ** rng.uniform(-7, -4)
        off = rng.uniform(-0.5, 0.5) * w * a                    # the ray passes within the needle's width of its axis
        v0, v1 = p - 0.5 * L * long_dir + off, p + 0.5 * L * long_dir + off
        v2 = p + rng.uniform(-0.5, 0.5) * L * long_dir + w * a + off
        if max(np.abs(v0).max(), np.abs(v1).max(), np.abs(v2).max()) > 0.99:
            continue
        tris.append([v0, v1, v2])
        if rng.random() < 0.33 and len(tris) < n_tris:          # the other half of a thin quad: shared long edge
            tris.append([v1, v0, p - w * a + off])
    pos = np.asarray(tris, np.float64).reshape(-1, 3)
    pos = np.concatenate([pos, [[-1, -1, -1], [1, 1, 1]]]).astype(np.float32)   # pin the bound to the unit cube
    nrm = rng.normal(size=pos.shape).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    return np.concatenate([pos, nrm], axis=1).astype(np.float32), np.arange(3 * len(tris), dtype=np.uint32)


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


def test_needles_host_lists_vs_brute_force(orc, hostcheck):
    """The product's list code (footprints, texel-local boxes and edges, radial words, scan) compiled for the CPU, on
    needles aimed at the rays, coarse to fine maps, against the oracle's brute force."""
    rng = np.random.default_rng(4242)
    for n_tris, N in ((40, 16), (200, 16), (120, 32)):
        vb, ib = needle_mesh(rng, n_tris, N)
        s = orc.Scene(vb, ib)
        assert np.allclose(s.bound, [0, 0, 0, 1])
        want = s.voxelize(N, algo=orc.ALGO_BRUTE)
        assert want.sum() > 0
        h = hostcheck(vb, ib, s.bound)
        for R in (16, 256):
            h.lists(R)
            got, ovf = h.voxelize(N, mode=12, stack=8)
            assert ovf == 0 and np.array_equal(got, want), (n_tris, N, R)


@pytest.mark.gpu
def test_needles_and_slivers_lists_superset(dxvlib, orc):
    """Adversarial coverage of the lists' superset claim on the device: needles and slivers (short edge 1e-7 .. 1e-4) aimed
    at the rays, map resolutions 16 / 256 / 4096, every grid against the oracle's brute force, and the exhaustive on-device
    check (dxv_debug_list_check): every triangle the canonical step accepts for a ray is selectable from the ray's list."""
    import dxrvoxelizer_amd as dxv
    v = dxv.Voxelizer(0)
    v.set_option("lists", 2)
    rng = np.random.default_rng(2024)
    pairs = 0
    for n_tris, N in ((1, 16), (30, 16), (300, 32), (1500, 32), (600, 64)):
        vb, ib = needle_mesh(rng, n_tris, N)
        s = orc.Scene(vb, ib)
        want = s.voxelize(N, algo=orc.ALGO_BRUTE)
        v.InitFromArrays(vb, ib)
        for res in (16, 256, 4096):
            v.set_option("listres", res)
            v.Voxelize(N)
            assert v.stats()["list_entries"] > 0 and v.stats()["list_res"] == res
            assert np.array_equal(v.Grid(), want), (n_tris, N, res)
            accepted, violations, first = v.list_check(N)
            assert violations == 0, (n_tris, N, res, first)
            pairs += accepted
        v.set_option("listres", 0)
    assert pairs > 2000                                        # the needles are hit: the check is not vacuous
    # the assets as well, at the resolution the library picks
    for name, N in (("bunny", 128), ("turingbowl", 96)):
        d = np.load(__import__("os").path.join(__import__("conftest").GOLD, "meshes", name + ".npz"))
        v.InitFromArrays(d["vb"], d["ib"])
        accepted, violations, first = v.list_check(N)
        assert accepted > 0 and violations == 0, (name, first)
    v.close()


@pytest.mark.gpu
def test_lists_frustum_corner_triangles(dxvlib, orc):
    """The soak failure of round 2 on the device (tests/test_hostcheck.py::frustum_corner_case: near radius of a triangle
    whose edge runs through a frustum corner), and lattice-snapped triangles -- which put edges on frustum corners and side
    planes all the time -- at a size the short soak does not reach: grids and the exhaustive superset check."""
    import dxrvoxelizer_amd as dxv
    from test_hostcheck import frustum_corner_case
    v = dxv.Voxelizer(0)
    v.set_option("lists", 2)
    vb, ib = frustum_corner_case()
    want = orc.Scene(vb, ib).voxelize(96, algo=orc.ALGO_BRUTE)
    v.InitFromArrays(vb, ib)
    for res in (16, 64, 1024):
        v.set_option("listres", res)
        v.Voxelize(96)
        assert np.array_equal(v.Grid(), want), res
        assert v.list_check(96)[1] == 0
    rng = np.random.default_rng(77001)
    for n_tris, L, N in ((1500, 32, 96), (1500, 16, 64), (400, 8, 48)):
        vb, ib = lattice_mesh(rng, n_tris, L)
        want = orc.Scene(vb, ib).voxelize(N, algo=orc.ALGO_BRUTE)
        v.InitFromArrays(vb, ib)
        for res in (64, 0):
            v.set_option("listres", res)
            v.Voxelize(N)
            assert np.array_equal(v.Grid(), want), (n_tris, L, N, res)
            if v.stats()["list_entries"] == 0:                  # cube-spanning triangles on the automatic (fine) map: over the
                assert res == 0                                 # size cap, the tree walk answered
                continue
            accepted, violations, first = v.list_check(N)
            assert violations == 0 and accepted > 0, (n_tris, L, N, res, first)
    v.set_option("listres", 0)


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
