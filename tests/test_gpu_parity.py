"""Parity tests proper (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle
on the same inputs, against the committed golden vectors, and at BASELINE.json's full sizes
through size-independent properties.  Bar: bit-exact (uint8 occupancy, uint32 texels, uint64
keys, node words)."""
# NOTE on Init: the tests of this file drive the C-ABI's own rules (dxv_set_mesh + dxv_build, then launches: option lists decides
# when the candidate lists are built) through Voxelizer.InitDynamic.  The static path of the host mirrors -- Init builds the lists,
# every launch is the same launch -- is what tests/test_gpu_configs.py, the smoke test and bench.py run, plus the tests below that
# say so.
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from dxrvoxelizer_amd import meshes
from dxrvoxelizer_amd.slabs import gather_slabs, slab_range

pytestmark = pytest.mark.gpu

DBG_SORTED_KEYS, DBG_NODES, DBG_TRI_POS, DBG_TRI_NRM, DBG_PARENTS, DBG_NODES32, DBG_NODES64, DBG_LIST_CELLS, DBG_LIST_ENTRIES, DBG_LIST_MIP = range(10)


@pytest.fixture(scope="module")
def dxv(dxvlib):
    import dxrvoxelizer_amd
    return dxrvoxelizer_amd


@pytest.fixture()
def vox(dxv):
    v = dxv.Voxelizer(0)
    yield v
    v.close()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---------------------------------------------------------------------------------------------
# dxv_set_mesh: the host's look at the caller's arrays (in threads from 200 k elements up)
# ---------------------------------------------------------------------------------------------
def test_set_mesh_scan_in_threads_gives_the_sequential_bound_and_names_the_offender(dxv, orc):
    """A mesh large enough for the threaded sweep: the bound is the oracle's (a sequential sweep) bit for bit -- signed zeros at
    chunk borders included; an index out of range and a non-finite position are reported with their own position, wherever in
    the arrays they sit; a failed call leaves the context's earlier mesh alone."""
    from dxrvoxelizer_amd import meshes
    vb, ib = meshes.torus(400, 200)                                           # 160,000 triangles, 80,000 vertices
    vb = vb.copy()
    vb[::7919, 0] *= 0.0                                                       # zeros of either sign spread over the chunks
    vb[3::7919, 2] = -0.0
    v = dxv.Voxelizer(0)
    v.InitDynamic(vb, ib)
    s = orc.Scene(vb, ib)
    assert np.array_equal(np.asarray(v.stats()["bound"], np.float32).view(np.uint32), s.bound.view(np.uint32))
    v.Voxelize(64)
    want = v.Grid().copy()
    for pos in (0, len(ib) // 2 + 1, len(ib) - 1):
        bad = ib.copy()
        bad[pos] = len(vb) + 5
        with pytest.raises(dxv.DxvError, match=f"index {len(vb) + 5} at position {pos} out of range"):
            v.InitDynamic(vb, bad)
    for vert in (0, len(vb) // 3, len(vb) - 1):
        for val in (np.nan, np.inf, -np.inf):
            bad = vb.copy()
            bad[vert, 1] = val
            with pytest.raises(dxv.DxvError, match=f"vertex {vert} has a non-finite position"):
                v.InitDynamic(bad, ib)
    v.Voxelize(64)                                                             # the earlier mesh is still there
    assert np.array_equal(v.Grid(), want)
    # ... and so is its BOUND after a mesh of zero extent was refused (the bound used to be overwritten before the check: a later
    # refit of the earlier mesh then normalised with the refused mesh's): the refit of the same vertices gives the same grid
    flat = vb.copy()
    flat[:, :3] = flat[0, :3]
    with pytest.raises(dxv.DxvError, match="degenerate or non-finite bound"):
        v.InitDynamic(flat, ib)
    assert np.array_equal(np.asarray(v.stats()["bound"], np.float32).view(np.uint32), s.bound.view(np.uint32))
    v.UpdateVertices(vb)
    v.Voxelize(64)
    assert np.array_equal(v.Grid(), want)
    small = meshes.torus(40, 20)                                               # a much smaller mesh: new buffers, not the old ones
    v.InitDynamic(*small)
    v.Voxelize(64)
    assert np.array_equal(v.Grid(), orc.Scene(*small).voxelize(64))
    v.InitDynamic(vb, ib)
    v.Voxelize(64)
    assert np.array_equal(v.Grid(), want)
    v.close()


# ---------------------------------------------------------------------------------------------
# build stages
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["bunny", "dragon", "turingbowl"])
def test_build_stages_match_host_code(vox, orc, hostcheck, request, name):
    vb, ib, _ = request.getfixturevalue(name)
    vox.InitDynamic(vb, ib)
    st = vox.stats()
    s = orc.Scene(vb, ib)
    assert np.array_equal(np.asarray(st["bound"], np.float32), s.bound)     # A2
    h = hostcheck(vb, ib, s.bound)
    keys = vox.debug(DBG_SORTED_KEYS)
    assert np.array_equal(keys, np.sort(keys)), "radix sort output is not sorted"
    assert np.array_equal(keys, h.keys()), "device Morton keys / sort differ from the host run of the same code"
    nodes, want = vox.debug(DBG_NODES), h.nodes()
    assert np.array_equal(nodes[:, 12:14], want[:, 12:14]), "Karras hierarchy differs"
    assert np.array_equal(nodes[:, :12], want[:, :12]), "refit boxes differ"
    assert np.array_equal(nodes[:, 14:16], want[:, 14:16]), "subtree heights differ"
    assert st["tree_height"] == h.height
    assert np.array_equal(vox.debug(DBG_NODES32), h.nodes32()), "compressed traversal nodes differ"
    assert np.array_equal(vox.debug(DBG_NODES64), h.nodes64()), "wide traversal nodes differ"
    vox.set_option("wide", 0)                             # the wide copy is a section of the scene: built on request
    try:
        vox.InitDynamic(vb, ib)
        with pytest.raises(Exception):
            vox.debug(DBG_NODES64)
        assert np.array_equal(vox.debug(DBG_NODES), nodes)
        vox.set_option("wide", 1)                         # ... asking for it builds the scene again with it
        assert np.array_equal(vox.debug(DBG_NODES64), h.nodes64())
    finally:
        vox.set_option("wide", 2)
    tp = vox.debug(DBG_TRI_POS)
    k = tp[:, 3].view(np.uint32)
    assert np.array_equal(np.sort(k), np.arange(len(k), dtype=np.uint32))
    for i in (0, len(k) // 2, len(k) - 1):                                    # tri prep == oracle
        p, _ = s.tri(int(k[i]))
        assert np.array_equal(tp[i].reshape(3, 4)[:, :3], p)
    tn = vox.debug(DBG_TRI_NRM)
    i = len(k) // 3
    assert np.array_equal(tn[i].reshape(3, 4)[:, :3], vb[ib[3 * int(k[i]):3 * int(k[i]) + 3], 3:])


@pytest.mark.parametrize("mesh", ["dragon", "turingbowl", "tiny"])
def test_refit_variants_identical(dxv, request, mesh):
    """The three box merges of the build -- one atomic pass (0), min/max pyramid with the heights from
    one climb per leaf (1, default), level sweeps (2) -- write the same node words, heights included."""
    if mesh == "tiny":                       # 2, 3 and 5 triangles: the smallest hierarchies
        rng = np.random.default_rng(5)
        cases = []
        for T in (2, 3, 5):
            pos = rng.uniform(-1, 1, (3 * T, 3)).astype(np.float32)
            nrm = np.tile(np.array([[0, 0, 1]], np.float32), (3 * T, 1))
            cases.append((np.hstack([pos, nrm]), np.arange(3 * T, dtype=np.uint32)))
    else:
        vb, ib, _ = request.getfixturevalue(mesh)
        cases = [(vb, ib)]
    vs = [dxv.Voxelizer(0) for _ in range(3)]
    for refit, v in enumerate(vs):
        v.set_option("refit", refit)
    for vb, ib in cases:
        for v in vs:
            v.InitDynamic(vb, ib)
        want = vs[0].debug(DBG_NODES)
        for refit, v in enumerate(vs[1:], 1):
            got = v.debug(DBG_NODES)
            assert np.array_equal(got, want), (mesh, "refit", refit, "differs from refit 0 in", int((got != want).sum()), "words, first rows",
                                               np.argwhere((got != want).any(1))[:4].ravel().tolist())
            assert v.stats()["tree_height"] == vs[0].stats()["tree_height"]
        for k in range(3):                   # rebuilds are deterministic
            vs[1].InitDynamic(vb, ib)
            got = vs[1].debug(DBG_NODES)
            assert np.array_equal(got, want), (mesh, "rebuild", k, "differs in", int((got != want).sum()), "words")
    for v in vs:
        v.close()


# ---------------------------------------------------------------------------------------------
# grids vs oracle / golden vectors
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["bunny", "dragon", "turingbowl"])
@pytest.mark.parametrize("mode", [0, 1])
def test_grid_64_equals_oracle_and_golden(vox, orc, request, grids_json, grids64, name, mode):
    vb, ib, _ = request.getfixturevalue(name)
    tag = "reference" if mode == 0 else "parity"
    want = np.unpackbits(grids64[f"{name}_64_{tag}"])[: 64 ** 3].reshape(64, 64, 64)   # brute-force oracle for mode 0
    # the mirrors' Init (static scene: LBVH + lists, the first launch is the lists kernel through its work queue) and the C-ABI's own
    # rules (InitDynamic: LBVH only, a small first launch walks the tree)
    for init, lists_at_first_launch in ((vox.InitFromArrays, mode == 0), (vox.InitDynamic, False)):
        init(vb, ib)
        vox.Voxelize(64, mode)
        g = vox.Grid()
        assert (vox.stats()["list_entries"] > 0) == lists_at_first_launch
        assert int(g.sum()) == grids_json[f"{name}/64/{tag}"]["solid"]
        assert np.array_equal(g, want), f"{int((g != want).sum())} voxels differ from the golden grid"
        assert np.array_equal(g, orc.Scene(vb, ib).voxelize(64, mode=mode))
        assert vox.CountSolid() == int(g.sum())


@pytest.mark.parametrize("name", ["bunny", "dragon"])
def test_grid_256_golden(vox, request, grids_json, name):
    """config 2: bunny at 256^3 on one MI355X, bit-exact (hash + per-slice popcounts)."""
    vb, ib, _ = request.getfixturevalue(name)
    vox.InitDynamic(vb, ib)
    # reference rule through the tree walk (lists = 0) and through the direction-space lists from the first
    # launch on (lists = 2, the shipped path from a scene's second launch), then the parity rule
    for mode, tag, lists in ((0, "reference", 0), (0, "reference", 2), (1, "parity", 1)):
        vox.set_option("lists", lists)
        vox.Voxelize(256, mode)
        assert (vox.stats()["list_entries"] > 0) == (lists == 2)
        g = vox.Grid()
        want = grids_json[f"{name}/256/{tag}"]
        assert [int(x) for x in g.reshape(256, -1).sum(1)] == want["slices"]
        assert sha(g) == want["sha256"]


def test_texels_equal_oracle(vox, orc, bunny, grids_json):
    vb, ib, _ = bunny
    vox.InitDynamic(vb, ib)
    vox.EnableTexels(True)
    vox.Voxelize(64)
    g, tex = vox.Grid(), vox.Texels()
    og, otex = orc.Scene(vb, ib).voxelize(64, texels=True)
    assert np.array_equal(g, og) and np.array_equal(tex, otex)
    assert sha(tex) == grids_json["bunny/64/texels"]["sha256"]


@pytest.mark.parametrize("gen,args", [("cube", ()), ("tetrahedron", ()), ("uv_sphere", (24, 12)), ("torus", (60, 30)),
                                      ("soup", (3000,))])
def test_synthetic_equals_brute_force_oracle(vox, orc, gen, args):
    vb, ib = getattr(meshes, gen)(*args)
    vox.InitDynamic(vb, ib)
    s = orc.Scene(vb, ib)
    for mode in ((0,) if gen == "soup" else (0, 1)):
        vox.Voxelize(16, mode)
        assert np.array_equal(vox.Grid(), s.voxelize(16, mode=mode, algo=orc.ALGO_BRUTE))


def test_single_triangle_and_duplicates(vox, orc):
    vb = np.zeros((5, 6), np.float32)
    vb[:3, :3] = [[-0.9, -0.9, 0.4], [0.9, -0.9, 0.4], [0.0, 0.9, 0.4]]
    vb[:3, 3:] = [0.57735, 0.57735, 0.57735]
    vb[3, :3], vb[4, :3] = [-1, -1, -1], [1, 1, 1]
    one = np.arange(3, dtype=np.uint32)
    for ib in (one, np.tile(one, 37), np.tile(one, 5000)):
        vox.InitDynamic(vb, ib)
        s = orc.Scene(vb, ib)
        for mode in (0, 1):
            vox.Voxelize(16, mode)
            assert np.array_equal(vox.Grid(), s.voxelize(16, mode=mode))


def test_every_kernel_variant_gives_the_same_grid(vox, orc, dragon):
    vb, ib, _ = dragon
    vox.InitDynamic(vb, ib)
    want = orc.Scene(vb, ib).voxelize(64)
    for brick in range(8):
        for stack, morton, region, queue, subbox in ((0, 1, 9, 1, 1), (32, 0, 0, 0, 0), (64, 1, 3, 1, 0), (0, 0, 20, 0, 1),
                                                     (48, 1, 24, 1, 1)):
            vox.set_option("brick", brick)
            vox.set_option("stack", stack)
            vox.set_option("morton", morton)
            vox.set_option("region", region)
            vox.set_option("queue", queue)
            vox.set_option("subbox", subbox)
            for lists in (0, 2):                                      # the tree walks, and the direction-space lists
                vox.set_option("lists", lists)
                vox.Voxelize(64)
                assert np.array_equal(vox.Grid(), want), (brick, stack, morton, region, queue, subbox, lists)
    vox.set_option("subbox", 1)
    vox.set_option("morton", 1)
    vox.set_option("region", 6)
    vox.set_option("queue", 1)
    # parity mode: row kernel (default) and the per-voxel kernels, whole grids and ragged slabs
    pwant = orc.Scene(vb, ib).voxelize(64, mode=1)
    for rows, queue, brick in ((1, 1, 4), (0, 1, 4), (0, 0, 1), (0, 1, 0)):
        vox.set_option("rows", rows)
        vox.set_option("queue", queue)
        vox.set_option("brick", brick)
        vox.Voxelize(64, 1)
        assert np.array_equal(vox.Grid(), pwant), (rows, queue, brick)
        vox.Voxelize(64, 1, 13, 9)
        assert np.array_equal(vox.Grid(), pwant[13:22]), (rows, queue, brick)
    vox.set_option("rows", 1)
    vox.set_option("queue", 1)
    vox.set_option("brick", 4)
    for n in (2, 6, 30, 66, 130, 258):
        vox.Voxelize(n, 1)
        assert np.array_equal(vox.Grid(), orc.Scene(vb, ib).voxelize(n, mode=1)), n
    for n in (2, 6, 30, 66):                 # grids that do not fill whole bricks
        vox.set_option("brick", 1)
        vox.set_option("stack", 0)
        vox.Voxelize(n)
        assert np.array_equal(vox.Grid(), orc.Scene(vb, ib).voxelize(n)), n


def test_slabs_concatenate_to_the_full_grid(vox, bunny):
    """config 4's scheme on one GPU: 8 Z-slabs looped == the single-pass grid."""
    vb, ib, _ = bunny
    vox.InitDynamic(vb, ib)
    vox.Voxelize(128)
    full = vox.Grid()
    parts = []
    for r in range(8):
        z0, nz = slab_range(128, r, 8)
        vox.Voxelize(128, 0, z0, nz)
        parts.append((z0, vox.Grid()))
    assert np.array_equal(gather_slabs(parts), full)
    vox.Voxelize(128, 0, 100, 28)
    assert np.array_equal(vox.Grid(), full[100:])


def test_interleaved_partition_equals_full_grid(vox, dragon):
    """The load-balanced partition bench.py uses for N > 1: block-cyclic Z blocks, ranks looped here."""
    from dxrvoxelizer_amd.slabs import interleaved_slices, scatter_interleaved
    vb, ib, _ = dragon
    vox.InitDynamic(vb, ib)
    vox.Voxelize(128)
    full = vox.Grid()
    for world, block in ((8, 4), (4, 8), (2, 64), (1, 128)):
        parts = []
        for r in range(world):
            vox.VoxelizeInterleaved(128, r, world, block)
            parts.append((r, vox.Grid()))
            assert np.array_equal(parts[-1][1], full[interleaved_slices(128, r, world, block)])
        assert np.array_equal(scatter_interleaved(parts, 128, world, block), full)
    import dxrvoxelizer_amd
    with pytest.raises(dxrvoxelizer_amd.DxvError):
        vox.VoxelizeInterleaved(128, 0, 3, 8)                 # 128 % 24 != 0


def test_refit_after_vertex_update_equals_rebuild(dxv, orc, bunny):
    """N4: animate the vertices (fixed topology), refit the existing hierarchy; the grid must equal
    the oracle on the deformed mesh normalised by the ORIGINAL bound (the reference's m_bound is set
    once in Init) -- here checked by keeping two bound-defining helper vertices fixed."""
    vb, ib, _ = bunny
    lo, hi = vb[:, :3].min(0), vb[:, :3].max(0)
    c = (lo + hi) / 2
    big = 1.25 * (hi - lo).max() / 2
    pins = np.zeros((2, 6), np.float32)
    pins[0, :3], pins[1, :3] = c - big, c + big              # unreferenced vertices pin the bound
    vb0 = np.concatenate([vb, pins]).astype(np.float32)
    v = dxv.Voxelizer(0)
    v.InitDynamic(vb0, ib)
    h0 = v.stats()["tree_height"]
    for phase in (0.7, 1.9):
        vb1 = vb0.copy()
        y = (vb1[:-2, 1] - lo[1]) / (hi[1] - lo[1])
        vb1[:-2, 0] += np.float32(0.12 * (hi[0] - lo[0])) * np.sin(6.0 * y + phase).astype(np.float32)   # sway
        v.UpdateVertices(vb1)
        assert v.stats()["tree_height"] == h0               # same hierarchy, new boxes
        v.Voxelize(64)
        want = orc.Scene(vb1, ib).voxelize(64)
        assert np.array_equal(v.Grid(), want)
    fresh = dxv.Voxelizer(0)
    fresh.InitDynamic(vb1, ib)                            # full rebuild of the deformed mesh
    fresh.Voxelize(64)
    assert np.array_equal(fresh.Grid(), v.Grid())
    with pytest.raises(dxv.DxvError):
        v.UpdateVertices(vb1[:-1])                           # vertex count must not change
    v.close(), fresh.close()


def test_thin_and_offcentre_scenes_with_subbox_launch(vox, orc):
    """Only the bricks around the scene's root box are launched (the rest is memset): flat, thin,
    off-centre scenes, slabs and the block-cyclic partition must still give the oracle's grids."""
    from dxrvoxelizer_amd.slabs import interleaved_slices
    base_vb, ib = meshes.uv_sphere(24, 12, 1.0)
    for scale, shift in (((1.0, 0.05, 1.0), (0, 0, 0)), ((0.03, 1.0, 0.4), (0, 0, 0)), ((0.3, 0.3, 0.3), (2.0, -1.0, 0.5))):
        vb = base_vb.copy()
        vb[:, :3] = vb[:, :3] * np.float32(scale) + np.float32(shift)
        if any(shift):                                            # an unreferenced far vertex moves the bound centre
            vb = np.concatenate([vb, np.zeros((1, 6), np.float32)])
        vox.InitDynamic(vb, ib)
        s = orc.Scene(vb, ib)
        for mode in (0, 1):
            want = s.voxelize(64, mode=mode)
            for rows in ((1,) if mode == 0 else (0, 1)):
                vox.set_option("rows", rows)
                vox.Voxelize(64, mode)
                assert np.array_equal(vox.Grid(), want), (scale, shift, mode, rows)
                vox.Voxelize(64, mode, 20, 30)
                assert np.array_equal(vox.Grid(), want[20:50])
                vox.VoxelizeInterleaved(64, 1, 4, 4, mode)
                assert np.array_equal(vox.Grid(), want[interleaved_slices(64, 1, 4, 4)])
        vox.set_option("rows", 1)
        vox.EnableTexels(True)
        vox.Voxelize(64)
        assert np.array_equal(vox.Texels(), s.voxelize(64, texels=True)[1])
        vox.EnableTexels(False)


def test_scene_blob_roundtrip_between_contexts(dxv, orc, dragon):
    """What rank 0 broadcasts: export -> (device buffer) -> import into a second context."""
    import torch
    vb, ib, _ = dragon
    a, b = dxv.Voxelizer(0), dxv.Voxelizer(0)
    a.set_option("wide", 0)                               # first a scene without the wide copy
    a.InitDynamic(vb, ib)
    n = a.scene_bytes()
    blob = torch.empty(n, dtype=torch.uint8, device="cuda")
    a.scene_export(blob.data_ptr(), n)
    torch.cuda.synchronize()
    b.scene_import(blob.data_ptr(), n)
    assert b.stats()["tree_height"] == a.stats()["tree_height"]
    a.Voxelize(64), b.Voxelize(64)
    assert np.array_equal(a.Grid(), b.Grid())
    bad = blob.clone()
    bad[:4] = 0
    with pytest.raises(dxv.DxvError):
        b.scene_import(bad.data_ptr(), n)
    # a scene built with the wide copy carries it in the blob; the importer walks it when asked to
    a.set_option("wide", 1)
    n2 = a.scene_bytes()
    assert n2 > n
    blob2 = torch.empty(n2, dtype=torch.uint8, device="cuda")
    a.scene_export(blob2.data_ptr(), n2)
    torch.cuda.synchronize()
    b.scene_import(blob2.data_ptr(), n2)
    b.set_option("wide", 1)
    want = a.Grid()
    a.Voxelize(64), b.Voxelize(64)
    assert np.array_equal(a.Grid(), want) and np.array_equal(b.Grid(), want)
    b.scene_import(blob.data_ptr(), n)                    # narrow blob with the option still on: binary walk
    b.Voxelize(64)
    assert np.array_equal(b.Grid(), want)
    # the candidate lists travel with the blob once they exist: the importer adopts them and builds nothing
    a.build_lists()
    n3 = a.scene_bytes()
    assert n3 > n2
    blob3 = torch.empty(n3, dtype=torch.uint8, device="cuda")
    a.scene_export(blob3.data_ptr(), n3)
    torch.cuda.synchronize()
    b.scene_import(blob3.data_ptr(), n3)
    a.Voxelize(64), a.Voxelize(64), b.Voxelize(64)          # (a's second launch of the scene: lists; b: lists from the first, none built)
    sa, sb = a.stats(), b.stats()
    assert sa["list_entries"] == sb["list_entries"] > 0 and sa["list_res"] == sb["list_res"] and sb["list_ms"] == 0.0
    assert np.array_equal(a.Grid(), want) and np.array_equal(b.Grid(), want)
    assert np.array_equal(a.debug(DBG_LIST_CELLS), b.debug(DBG_LIST_CELLS)) and np.array_equal(a.debug(DBG_LIST_ENTRIES), b.debug(DBG_LIST_ENTRIES))
    # a blob whose header is consistent but whose list payload is not (a texel pointing outside the entries, an entry naming
    # a triangle the scene does not have) is refused instead of being indexed by the kernel
    hdr = blob3[:512].cpu().numpy().view(np.uint64)
    off_cells, off_entries = int(hdr[15]), int(hdr[16])      # SceneHeader: offListCells, offListEntries (dxv_types.h)
    cells_np = blob3[off_cells:off_cells + 16 * 6 * sb["list_res"] ** 2].cpu().numpy().view(np.uint32).reshape(-1, 4)
    k = int(np.flatnonzero(cells_np[:, 1] & 0xffff)[0])     # a non-empty texel
    for off, word, value in ((off_cells + 16 * k, 0, 0x7fffff00), (off_entries, 3, 0x03ffffff)):
        bad = blob3.clone()
        bad[off:off + 16].view(torch.int32)[word] = value
        torch.cuda.synchronize()
        with pytest.raises(dxv.DxvError, match="inconsistent"):
            b.scene_import(bad.data_ptr(), n3)
    b.scene_import(blob3.data_ptr(), n3)                      # (and the good blob again)
    with pytest.raises(dxv.DxvError):
        b.scene_import(blob3.data_ptr(), n3 - 256)         # truncated blob
    b.set_option("listres", 64)                           # an importer that wants another map builds its own
    b.set_option("lists", 2)                              # (at its first launch)
    b.scene_import(blob3.data_ptr(), n3)
    b.Voxelize(64)
    assert b.stats()["list_res"] == 64 and np.array_equal(b.Grid(), want)
    b.set_option("listres", 0)
    # the parity rule's row lists travel the same way once the exporter has built them
    a.build_lists(parity=True)
    n4 = a.scene_bytes()
    assert n4 > n3
    blob4 = torch.empty(n4, dtype=torch.uint8, device="cuda")
    a.scene_export(blob4.data_ptr(), n4)
    torch.cuda.synchronize()
    b.scene_import(blob4.data_ptr(), n4)
    wantp = orc.Scene(vb, ib).voxelize(64, mode=1)
    a.Voxelize(64, dxv.MODE_PARITY), b.Voxelize(64, dxv.MODE_PARITY)       # b's first parity launch: through the imported row lists
    sa, sb = a.stats(), b.stats()
    assert sa["list_entries"] == sb["list_entries"] > 0 and sa["list_res"] == sb["list_res"] and sb["list_ms"] == 0.0
    assert np.array_equal(a.Grid(), wantp) and np.array_equal(b.Grid(), wantp)
    b.Voxelize(64)                                                          # (and the reference rule through the imported direction lists)
    assert b.stats()["list_entries"] > 0 and np.array_equal(b.Grid(), want)
    hdr = blob4[:512].cpu().numpy().view(np.uint64)
    bad = blob4.clone()
    bad[int(hdr[19]):int(hdr[19]) + 4].view(torch.int32)[0] = 0x7ffffff0   # SceneHeader.offPlEntries: the first row entry names no triangle
    torch.cuda.synchronize()
    with pytest.raises(dxv.DxvError, match="inconsistent"):
        b.scene_import(bad.data_ptr(), n4)
    a.close(), b.close()


def test_frames_in_flight_share_one_scene(dxv, orc, bunny, dragon):
    """FrameCount grids per context (Content/Voxelizer.h:24, :110): launches of different frames overlap on their
    own streams, read ONE scene and ONE set of lists, and every frame's grid is the oracle's."""
    vb, ib, _ = bunny
    v = dxv.Voxelizer(0)
    assert v.FrameCount == 3
    v.set_option("lists", 2)
    v.InitDynamic(vb, ib)
    s = orc.Scene(vb, ib)
    jobs = [(64, 0), (96, 1), (128, 0)]
    for rounds in range(3):                                # back to back, no host sync in between
        for f, (N, mode) in enumerate(jobs):
            v.Voxelize(N, mode, sync=False, frameIndex=f)
    v.SyncAll()
    for f, (N, mode) in enumerate(jobs):
        v.SetFrame(f)
        assert np.array_equal(v.Grid(), s.voxelize(N, mode=mode)), f
        assert v.CountSolid() == int(s.voxelize(N, mode=mode).sum())
        st = v.stats()
        assert (st["grid_dim"], st["voxelize_ms"] > 0) == (N, True)
    v.SetFrame(0)
    e0 = v.stats()["list_entries"]
    v.SetFrame(2)
    assert v.stats()["list_entries"] == e0 > 0             # one set of lists serves both frames
    # a scene change with launches still in flight waits for them; afterwards every frame sees the new scene
    for f in range(3):
        v.Voxelize(128, 0, sync=False, frameIndex=f)
    vb2, ib2, _ = dragon
    v.InitDynamic(vb2, ib2)
    want = orc.Scene(vb2, ib2).voxelize(64)
    for f in (2, 0, 1):
        v.Voxelize(64, 0, sync=False, frameIndex=f)
    for f in range(3):
        v.SetFrame(f)
        assert np.array_equal(v.Grid(), want), f           # Grid() finishes the frame's pending launch itself
    # slabs and the block-cyclic partition per frame
    v.Voxelize(64, 0, 16, 20, sync=False, frameIndex=1)
    v.VoxelizeInterleaved(64, 1, 2, 8, sync=False, frameIndex=2)
    v.SetFrame(1)
    assert np.array_equal(v.Grid(), want[16:36])
    v.SetFrame(2)
    from dxrvoxelizer_amd.slabs import interleaved_slices
    assert np.array_equal(v.Grid(), want[interleaved_slices(64, 1, 2, 8)])
    with pytest.raises(dxv.DxvError):
        v.SetFrame(3)
    v.close()


def test_import_over_a_built_context_drops_its_mesh_state(dxv, orc, bunny, dragon):
    """A context that built a SMALL mesh and then imports a bigger scene must not run build / refit / vertex
    updates over the old buffers: they fail cleanly, and the imported scene voxelizes like its source."""
    import torch
    small_vb, small_ib = meshes.tetrahedron()
    vb, ib, _ = dragon
    a, b = dxv.Voxelizer(0), dxv.Voxelizer(0)
    a.InitDynamic(vb, ib)
    b.InitDynamic(small_vb, small_ib)
    n = a.scene_bytes()
    blob = torch.empty(n, dtype=torch.uint8, device="cuda")
    a.scene_export(blob.data_ptr(), n)
    torch.cuda.synchronize()
    b.scene_import(blob.data_ptr(), n)
    for call in (lambda: b._check(b._lib.dxv_build(b._ctx)), lambda: b._check(b._lib.dxv_refit(b._ctx)),
                 lambda: b.UpdateVertices(vb, refit=False), lambda: b.UpdateVertices(small_vb, refit=False)):
        with pytest.raises(dxv.DxvError):
            call()
    a.Voxelize(64), b.Voxelize(64)
    assert np.array_equal(a.Grid(), b.Grid())
    b.InitDynamic(small_vb, small_ib)                  # and the context is still usable for a mesh of its own
    b.Voxelize(16)
    assert np.array_equal(b.Grid(), orc.Scene(small_vb, small_ib).voxelize(16))
    a.close(), b.close()


def test_non_finite_vertices_are_rejected(dxv):
    vb, ib = meshes.cube()
    v = dxv.Voxelizer(0)
    for bad in (np.nan, np.inf, -np.inf):
        w = vb.copy()
        w[5, 1] = bad
        with pytest.raises(dxv.DxvError, match="vertex 5"):
            v.InitDynamic(w, ib)
    v.InitDynamic(vb, ib)
    v.close()


def test_errors_are_loud(dxv, bunny):
    vb, ib, _ = bunny
    v = dxv.Voxelizer(0)
    with pytest.raises(dxv.DxvError):
        v.Voxelize(64)                                    # before Init
    v.InitDynamic(vb, ib)
    for bad in ((63, 0, 63), (64, 60, 8), (64, 0, 0), (4096, 0, 1)):
        with pytest.raises(dxv.DxvError):
            v.Voxelize(bad[0], 0, bad[1], bad[2])
    with pytest.raises(dxv.DxvError):
        v.InitDynamic(vb, np.array([0, 1, 10 ** 6], np.uint32))
    with pytest.raises(dxv.DxvError):
        v.InitDynamic(np.zeros((3, 6), np.float32), np.arange(3, dtype=np.uint32))   # zero extent
    # 2^17+ identical triangles: every ray that hits them descends both children at every level
    tri = np.zeros((5, 6), np.float32)
    tri[:3, :3] = [[-0.9, -0.9, 0.4], [0.9, -0.9, 0.4], [0.0, 0.9, 0.4]]
    tri[:3, 3:] = [0.57735, 0.57735, 0.57735]
    tri[3, :3], tri[4, :3] = [-1, -1, -1], [1, 1, 1]
    deep = np.tile(np.arange(3, dtype=np.uint32), 140000)
    v.set_option("lists", 0)                              # this test is about the tree walks' columns
    v.InitDynamic(tri, deep)
    assert v.stats()["tree_height"] >= 17
    v.Voxelize(16)
    want16 = v.Grid()
    v.Voxelize(128)
    want128 = v.Grid()
    for wide in (1, 0):
        v.set_option("wide", wide)
        v.set_option("stack", 8)                          # forced shallow column: the rays that run out of it
        v.Voxelize(16)                                    # are finished by the deep-stack redo pass
        assert v.stats()["redo_rays"] > 0 and np.array_equal(v.Grid(), want16)
        with pytest.raises(dxv.DxvError) as e:            # ... unless there are too many of them: reported
            v.Voxelize(128)
        assert "stack" in str(e.value)
        v.set_option("stack", 0)                          # adaptive: the column grows instead, then succeeds
        v.set_option("stack0", 8)
        v.Voxelize(128)
        assert v.stats()["stack_entries"] > 8 and np.array_equal(v.Grid(), want128)
        v.set_option("stack0", 20)
    v.set_option("wide", 2)
    one = dxv.Voxelizer(0)
    one.InitDynamic(tri, np.arange(3, dtype=np.uint32))
    one.Voxelize(16)
    assert np.array_equal(want16, one.Grid())             # 140000 coincident copies == one triangle
    one.close()
    v.close()


def test_cpp_voxelizer_mirror(orc, tmp_path):
    """The C++ host mirror (include/dxv_voxelizer.hpp): Init(file) + Voxelize(N) from a C++ program."""
    vb, ib = meshes.uv_sphere(48, 24, 0.8, (0.1, -0.2, 0.05))
    obj = tmp_path / "sphere.obj"
    with open(obj, "w") as fh:                             # positions only: normals are recomputed
        for p in vb[:, :3]:
            fh.write("v %.6f %.6f %.6f\n" % tuple(p))
        for t in ib.reshape(-1, 3) + 1:
            fh.write("f %d %d %d\n" % tuple(t))
    exe = tmp_path / "voxelize_obj"
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "voxelize_obj.cpp"),
                           "-o", str(exe), "-L" + os.path.join(ROOT, "dxrvoxelizer_amd"), "-ldxv",
                           "-Wl,-rpath," + os.path.join(ROOT, "dxrvoxelizer_amd")])
    out = tmp_path / "grid.bin"
    r = subprocess.run([str(exe), str(obj), "64", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    g = np.fromfile(out, np.uint8).reshape(64, 64, 64)
    ovb, oib, _ = orc.obj_load(str(obj))
    want = orc.Scene(ovb, oib).voxelize(64)
    assert np.array_equal(g, want) and int(r.stdout.strip()) == int(want.sum())


def test_cpp_refit_loop_with_overlapped_upload(orc, bunny, tmp_path):
    """The animated-mesh loop from C++ (tests/cpp/refit_loop.cpp over include/dxv_voxelizer.hpp): VoxelizeAsync, UploadVertices of
    the next pose beside the launch, Refit -- the last frame's grid equals the oracle's on that pose, whichever it is."""
    vb, ib, _ = bunny
    lo, hi = vb[:, :3].min(0) - 0.1, vb[:, :3].max(0) + 0.1
    pins = np.zeros((2, 6), np.float32)
    pins[0, :3], pins[1, :3] = lo, hi                        # unreferenced vertices pin the bound of both poses
    a = np.concatenate([vb, pins]).astype(np.float32)
    b = a.copy()
    b[:-2, 0] += np.float32(0.04) * np.sin(9.0 * b[:-2, 1]).astype(np.float32)
    a.tofile(tmp_path / "a.bin"), b.tofile(tmp_path / "b.bin"), np.ascontiguousarray(ib, np.uint32).tofile(tmp_path / "ib.bin")
    exe = tmp_path / "refit_loop"
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "refit_loop.cpp"),
                           "-o", str(exe), "-L" + os.path.join(ROOT, "dxrvoxelizer_amd"), "-ldxv",
                           "-Wl,-rpath," + os.path.join(ROOT, "dxrvoxelizer_amd")])
    want = {"a": orc.Scene(a, ib).voxelize(64), "b": orc.Scene(b, ib).voxelize(64)}
    assert not np.array_equal(want["a"], want["b"])
    for frames in (1, 4, 7):
        out = tmp_path / ("grid%d.bin" % frames)
        r = subprocess.run([str(exe), str(tmp_path / "a.bin"), str(tmp_path / "b.bin"), str(tmp_path / "ib.bin"), "64", str(frames), str(out)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        solid, pose = r.stdout.split()
        assert pose == ("a" if frames % 2 else "b")
        g = np.fromfile(out, np.uint8).reshape(64, 64, 64)
        assert np.array_equal(g, want[pose]) and int(solid) == int(want[pose].sum()), frames


def test_cpp_multi_gpu_host(orc, bunny, grids_json, tmp_path):
    """The multi-GPU host in C++ (include/dxv_multi.hpp): device-set constructor, rank-0 build, ncclBroadcast of the scene blob
    through the RCCL C API, block-cyclic and slab Voxelize, reassembly -- no Python in the path.  Runs with the devices this
    box has (one here: the same code path as N) and must reproduce the oracle's bunny grid at 128^3."""
    vb, ib, _ = bunny
    mesh = tmp_path / "bunny.bin"
    with open(mesh, "wb") as fh:
        fh.write(np.array([len(vb), len(ib) // 3], np.uint32).tobytes())
        fh.write(np.ascontiguousarray(vb, np.float32).tobytes())
        fh.write(np.ascontiguousarray(ib, np.uint32).tobytes())
    rocm = "/opt/rocm"
    exe = tmp_path / "multi_gpu"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"),
                           os.path.join(ROOT, "tests", "cpp", "multi_gpu.cpp"), "-o", str(exe),
                           "-L" + os.path.join(ROOT, "dxrvoxelizer_amd"), "-l:libdxv.so", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lrccl",
                           "-Wl,-rpath," + os.path.join(ROOT, "dxrvoxelizer_amd"), "-Wl,-rpath," + os.path.join(rocm, "lib")])
    out = tmp_path / "grid.bin"
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev_here = 0                                                       # ("all devices of the box" + the eight-share arithmetic on the first one)
    r = subprocess.run([str(exe), str(mesh), "128", str(out), str(ndev_here or 64), "shares8"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    f = r.stdout.strip().splitlines()[-1].split()                       # (RCCL may print its version banner first)
    solid, ndev, blob, bcast_ms, checksum = int(f[0]), int(f[1]), int(f[2]), float(f[3]), int(f[4], 16)
    want = grids_json["bunny/128/reference"]
    assert solid == want["solid"] and ndev >= 1 and blob > 0
    assert bcast_ms > 0.0 and checksum != 0                             # the broadcast was timed and every device's copy summed
    g = np.fromfile(out, np.uint8).reshape(128, 128, 128)
    assert sha(g) == want["sha256"]
    assert np.array_equal(g, orc.Scene(vb, ib).voxelize(128))


# ---------------------------------------------------------------------------------------------
# full size (BASELINE.json: 1 M-triangle mesh at 512^3)
# ---------------------------------------------------------------------------------------------
def test_full_size_properties_torus_1m_512(vox, orc):
    vb, ib = meshes.torus()                                # exactly 1,000,000 triangles
    assert len(ib) // 3 == 1_000_000
    vox.InitDynamic(vb, ib)
    N = 512
    vox.Voxelize(N)
    g = vox.Grid()
    n1 = vox.CountSolid()
    assert n1 == int(g.sum(dtype=np.uint64))               # device count == host count
    h1 = sha(g)
    vox.Voxelize(N)                                        # idempotence
    assert sha(vox.Grid()) == h1
    parts = []                                             # slab concatenation == whole grid
    for r in range(4):
        z0, nz = slab_range(N, r, 4)
        vox.Voxelize(N, 0, z0, nz)
        parts.append((z0, vox.Grid()))
    assert sha(gather_slabs(parts)) == h1
    # spot slices against the oracle (bit-exact), incl. the first, a grazing and a central one
    s = orc.Scene(vb, ib)
    for z in (0, 90, 255, 256, 400):
        assert np.array_equal(g[z], s.voxelize(N, z0=z, nz=1)[0]), z
    # parity mode = exact interior: analytic torus volume, bound half extent 0.9
    vox.Voxelize(N, 1)
    p = vox.Grid()
    vol = 2 * np.pi ** 2 * 0.6 * 0.3 ** 2 / (2 * 0.9) ** 3 * N ** 3
    assert abs(int(p.sum(dtype=np.uint64)) - vol) < 0.002 * vol
    for z in (90, 256):
        assert np.array_equal(p[z], s.voxelize(N, mode=1, z0=z, nz=1)[0]), z
    # the two occupancy rules agree except near grazing exits
    assert (p != g).mean() < 1e-3


def test_largest_grid_2048_indexing(vox, orc, dragon):
    """Maximum size: 2048^3 = 8.6 G voxels, ids beyond 2^32.  The full grid's solid count must equal
    the sum over 8 Z slabs (same voxels, small ids), and spot slices must equal the oracle."""
    vb, ib, _ = dragon
    vox.InitDynamic(vb, ib)
    N = 2048
    s = orc.Scene(vb, ib)
    for mode in (1, 0):
        vox.Voxelize(N, mode)
        total = vox.CountSolid()
        parts = 0
        for r in range(8):
            z0, nz = slab_range(N, r, 8)
            vox.Voxelize(N, mode, z0, nz)
            parts += vox.CountSolid()
            if r == 5:                                   # slices 1280.. : first slice of the slab vs oracle row sample
                g = vox.Grid()[0]
                want = orc.voxelize_slices(s, N, [z0], mode=mode)[0]
                assert np.array_equal(g, want)
        assert total == parts and total > 0


def test_device_resident_full_grid_by_allgather(dxv, dragon, tmp_path):
    """The optional collective: a device-resident full grid via all_gather (world 1 here; the
    permutation from rank-major blocks to Z order is what matters and is checked for W = 1 and, on
    the host, for W = 4 below)."""
    import torch
    import torch.distributed as dist
    from dxrvoxelizer_amd.slabs import allgather_grid, device_grid_tensor, interleaved_slices
    vb, ib, _ = dragon
    v = dxv.Voxelizer(0)
    v.InitDynamic(vb, ib)
    v.Voxelize(64)
    full = v.Grid()
    assert np.array_equal(device_grid_tensor(v, "cuda").cpu().numpy().reshape(64, 64, 64), full)   # zero-copy view
    dist.init_process_group("gloo", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1)
    try:
        v.VoxelizeInterleaved(64, 0, 1, 8)
        g = allgather_grid(v, dist, 64, 1, 8, "cuda")
        assert np.array_equal(g.cpu().numpy(), full)
    finally:
        dist.destroy_process_group()
    # the same permutation for 4 ranks, emulated on the host
    parts = []
    for r in range(4):
        v.VoxelizeInterleaved(64, r, 4, 8)
        parts.append(torch.from_numpy(v.Grid().reshape(-1)))
    out = torch.cat(parts)
    z = out.view(4, 2, 8, 64, 64).permute(1, 0, 2, 3, 4).reshape(64, 64, 64).numpy()
    assert np.array_equal(z, full)
    v.close()


@pytest.mark.parametrize("n,z0,nz", [(64, 0, 64), (2, 0, 2), (6, 1, 3), (10, 0, 7), (256, 100, 9)])
def test_bit_packed_download_equals_packbits(vox, bunny, n, z0, nz):
    """dxv_grid_download_packed: voxel 8j+i in bit i of byte j, any slab size (ragged tails)."""
    vb, ib, _ = bunny
    vox.InitDynamic(vb, ib)
    vox.Voxelize(n, 0, z0, nz)
    g = vox.Grid()
    bits = vox.GridBits()
    assert bits.shape == ((g.size + 7) // 8,)
    assert np.array_equal(bits, np.packbits(g.reshape(-1), bitorder="little"))
    with pytest.raises(Exception):
        vox.GridBits(np.empty(bits.size + 1, np.uint8))


@pytest.mark.parametrize("name,n", [("bunny", 64), ("dragon", 64), ("turingbowl", 64), ("dragon", 256)])
def test_wide_walk_equals_binary_walk(vox, orc, request, name, n):
    """Option wide = 1: the reference rule over the four-box nodes (Node64) -- same grid, same texels,
    and at 64^3 the oracle's grid."""
    vb, ib, _ = request.getfixturevalue(name)
    vox.set_option("wide", 0)
    vox.set_option("lists", 0)
    vox.InitDynamic(vb, ib)
    vox.EnableTexels(True)
    vox.Voxelize(n)
    g0, t0 = vox.Grid(), vox.Texels()
    vox.set_option("wide", 1)
    try:
        vox.Voxelize(n)
        g1, t1 = vox.Grid(), vox.Texels()
        vox.set_option("stack", 8 if n == 64 else 12)     # shallow column: many rays take the redo pass
        vox.Voxelize(n)
        g2 = vox.Grid()
        redo = vox.stats()["redo_rays"]
        vox.set_option("stack", 0)
        vox.set_option("wide", 2)                         # four-box nodes on wave-uniform visits only
        vox.EnableTexels(False)
        vox.Voxelize(n)
        assert np.array_equal(vox.Grid(), g0)
    finally:
        vox.set_option("stack", 0)
        vox.set_option("wide", 2)
        vox.set_option("lists", 1)
        vox.EnableTexels(False)
    assert np.array_equal(g0, g1) and np.array_equal(t0, t1) and np.array_equal(g0, g2)
    assert redo > 0
    if n == 64:
        assert np.array_equal(g0, orc.Scene(vb, ib).voxelize(64))


@pytest.mark.parametrize("name,wide", [("bunny", 2), ("dragon", 2), ("turingbowl", 2), ("dragon", 0)])
def test_parity_row_blocks_equal_single_rows(vox, orc, request, name, wide):
    """k_parity_rows with one row per wave and with 2 x 2 rows per wave (one walk over the union of
    the rows): same grid as the per-voxel kernel and the oracle -- whole grids, odd slabs (the last
    slice is repeated inside a block), offsets, and the block-cyclic partition."""
    vb, ib, _ = request.getfixturevalue(name)
    vox.set_option("wide", wide)                      # 0: the scene has no four-box nodes, the rows walk the binary ones
    vox.InitDynamic(vb, ib)
    s = orc.Scene(vb, ib)
    want = s.voxelize(64, mode=1)
    try:
        for rowblock in (1, 2, 4):
            vox.set_option("rowblock", rowblock)
            vox.Voxelize(64, 1)
            assert vox.stats()["row_block"] == rowblock
            assert np.array_equal(vox.Grid(), want), rowblock
            for z0, nz in ((0, 1), (5, 3), (17, 7), (62, 2), (31, 33)):
                vox.Voxelize(64, 1, z0, nz)
                assert np.array_equal(vox.Grid(), want[z0:z0 + nz]), (rowblock, z0, nz)
            for rank in range(4):
                vox.VoxelizeInterleaved(64, rank, 4, 2, 1)
                zs = np.concatenate([np.arange(b, b + 2) for b in range(rank * 2, 64, 8)])
                assert np.array_equal(vox.Grid(), want[zs]), (rowblock, rank)
            vox.Voxelize(30, 1)                                   # rows that do not fill a wave, tiles that do not divide
            assert np.array_equal(vox.Grid(), s.voxelize(30, mode=1)), rowblock
        vox.set_option("rowblock", 0)
        vox.Voxelize(64, 1)
        assert vox.stats()["row_block"] in (1, 2, 4) and vox.stats()["tri_extent"] > 0
        assert np.array_equal(vox.Grid(), want)
    finally:
        vox.set_option("rowblock", 0)
        vox.set_option("wide", 2)


def test_pyramid_refit_equals_sweep_refit_word_for_word(dxv, bunny):
    """dxv_refit builds the boxes from a min/max pyramid over the Morton-ordered leaves (refit=1);
    the level sweeps (refit=2) and the atomic climb (refit=0) must produce the same node words --
    boxes, links, heights -- for tiny, odd-sized and asset meshes."""
    rng = np.random.default_rng(11)
    vbB, ibB, _ = bunny
    cases = [meshes.soup(n, seed=77 + n, edge=0.2) for n in (1, 2, 3, 5, 17, 1023, 1024, 1025, 3000)] + [(vbB, ibB)]
    for vb, ib in cases:
        vb = np.ascontiguousarray(vb, np.float32)
        moved = vb.copy()
        moved[:, :3] += rng.uniform(-0.01, 0.01, size=(len(vb), 3)).astype(np.float32)
        lo, hi = vb[:, :3].min(0) - 1, vb[:, :3].max(0) + 1
        pins = np.zeros((2, 6), np.float32)
        pins[0, :3], pins[1, :3] = lo, hi                        # unreferenced vertices pin the bound
        vb0, vb1 = np.concatenate([vb, pins]), np.concatenate([moved, pins])
        words = []
        for refit in (1, 2, 0):
            v = dxv.Voxelizer(0)
            v.set_option("refit", refit)
            v.InitDynamic(vb0, ib)
            v.UpdateVertices(vb1)
            words.append((v.debug(DBG_NODES).copy(), v.debug(DBG_NODES32).copy(), v.stats()["tree_height"]))
            v.close()
        for other in words[1:]:
            assert np.array_equal(words[0][0], other[0]) and np.array_equal(words[0][1], other[1]) and words[0][2] == other[2], len(ib) // 3


def test_refit_defers_node_boxes_until_a_walk_needs_them(dxv, orc, bunny):
    """A refit while lists are wanted stops at the min/max pyramid (deferboxes=1, default): the launch through the lists
    needs triangle records and root box only.  The node boxes -- exact, half-float and four-box -- written later, when a
    tree walk or a download asks, are word for word those of a refit that writes them at once, and the walk's grid is
    the oracle's."""
    vb, ib, _ = bunny
    lo, hi = vb[:, :3].min(0) - 0.1, vb[:, :3].max(0) + 0.1
    pins = np.zeros((2, 6), np.float32)
    pins[0, :3], pins[1, :3] = lo, hi
    vb0 = np.concatenate([vb, pins]).astype(np.float32)
    rng = np.random.default_rng(5)
    late, now = dxv.Voxelizer(0), dxv.Voxelizer(0)
    now.set_option("deferboxes", 0)
    for v in (late, now):
        v.set_option("lists", 2)
        v.InitDynamic(vb0, ib)
    for step in range(3):
        vb1 = vb0.copy()
        vb1[:-2, :3] += rng.uniform(-0.004, 0.004, size=(len(vb), 3)).astype(np.float32)
        want = orc.Scene(vb1, ib).voxelize(64)
        for v in (late, now):
            v.UpdateVertices(vb1)
            v.Voxelize(64)                                   # through the lists: no node is read
            assert v.stats()["list_entries"] > 0
            assert np.array_equal(v.Grid(), want), step
        assert late.stats()["refit_ms"] > 0 and np.array_equal(late.stats()["bound"], now.stats()["bound"])
        if step == 1:
            continue                                         # (two refits in a row without a walk in between)
        for what in (DBG_NODES, DBG_NODES32, DBG_NODES64):
            assert np.array_equal(late.debug(what), now.debug(what)), (step, what)
        late.set_option("lists", 0)
        late.Voxelize(64)                                    # the tree walk over the late boxes
        assert np.array_equal(late.Grid(), want)
        late.set_option("lists", 2)
    # a walk straight after a deferred refit (no download in between) finishes the boxes itself
    late.UpdateVertices(vb0)
    late.set_option("lists", 0)
    late.Voxelize(64, 1)                                     # parity rule, tree walk (first parity launch: no row lists yet)
    late.Voxelize(64)
    assert np.array_equal(late.Grid(), orc.Scene(vb0, ib).voxelize(64))
    late.close(), now.close()


def test_vertex_upload_overlaps_a_launch_in_flight(dxv, orc, bunny):
    """dxv_update_vertices does not wait for the frames (launches read the scene, not the vertex buffer): an upload made
    while a launch is in flight changes nothing about that launch's grid, and the following refit + launch give the moved
    mesh's grid.  A device update queued before a host update lands first."""
    import torch
    vb, ib, _ = bunny
    lo, hi = vb[:, :3].min(0) - 0.1, vb[:, :3].max(0) + 0.1
    pins = np.zeros((2, 6), np.float32)
    pins[0, :3], pins[1, :3] = lo, hi
    vb0 = np.concatenate([vb, pins]).astype(np.float32)
    vb1 = vb0.copy()
    vb1[:-2, 1] += np.float32(0.05)
    s0, s1 = orc.Scene(vb0, ib).voxelize(96), orc.Scene(vb1, ib).voxelize(96)
    assert not np.array_equal(s0, s1)
    v = dxv.Voxelizer(0)
    v.set_option("lists", 2)
    v.InitDynamic(vb0, ib)
    for frame in range(4):
        cur, nxt = (vb0, vb1) if frame % 2 == 0 else (vb1, vb0)
        v.Voxelize(96, sync=False)
        v.UpdateVertices(nxt, refit=False)                   # upload beside the launch
        v.Sync()
        assert np.array_equal(v.Grid(), s0 if cur is vb0 else s1), frame
        v.Refit()
    v.Voxelize(96)
    assert np.array_equal(v.Grid(), s0)
    d = torch.from_numpy(vb1).cuda()
    v.UpdateVerticesDevice(d.data_ptr(), len(vb1), refit=False)      # queued on the context's stream ...
    v.UpdateVertices(vb0, refit=False)                               # ... the later host update must win
    v.Refit()
    v.Voxelize(96)
    assert np.array_equal(v.Grid(), s0)
    v.close()


# ---------------------------------------------------------------------------------------------
# direction-space lists (option lists): the reference rule without a tree walk
# ---------------------------------------------------------------------------------------------
def test_lists_of_many_large_footprints_equal_host_lists(dxv, orc, hostcheck):
    """The list build hands footprints out by size (a thread, a wave each, the whole GPU for the first 256 of more than 16,384 texels,
    a wave each for the rest): 300 stacked triangles that each cover 143 x 143 texels of the 512 map's +z face -- more "whole"
    footprints than that list holds -- a few of a whole face (at the grid's centre), and the corner triangles' one-texel ones; the
    lists are the host's word for word, the grid the tree walk's and the oracle's."""
    tris, nrm = [], []
    for k in range(300):
        d = 0.2 + 0.0026 * k
        tris += [(-0.28 * d, -0.28 * d, d), (0.28 * d, -0.28 * d, d), (0.0, 0.28 * d, d)]
        nrm += [(0.0, 0.0, -1.0)] * 3
    for k in range(6):                                                 # through the centre: whole faces
        d = 0.002 * (k + 1)
        tris += [(-0.9, -0.9, -d), (0.9, -0.9, -d), (0.0, 0.9, -d)]
        nrm += [(0.0, 0.0, 1.0)] * 3
    for sgn in (-1.0, 1.0):                                            # the bound: [-1, 1]^3 whatever the rest is
        tris += [(sgn, sgn, sgn), (sgn * 0.999, sgn, sgn), (sgn, sgn * 0.999, sgn)]
        nrm += [(0.0, 0.0, 1.0)] * 3
    vb = np.hstack([np.asarray(tris, np.float32), np.asarray(nrm, np.float32)])
    ib = np.arange(len(vb), dtype=np.uint32)
    s = orc.Scene(vb, ib)
    v = dxv.Voxelizer(0)
    v.set_option("lists", 0)
    v.InitDynamic(vb, ib)
    v.Voxelize(64)
    want = v.Grid().copy()
    assert np.array_equal(want, s.voxelize(64, algo=orc.ALGO_BVH))
    v.set_option("listres", 512)
    v.set_option("lists", 2)
    v.Voxelize(64)
    st = v.stats()
    assert st["list_res"] == 512 and st["list_entries"] > 300 * 16384
    assert np.array_equal(v.Grid(), want)
    cells, entries = hostcheck(vb, ib, s.bound).lists(512)
    assert np.array_equal(v.debug(DBG_LIST_CELLS), cells)
    assert np.array_equal(v.debug(DBG_LIST_ENTRIES), entries)
    v.close()


@pytest.mark.parametrize("name", ["bunny", "dragon", "turingbowl"])
def test_lists_equal_tree_walk_and_host_lists(dxv, orc, hostcheck, request, name):
    """Every grid through the lists equals the tree walk's (and the oracle's); the device-built lists
    equal, word for word, the lists the same footprint code builds on the host."""
    vb, ib, _ = request.getfixturevalue(name)
    v = dxv.Voxelizer(0)
    v.InitDynamic(vb, ib)
    s = orc.Scene(vb, ib)
    for N, R in ((64, 256), (50, 64), (128, 1024)):
        v.set_option("lists", 0)
        v.Voxelize(N)
        want = v.Grid().copy()
        assert v.stats()["list_entries"] == 0
        v.set_option("listres", R)
        v.set_option("lists", 2)
        v.Voxelize(N)
        st = v.stats()
        assert st["list_entries"] > 0 and st["list_res"] == R and st["redo_rays"] == 0
        assert np.array_equal(v.Grid(), want), (name, N, R)
        if N == 64:
            assert np.array_equal(want, s.voxelize(64, algo=orc.ALGO_BVH))
            cells, entries = hostcheck(vb, ib, s.bound).lists(R)
            assert np.array_equal(v.debug(DBG_LIST_CELLS), cells)
            assert np.array_equal(v.debug(DBG_LIST_ENTRIES), entries)
        v.Voxelize(N, 0, N // 4, N // 2)                              # a slab through the lists
        assert np.array_equal(v.Grid(), want[N // 4:N // 4 + N // 2])
    v.EnableTexels(True)
    v.Voxelize(64)
    tl = v.Texels().copy()
    v.set_option("lists", 0)
    v.Voxelize(64)
    assert np.array_equal(v.Texels(), tl)                             # the normal-carrying texel too
    v.close()


def test_lists_follow_the_scene_and_fall_back_over_the_cap(dxv, orc, bunny):
    vb, ib, _ = bunny
    v = dxv.Voxelizer(0)
    v.InitDynamic(vb, ib)
    v.Voxelize(64)                                                    # default: the first launch of a scene walks the tree,
    assert v.stats()["list_entries"] == 0
    a = v.Grid().copy()
    v.Voxelize(64)                                                    # the second one builds and uses the lists
    assert v.stats()["list_entries"] > 0 and np.array_equal(v.Grid(), a)
    v.set_option("lists", 2)                                          # 2: from the first launch on
    moved = np.array(vb, np.float32, copy=True)
    moved[:, 0] *= 0.5                                                # squash the mesh: refit -> the lists are rebuilt
    v.UpdateVertices(moved)
    v.Voxelize(64)
    b = v.Grid().copy()
    assert v.stats()["list_entries"] > 0
    v.set_option("lists", 0)                                          # the refitted tree gives the same grid
    v.Voxelize(64)
    assert np.array_equal(b, v.Grid()) and not np.array_equal(a, b)
    v.set_option("lists", 2)
    # triangles through the grid centre cover whole cube faces: over the cap the tree walk is used
    rng = np.random.default_rng(3)
    T = 400
    pos = rng.uniform(-1, 1, (3 * T, 3)).astype(np.float32)
    big = np.hstack([pos, np.tile(np.array([[0, 0, 1]], np.float32), (3 * T, 1))])
    v.set_option("listres", 4096)
    v.InitDynamic(big, np.arange(3 * T, dtype=np.uint32))
    v.Voxelize(32)
    assert v.stats()["list_entries"] == 0
    assert np.array_equal(v.Grid(), orc.Scene(big, np.arange(3 * T, dtype=np.uint32)).voxelize(32, algo=orc.ALGO_BRUTE))
    v.close()


@pytest.mark.gpu
def test_first_launch_list_policy(dxv, dragon):
    """Option lists=1 (default): a scene's FIRST launch builds the lists only when that pays on this one launch (build_lists'
    estimate after its counting pass) -- a mesh refitted every frame never has a second one.  Same grid every way."""
    from bench import make_mesh
    vb, ib, _ = dragon
    v = dxv.Voxelizer(0)
    v.InitDynamic(vb, ib)
    v.Voxelize(128)                                                    # 2 M voxels: the tree walk is cheaper than any build
    assert v.stats()["list_entries"] == 0
    v.InitDynamic(vb, ib)
    v.Voxelize(416)                                                    # 72 M voxels, 2.7 M entries: the build pays at once
    st = v.stats()
    assert st["list_entries"] > 0 and st["list_res"] == 256
    solid = v.CountSolid()
    v.UpdateVertices(np.ascontiguousarray(vb, np.float32))             # a refit: the next launch is a first launch again
    v.Voxelize(416)
    assert v.stats()["list_entries"] > 0 and v.CountSolid() == solid
    v.set_option("lists", 0)
    v.Voxelize(416)
    assert v.stats()["list_entries"] == 0 and v.CountSolid() == solid
    v.set_option("lists", 1)
    vb9, ib9, _ = make_mesh("dragon9")                                  # a first-launch build keeps the base map (it must pay at once);
    v.InitDynamic(vb9, ib9)                                          # launched again, un-refitted, the scene is static: once, the 512 map
    res, counts = [], []
    for _ in range(3):
        v.Voxelize(416)
        res.append(v.stats()["list_res"]); counts.append(v.CountSolid())
    assert res == [256, 512, 512] and len(set(counts)) == 1, (res, counts)
    v.UpdateVertices(np.ascontiguousarray(vb9, np.float32))             # a mesh that is being animated keeps the base map, however often it is launched
    res = []
    for _ in range(3):
        v.Voxelize(416)
        res.append(v.stats()["list_res"]); counts.append(v.CountSolid())
    assert res == [256, 256, 256] and len(set(counts)) == 1, (res, counts)
    v.close()


@pytest.mark.gpu
def test_update_vertices_from_a_device_buffer(dxv, orc, bunny):
    """dxv_update_vertices_device (a mesh animated on the GPU): same grid as the host-buffer update and as the oracle on the
    moved mesh; wrong counts and null pointers are refused."""
    import torch
    vb, ib, _ = bunny
    moved = np.array(vb, np.float32, copy=True)
    moved[:, 1] *= 0.7
    moved[:, 0] += 0.1 * np.sin(7.0 * moved[:, 2])
    v = dxv.Voxelizer(0)
    v.InitDynamic(vb, ib)
    v.Voxelize(64)
    d = torch.from_numpy(moved).cuda()
    torch.cuda.synchronize()
    v.UpdateVerticesDevice(d.data_ptr(), len(moved))
    v.Voxelize(64)
    a = v.Grid().copy()
    v.InitDynamic(vb, ib)
    v.UpdateVertices(moved)
    v.Voxelize(64)
    assert np.array_equal(a, v.Grid())
    s = orc.Scene(moved, ib)                                           # (the oracle normalises by the moved mesh's own bound: compare
    w = dxv.Voxelizer(0)                                               # with a context built on the moved mesh only when bounds agree)
    w.InitDynamic(moved, ib)
    w.Voxelize(64)
    if np.allclose(s.bound, orc.Scene(vb, ib).bound):
        assert np.array_equal(a, w.Grid())
    with pytest.raises(dxv.DxvError):
        v.UpdateVerticesDevice(d.data_ptr(), len(moved) - 1)
    with pytest.raises(dxv.DxvError):
        v.UpdateVerticesDevice(0, len(moved))
    # one NaN inside an otherwise finite mesh (the root box would not show it): the refit counts it and fails; the next good
    # update recovers
    bad = moved.copy()
    bad[ib[30], 1] = np.nan
    dbad = torch.from_numpy(bad).cuda()
    torch.cuda.synchronize()
    with pytest.raises(dxv.DxvError, match="non-finite"):
        v.UpdateVerticesDevice(dbad.data_ptr(), len(bad))
    v.UpdateVerticesDevice(d.data_ptr(), len(moved))
    v.Voxelize(64)
    assert np.array_equal(a, v.Grid())
    v.close(); w.close()


@pytest.mark.gpu
def test_parity_row_lists_equal_tree_walk_and_oracle(dxv, orc, bunny, dragon):
    """The parity rule through its row lists (`plists`, dirmap.hip: per texel of the (y, z) plane the triangles whose padded box
    reaches it) gives the grids of the row walk over the tree and of the oracle: assets, lattice-snapped adversarial meshes
    (cube-spanning triangles: over the size cap, the tree answers), slabs, block-cyclic ranks, rows longer than a wave's run,
    refits, the policy of option plists = 1."""
    from test_fuzz import lattice_mesh
    v = dxv.Voxelizer(0)
    for vb, ib, _ in (bunny, dragon):
        s = orc.Scene(vb, ib)
        v.InitDynamic(vb, ib)
        for N in (64, 130):
            want = s.voxelize(N, mode=1)
            v.set_option("plists", 2)
            v.Voxelize(N, dxv.MODE_PARITY)
            st = v.stats()
            assert st["list_entries"] > 0 and st["row_block"] == 1 and np.array_equal(v.Grid(), want), N
            v.set_option("plists", 0)
            v.Voxelize(N, dxv.MODE_PARITY)
            assert v.stats()["list_entries"] == 0 and np.array_equal(v.Grid(), want), N
        v.set_option("plists", 2)
        z0, nz = 17, 40
        v.Voxelize(130, dxv.MODE_PARITY, z0, nz)
        assert np.array_equal(v.Grid(), s.voxelize(130, mode=1)[z0:z0 + nz])
        v.VoxelizeInterleaved(128, 1, 4, 2, dxv.MODE_PARITY)
        zs = np.concatenate([np.arange(b, b + 2) for b in range(2, 128, 8)])
        assert np.array_equal(v.Grid(), s.voxelize(128, mode=1)[zs])
    # rows of 1024 voxels: two runs per row
    vb, ib, _ = dragon
    v.InitDynamic(vb, ib)
    v.set_option("plists", 2)
    v.Voxelize(1024, dxv.MODE_PARITY, 500, 24)
    a = v.Grid().copy()
    v.set_option("plists", 0)
    v.Voxelize(1024, dxv.MODE_PARITY, 500, 24)
    assert a.sum() > 0 and np.array_equal(a, v.Grid())
    # policy of the default: a small first launch walks the tree, the second one has the lists; a refit starts over
    v.set_option("plists", 1)
    v.InitDynamic(vb, ib)
    v.Voxelize(64, dxv.MODE_PARITY)
    assert v.stats()["list_entries"] == 0
    v.Voxelize(64, dxv.MODE_PARITY)
    assert v.stats()["list_entries"] > 0
    moved = np.array(vb, np.float32, copy=True)
    moved[:, 2] *= 0.6
    v.UpdateVertices(moved)
    v.Voxelize(64, dxv.MODE_PARITY)
    assert v.stats()["list_entries"] == 0
    b = v.Grid().copy()
    v.Voxelize(64, dxv.MODE_PARITY)
    assert v.stats()["list_entries"] > 0 and np.array_equal(v.Grid(), b)
    # a wall facing the rays covers the whole (y, z) plane: one thread of the fill would walk millions of texels -- the tree stays
    vb, ib, _ = dragon
    lo, hi = vb[:, :3].min(0), vb[:, :3].max(0)
    wall = np.array([[hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [hi[0], hi[1], hi[2]], [hi[0], lo[1], hi[2]]], np.float32)
    wvb = np.concatenate([vb, np.hstack([wall, np.tile(np.array([[1, 0, 0]], np.float32), (4, 1))])]).astype(np.float32)
    n0 = len(vb)
    wib = np.concatenate([ib, np.array([n0, n0 + 1, n0 + 2, n0, n0 + 2, n0 + 3], np.uint32)])
    v.set_option("plists", 2)
    v.InitDynamic(wvb, wib)
    v.Voxelize(64, dxv.MODE_PARITY)
    assert v.stats()["list_entries"] == 0
    assert np.array_equal(v.Grid(), orc.Scene(wvb, wib).voxelize(64, mode=1))
    # adversarial meshes
    rng = np.random.default_rng(515)
    v.set_option("plists", 2)
    v.set_option("plistres", 64)                                         # (on the automatic grids their rectangles are over the cap)
    served = 0
    for n_tris, L, N in ((1, 8, 16), (7, 8, 32), (200, 16, 64), (30, 32, 96), (1500, 32, 64)):
        vb, ib = lattice_mesh(rng, n_tris, L)
        want = orc.Scene(vb, ib).voxelize(N, mode=1, algo=orc.ALGO_BRUTE)
        v.InitDynamic(vb, ib)
        v.Voxelize(N, dxv.MODE_PARITY)
        served += v.stats()["list_entries"] > 0
        assert np.array_equal(v.Grid(), want), (n_tris, L, N)
    assert served >= 4
    v.close()


@pytest.mark.gpu
def test_kept_memset_of_partial_launches(dxv, orc):
    """A partial launch clears the whole grid and writes only the bricks around the scene's root box; the same launch again
    relies on that memset still being there.  Every other writer of the frame's grid must make the next partial launch
    clear it again: the row kernel of the parity rule (an open mesh -- a quad facing the rays -- leaves ones far outside the
    box), a launch with another slab or grid size, the texel image switched on, a caller writing through
    dxv_grid_device_ptr; and each frame keeps its own account."""
    import ctypes as C
    vb = np.array([[0.2, -0.3, -0.3, 1, 0, 0], [0.2, 0.3, -0.3, 1, 0, 0], [0.2, 0.3, 0.3, 1, 0, 0], [0.2, -0.3, 0.3, 1, 0, 0],
                   [-1, -1, -1, 0, 0, 1], [1, 1, 1, 0, 0, 1]], np.float32)          # (two unreferenced vertices pin the bound)
    ib = np.array([0, 1, 2, 0, 2, 3], np.uint32)
    N = 256
    s = orc.Scene(vb, ib)
    ref, par = s.voxelize(N, mode=0), s.voxelize(N, mode=1)
    assert int((par & (1 - ref))[:, :, : N // 2 - 8].sum()) > 100000          # parity ones far to the left of the quad's box
    v = dxv.Voxelizer(0)
    v.InitDynamic(vb, ib)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    for lists, plan in ((2, 1), (2, 0), (0, 0)):                        # lists through the work queue / over the brick box, tree walk
        v.set_option("lists", lists)
        v.set_option("plan", plan)
        for frame in (0, 2):
            v.SetFrame(frame)
            v.Voxelize(N); assert np.array_equal(v.Grid(), ref)
            v.Voxelize(N); assert np.array_equal(v.Grid(), ref)          # (the memset is skipped here)
            v.Voxelize(N, dxv.MODE_PARITY); assert np.array_equal(v.Grid(), par)
            v.Voxelize(N); assert np.array_equal(v.Grid(), ref), "stale parity voxels outside the box"
            v.Voxelize(N, 0, 100, 40); assert np.array_equal(v.Grid(), ref[100:140])
            v.Voxelize(N); assert np.array_equal(v.Grid(), ref)
            v.Voxelize(128); assert np.array_equal(v.Grid(), s.voxelize(128))
            v.Voxelize(N); assert np.array_equal(v.Grid(), ref)
            v.Sync()
            ptr = v.grid_device_ptr()
            assert hip.hipMemset(C.c_void_p(ptr), 1, N ** 3) == 0 and hip.hipDeviceSynchronize() == 0
            v.Voxelize(N); assert np.array_equal(v.Grid(), ref), "the caller's bytes outside the box"
            # a caller who KEEPS the pointer and writes through it later, without fetching it again (the pointer stays
            # valid while the grid is not reallocated): the frame never trusts its own memset again
            v.Voxelize(N); assert np.array_equal(v.Grid(), ref)
            assert hip.hipMemset(C.c_void_p(ptr), 1, N ** 3) == 0 and hip.hipDeviceSynchronize() == 0
            v.Voxelize(N); assert np.array_equal(v.Grid(), ref), "bytes written through a cached pointer survived"
            assert v.grid_device_ptr(writable=False) == ptr
        v.SetFrame(0)
        v.EnableTexels(True)
        v.Voxelize(N)
        assert np.array_equal(v.Grid(), ref) and np.array_equal(v.Texels(), s.voxelize(N, texels=True)[1])
        v.EnableTexels(False)
        v.Voxelize(N); assert np.array_equal(v.Grid(), ref)
    v.close()


def waves_persistent(v):
    """persistent waves of a queue launch on this device (stats of a launch that cannot know its size: plan = 2)"""
    w = v.__class__(0)
    try:
        w.set_option("lists", 2)
        w.set_option("plan", 2)
        import numpy as _np
        tri = _np.array([[-.5, -.5, 0, 0, 0, 1], [.5, -.5, 0, 0, 0, 1], [0, .5, .1, 0, 0, 1]], _np.float32)
        w.InitDynamic(tri, _np.arange(3, dtype=_np.uint32))
        w.Voxelize(32)
        return w.stats()["plan_waves"]
    finally:
        w.close()


@pytest.mark.gpu
def test_work_queue_equals_plain_launch_and_oracle(dxv, orc, bunny, dragon, hostcheck):
    """Lists kernel through the work queue (option plan: only bricks that can hold a live ray are run, decided per brick on
    the device in front of the kernel and taken from per-XCD queues by persistent waves; everything else is the launch's
    memset): same grid as the launch over the brick box and the oracle -- whole grids, grids that are no multiple of the
    brick or the region, slabs, block-cyclic ranks, texels, frames in flight, a refit (new lists: a new queue), every launch
    rebuilt (plan = 2), the caller writing into the grid between two launches -- and the queue's claim checked exhaustively
    (no live ray in a brick that is not queued; no brick queued twice)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    v = dxv.Voxelizer(0)
    v.set_option("lists", 2)
    exposed, grid_cap = False, 0
    for (vb, ib, _), sizes in ((bunny, (64, 50, 100, 256)), (dragon, (66, 128))):
        s = orc.Scene(vb, ib)
        v.InitDynamic(vb, ib)
        for N in sizes:
            want = s.voxelize(N)
            v.set_option("plan", 0)
            v.Voxelize(N)
            assert np.array_equal(v.Grid(), want) and v.stats()["plan_bricks"] == 0
            for plan, dispatch in ((1, 1), (1, 0), (2, 1)):
                v.set_option("plan", plan)
                v.set_option("dispatch", dispatch)
                v.InitDynamic(vb, ib)                                # (a new queue: its first launch does not know its size)
                waves = []
                for again in range(3):
                    v.Voxelize(N)
                    st = v.stats()
                    assert st["plan_bricks"] > 0 and st["plan_waves"] % 8 == 0 and st["plan_waves"] > 0
                    assert np.array_equal(v.Grid(), want), (N, plan, dispatch, again)
                    waves.append(st["plan_waves"])
                # a kept queue whose lengths a sync has read is dealt out by the hardware, one workgroup per slot of the longest of
                # the eight queues and per queue; a first launch, plan = 2 and dispatch = 0 use the persistent waves
                # (a grid whose pointer the caller holds is cleared, and its queue rebuilt, in every launch: until it is reallocated)
                if plan == 1 and dispatch == 1 and not (exposed and N ** 3 <= grid_cap):
                    assert waves[0] == waves_persistent(v) and waves[1] == waves[2] != waves[0] and st["plan_bricks"] <= waves[1] <= 8 * st["plan_bricks"]
                else:
                    assert len(set(waves)) == 1
                chk = v.plan_check()
                assert chk["violations"] == 0 and chk["duplicates"] == 0 and chk["queued_bricks"] == st["plan_bricks"], (N, chk)
                assert chk["live_bricks"] <= chk["queued_bricks"]
            v.set_option("plan", 1)
            exposed, grid_cap = True, max(grid_cap, N ** 3)
            assert hip.hipMemset(C.c_void_p(v.grid_device_ptr()), 1, N ** 3) == 0 and hip.hipDeviceSynchronize() == 0
            v.Voxelize(N)
            assert np.array_equal(v.Grid(), want), "the caller's bytes in bricks that are not queued"
            if N % 16 == 0:
                for z0, nz in ((0, N // 2), (N // 4, 10), (N - 6, 6)):
                    v.Voxelize(N, 0, z0, nz)
                    assert np.array_equal(v.Grid(), want[z0:z0 + nz]), (N, z0, nz)
                    assert v.plan_check()["violations"] == 0
                for world, zb in ((2, 4), (4, 2), (4, 1)):
                    for r in range(world):
                        v.VoxelizeInterleaved(N, r, world, zb)
                        zs = [z for z in range(N) if (z // zb) % world == r]
                        assert np.array_equal(v.Grid(), want[zs]), (N, world, zb, r)
                        assert v.plan_check()["violations"] == 0
    # the device's max-mip of the far radii is the host's, word for word
    vb, ib, _ = bunny
    s = orc.Scene(vb, ib)
    v.InitDynamic(vb, ib)
    v.Voxelize(64)
    R = v.stats()["list_res"]
    h = hostcheck(vb, ib, s.bound)
    h.lists(R)
    assert np.array_equal(v.debug(DBG_LIST_MIP), h.mip(R))
    # texels, frames in flight (every frame owns its queue and its grid's memset)
    N = 128
    want, wtex = s.voxelize(N, texels=True)
    v.EnableTexels(True)
    for _ in range(2):
        v.Voxelize(N)
        assert np.array_equal(v.Grid(), want) and np.array_equal(v.Texels(), wtex)
    v.EnableTexels(False)
    for rounds in range(2):
        for f in range(v.FrameCount):
            v.Voxelize(N, 0, sync=False, frameIndex=f)
    v.SyncAll()
    for f in range(v.FrameCount):
        v.SetFrame(f)
        assert v.stats()["plan_bricks"] > 0 and np.array_equal(v.Grid(), want), f
    v.SetFrame(0)
    # a refit makes new lists: the next launch builds its queue against them
    lo, hi = vb[:, :3].min(0), vb[:, :3].max(0)
    pins = np.zeros((2, 6), np.float32)
    pins[0, :3], pins[1, :3] = (lo + hi) / 2 - 1.25 * (hi - lo).max() / 2, (lo + hi) / 2 + 1.25 * (hi - lo).max() / 2
    vb0 = np.concatenate([vb, pins]).astype(np.float32)                 # (unreferenced vertices pin the bound through the refit)
    v.InitDynamic(vb0, ib)
    v.Voxelize(N); v.Voxelize(N)
    b0 = v.stats()["plan_bricks"]
    assert b0 > 0
    vb1 = vb0.copy()
    vb1[:-2, 0] = (vb1[:-2, 0] - (lo[0] + hi[0]) / 2) * np.float32(0.8) + (lo[0] + hi[0]) / 2
    v.UpdateVertices(vb1)
    w2 = orc.Scene(vb1, ib).voxelize(N)
    for k in range(3):
        v.Voxelize(N)
        assert 0 < v.stats()["plan_bricks"] < b0                         # (the squeezed mesh fills fewer bricks)
        assert np.array_equal(v.Grid(), w2), k
    assert v.plan_check()["violations"] == 0
    g_auto = v.Grid()
    v.set_option("plan", 0)
    v.Voxelize(N)
    assert np.array_equal(v.Grid(), g_auto)
    # launches without the library's events (a caller timing its own loop)
    for plan, events in ((1, 0), (2, 0), (2, 1), (1, 1)):
        v.set_option("plan", plan); v.set_option("events", events)
        for _ in range(2):
            v.Voxelize(N)
        st = v.stats()
        assert st["plan_bricks"] > 0 and (st["voxelize_ms"] > 0) == bool(events)
        if plan == 2 and events:
            assert 0 < st["plan_ms"] < st["voxelize_ms"]
        assert np.array_equal(v.Grid(), g_auto), (plan, events)
    v.close()


@pytest.mark.gpu
def test_deferred_list_verdict_and_queued_refit_frames(dxv, orc, bunny):
    """The dynamic case with one host round trip per frame (XUSGRayTracing.h:13-22): a launch is queued behind the build of its
    lists without waiting for the build's verdict, a refit waits for the frames' launches on the device and runs the lists'
    counting pass behind its own kernels.
    (a) The verdict the host must act on -- a texel with more entries than its 16-bit count holds -- arrives after the launch
        was queued: the lists are withdrawn and the frame is launched again through the tree, synchronously or not.
    (b) Frames of a refit-per-frame loop that nothing waits for give the grids of the same meshes voxelized one by one; a
        launch of ANOTHER frame that is still running when the refit comes reads the old scene to its end."""
    import torch
    # (a) 2 x 70,000 small plates stacked along the x axis through the grid centre: the texels around it hold 70,000 entries each
    K = 70000
    r = np.linspace(0.05, 1.0, K, dtype=np.float32)
    h = np.float32(0.04) * r                                           # (the same few texels for every plate; wide enough for the rays next to the axis)
    one = np.stack([np.stack([r, -h, -h], 1), np.stack([r, h, -h], 1), np.stack([r, np.zeros_like(r), h], 1)], 1).reshape(-1, 3)
    pos = np.concatenate([one, one * np.array([-1, 1, 1], np.float32)])
    vbp = np.ascontiguousarray(np.hstack([pos, np.tile(np.array([[1, 0, 0]], np.float32), (len(pos), 1))]), np.float32)
    ibp = np.arange(len(pos), dtype=np.uint32)
    t = dxv.Voxelizer(0)
    t.set_option("lists", 0)
    t.InitDynamic(vbp, ibp)
    t.Voxelize(64)
    want = t.Grid().copy()
    assert want.any()
    v = dxv.Voxelizer(0)
    v.set_option("lists", 2)
    v.InitDynamic(vbp, ibp)
    v.Voxelize(64)
    st = v.stats()
    assert np.array_equal(v.Grid(), want) and st["list_entries"] == 0 and st["list_ms"] > 0      # (built, then withdrawn)
    v.InitDynamic(vbp, ibp)
    for f in (0, 1, 0):
        v.Voxelize(64, sync=False, frameIndex=f)                       # queued behind the build, and behind each other
    v.SyncAll()
    for f in (0, 1):
        v.SetFrame(f)
        assert np.array_equal(v.Grid(), want) and v.stats()["list_entries"] == 0, f
    v.SetFrame(0)
    t.close()
    # (b)
    vb, ib, _ = bunny
    base = np.ascontiguousarray(vb, np.float32)

    def pose(k):
        m = base.copy()
        m[:, 1] *= np.float32(1.0 - 0.04 * k)
        m[:, 0] += np.float32(0.05) * np.sin(np.float32(5.0 + k) * m[:, 2])
        return m

    v.InitDynamic(base, ib)
    v.Voxelize(128)
    e0 = v.stats()["list_entries"]
    assert e0 > 0
    dev = [torch.from_numpy(pose(k)).cuda() for k in range(5)]
    torch.cuda.synchronize()
    old = v.Grid().copy()
    for k in range(5):
        if k == 3:
            v.Voxelize(256, sync=False, frameIndex=1)                  # still running (pose 2) when the next refit is queued
        v.UpdateVerticesDevice(dev[k].data_ptr(), len(base))
        v.Voxelize(128, sync=False, frameIndex=0)
    v.SyncAll()
    w = dxv.Voxelizer(0)
    w.set_option("lists", 0)
    w.InitDynamic(base, ib)
    w.UpdateVertices(pose(4))
    w.Voxelize(128)
    v.SetFrame(0)
    assert np.array_equal(v.Grid(), w.Grid()) and not np.array_equal(v.Grid(), old)
    st = v.stats()
    assert st["list_entries"] > 0 and st["list_ms"] > 0 and st["plan_bricks"] > 0
    w.UpdateVertices(pose(2))
    w.Voxelize(256)
    v.SetFrame(1)
    assert np.array_equal(v.Grid(), w.Grid())
    # the same loop with the host waiting for every frame, and the lists checked exhaustively against the refitted scene
    v.SetFrame(0)
    for k in (1, 4):
        v.UpdateVerticesDevice(dev[k].data_ptr(), len(base))
        v.Voxelize(128)
    assert np.array_equal(v.Grid(), (w.UpdateVertices(pose(4)), w.Voxelize(128), w.Grid())[2])
    assert v.list_check(128)[1] == 0
    v.close(); w.close()


def test_ray_setup_divisions_equal_ieee_quotients_for_every_voxel_origin(dxv):
    """The ray set-up's divisions (origin of grids whose side is no power of two, cube-map point, direction, 1 / direction, shear) run as a
    scale-free sequence that shares the denominator's refined reciprocal (csrc/dxv_math.h: div_by) -- the canonical rules, the oracle and
    the host-side checks say `/`.  Equal bit for bit for EVERY voxel origin of EVERY even grid size up to 2048 -- 2.2 x 10^12 origins, 14
    words each -- exhaustively on the device (20 s; tools/division_check_all.py leaves the same as a record in the evidence run)."""
    v = dxv.Voxelizer(0)
    try:
        total = 0
        for lo in range(2, 2049, 256):
            hi = min(lo + 254, 2048)
            checked, differing, first = v.division_check(lo, hi)
            assert differing == 0, (lo, hi, differing, first)
            assert checked == sum(n ** 3 for n in range(lo, hi + 1, 2))
            total += checked
        assert total == 8 * (1024 * 1025 // 2) ** 2              # sum of (2k)^3, k = 1 .. 1024
    finally:
        v.close()
