"""BASELINE.json configurations 3, 4 and 5 (and the metric's own meshes) at their full sizes, through the
C-ABI, against the committed whole-grid fixtures of the CPU oracle (tests/golden/configs.json, made by
oracle/gen_fixtures_configs.py in the build container): SHA-256 over the whole uint8 grid, the solid count
and every slice's popcount -- for both candidate structures of the reference rule (direction-space lists,
`lists=2`, and the LBVH walk, `lists=0`).  Grid indexing: Content/Voxelizer.cpp:366-368, hlsl:64-67."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLD
from dxrvoxelizer_amd import meshes
from dxrvoxelizer_amd.slabs import slab_range

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def configs():
    with open(os.path.join(GOLD, "configs.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def dxv(dxvlib):
    import dxrvoxelizer_amd
    return dxrvoxelizer_amd


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def mesh_sha(vb, ib):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(vb, np.float32).tobytes())
    h.update(np.ascontiguousarray(ib, np.uint32).tobytes())
    return h.hexdigest()


_cache = {}


def make(name):
    """The synthetic meshes of BASELINE.md section 4, regenerated here exactly as the fixture generator made them."""
    if name not in _cache:
        _cache.clear()                                        # one big mesh at a time
        gold = lambda n: np.load(os.path.join(GOLD, "meshes", n + ".npz"))
        if name == "dragon9":
            d = gold("dragon")
            _cache[name] = meshes.trisect(d["vb"], d["ib"])
        elif name == "bunny16":
            b = gold("bunny")
            _cache[name] = meshes.midpoint_subdivide(b["vb"], b["ib"], 2)
        elif name == "torus1m":
            _cache[name] = meshes.torus()
        elif name == "soup10m":
            _cache[name] = meshes.soup()
        elif name == "soup1m":
            _cache[name] = meshes.soup(1_000_000)
        elif name in ("bunny", "dragon"):
            d = gold(name)
            _cache[name] = (d["vb"], d["ib"])
    return _cache[name]


def check_whole(g, want, what):
    N = g.shape[1]
    slices = [int(x) for x in g.reshape(g.shape[0], -1).sum(1, dtype=np.uint64)]
    bad = [z for z in range(len(slices)) if slices[z] != want["slices"][z]]
    assert not bad, f"{what}: {len(bad)} slices differ in their solid count, first z = {bad[0]} (N = {N})"
    assert sum(slices) == want["solid"]
    assert sha(g) == want["sha256"], f"{what}: per-slice counts agree but the grid hash differs"


def init(v, cfg, key):
    name = key.split("/")[0]
    vb, ib = make(name)
    assert mesh_sha(vb, ib) == cfg[key]["mesh_sha256"], f"{name}: the mesh generator no longer produces the fixture's mesh"
    v.InitFromArrays(vb, ib)
    assert v.stats()["num_tris"] == cfg[key]["tris"]


# config 3 (dragon x9, ~900k triangles, 512^3), the metric's meshes (1 M triangles, 256^3 and 512^3) and a 1 M-triangle soup
@pytest.mark.parametrize("key", ["dragon9/512/reference", "torus1m/512/reference", "torus1m/256/reference",
                                 "bunny16/512/reference", "soup1m/256/reference", "torus1m/512/parity"])
def test_config_grid_equals_oracle_fixture(dxv, configs, key):
    name, N, rule = key.split("/")
    N, mode = int(N), (0 if rule == "reference" else 1)
    v = dxv.Voxelizer(0)
    try:
        init(v, configs, key)
        for lists in ((2, 0) if mode == 0 else (1,)):
            v.set_option("lists", lists)
            v.Voxelize(N, mode)
            st = v.stats()
            if mode == 0 and lists == 2 and name != "soup1m":
                assert st["list_entries"] > 0, "the lists were expected to serve this scene"
            if lists == 0:
                assert st["list_entries"] == 0
            check_whole(v.Grid(), configs[key], f"{key} lists={lists}")
            assert v.CountSolid() == configs[key]["solid"]
            if mode == 0 and lists == 2 and st["list_entries"] > 0:
                # (the launch above went through the work queue -- the default: live bricks only, decided on the device in
                # front of the kernel, queue built and grid cleared in every launch.)  Exhaustively: no live ray sits in a brick
                # that was not queued; then the same launch again, the kept queue (plan = 1: the second launch keeps queue and
                # zeros, the third is dealt out by the hardware), and over the brick box
                assert 0 < st["plan_bricks"] <= (N // 4) ** 3 and st["plan_waves"] % 8 == 0 and st["plan_waves"] > 0
                chk = v.plan_check()
                assert chk["violations"] == 0 and chk["duplicates"] == 0 and chk["queued_bricks"] == st["plan_bricks"], chk
                assert chk["live_bricks"] <= chk["queued_bricks"] <= 1.5 * chk["live_bricks"] + 64, chk
                for again in range(2):
                    v.Voxelize(N, mode)
                    assert v.stats()["plan_bricks"] == st["plan_bricks"] and v.stats()["plan_waves"] == st["plan_waves"]
                    check_whole(v.Grid(), configs[key], f"{key} queue rebuilt, launch {again + 2}")
                v.set_option("plan", 1)
                for again in range(3):
                    v.Voxelize(N, mode)
                    assert v.stats()["plan_bricks"] == st["plan_bricks"]
                    check_whole(v.Grid(), configs[key], f"{key} kept queue, launch {again + 1}")
                assert 8 * ((st["plan_bricks"] + 7) // 8) == v.stats()["plan_waves"]      # (one workgroup per item of an XCD's equal share)
                chk = v.plan_check()
                assert chk["violations"] == 0 and chk["duplicates"] == 0 and chk["queued_bricks"] == st["plan_bricks"], chk
                v.set_option("plan", 0)
                v.Voxelize(N, mode)
                assert v.stats()["plan_bricks"] == 0
                check_whole(v.Grid(), configs[key], f"{key} brick box")
                v.set_option("plan", 2)
    finally:
        v.close()


# N1 at the sizes the path is measured at: the reference's per-frame output IS the R10G10B10A2_UNORM texel (float4(Normal, 1) where the
# ray is inside, hlsl:83-84; Content/Voxelizer.cpp:65) -- the whole uint32 image against the oracle's digest (SHA-256 + every slice's
# wrapping sum), through the prepared launch, a launch that builds its queue, the kept queue, the brick box and the tree walk
@pytest.mark.parametrize("key", ["torus1m/512/reference", "dragon9/512/reference", "bunny/256/reference"])
def test_texel_image_at_config_scale_equals_oracle_fixture(dxv, configs, key):
    name, N, _ = key.split("/")
    N = int(N)
    want = configs[key]
    v = dxv.Voxelizer(0)
    try:
        init(v, configs, key)
        v.EnableTexels(True)

        def check(what):
            t = v.Texels()
            sums = [int(x) for x in t.reshape(N, -1).sum(1, dtype=np.uint64)]
            bad = [z for z in range(N) if sums[z] != want["texel_slice_sums"][z]]
            assert not bad, f"{key} texels, {what}: {len(bad)} slices differ, first z = {bad[0]}"
            assert sha(t) == want["texels_sha256"], f"{key} texels, {what}"
            # alpha is what the grid holds (the consumer reads nothing else, PSRayCast.hlsl:108): texel != 0 <=> occupancy 1
            g = v.Grid()
            assert np.array_equal(g != 0, (t >> 30) == 3) and np.array_equal(g == 0, t == 0)
            check_whole(g, want, f"{key} grid beside its texels, {what}")

        v.Voxelize(N)
        assert v.stats()["list_entries"] > 0 and v.stats()["plan_prepared"] == 0
        check("queue built inside the launch")
        v.PrepareLaunch(N)
        v.Voxelize(N)
        assert v.stats()["plan_prepared"] == 1
        check("prepared launch")
        v.set_option("prepared", 0)
        v.set_option("plan", 1)
        for _ in range(3):
            v.Voxelize(N)
        check("kept queue, dealt out by the hardware")
        v.set_option("plan", 0)
        v.Voxelize(N)
        check("brick box")
        v.set_option("plan", 2)
        v.set_option("prepared", 1)
        v.set_option("lists", 0)
        v.Voxelize(N)
        assert v.stats()["list_entries"] == 0
        check("tree walk")
    finally:
        v.close()


def test_config4_dragon9_1024_slabs_and_block_cyclic(dxv, configs):
    """config 4: dragon x9 at 1024^3 -- the whole grid, the 8 contiguous Z slabs of north_star and the block-cyclic
    partition bench.py uses (Z blocks of 8 slices dealt round-robin over 8 ranks), every part against the fixture;
    lists and tree walk.  One GPU plays the 8 ranks in turn (the partition arithmetic is the ranks')."""
    key = "dragon9/1024/reference"
    want = configs[key]
    N, W, blk = 1024, 8, 8
    v = dxv.Voxelizer(0)
    try:
        init(v, configs, key)
        for lists in (2, 0):
            v.set_option("lists", lists)
            v.Voxelize(N)
            check_whole(v.Grid(), want, f"{key} whole grid lists={lists}")
            for r in range(W):
                z0, nz = slab_range(N, r, W)
                v.Voxelize(N, 0, z0, nz)
                g = v.Grid()
                assert [int(x) for x in g.reshape(nz, -1).sum(1, dtype=np.uint64)] == want["slices"][z0:z0 + nz], (lists, r)
                assert sha(g) == want["slabs8_sha256"][r], f"slab {r} lists={lists}"
            for r in range(W):
                v.VoxelizeInterleaved(N, r, W, blk)
                assert sha(v.Grid()) == want["cyclic8x8_sha256"][r], f"block-cyclic rank {r} lists={lists}"
            if lists == 2:
                # (launched at 1024^3 more than once: the lists have moved to the 512 map, a texel stays ~2 voxels wide)
                assert v.stats()["list_res"] == 512
                # (every launch above went through its partition's work queue; one rank's share and one slab checked exhaustively)
                assert v.stats()["plan_bricks"] > 0
                for part in ("cyclic", "slab"):
                    if part == "cyclic":
                        v.VoxelizeInterleaved(N, 5, W, blk)
                    else:
                        v.Voxelize(N, 0, *slab_range(N, 3, W))
                    chk = v.plan_check()
                    assert chk["violations"] == 0 and chk["duplicates"] == 0 and chk["queued_bricks"] == v.stats()["plan_bricks"], (part, chk)
                # ... and over the brick box (no queue)
                v.set_option("plan", 0)
                for r in (0, 3):
                    z0, nz = slab_range(N, r, W)
                    v.Voxelize(N, 0, z0, nz)
                    assert v.stats()["plan_bricks"] == 0 and sha(v.Grid()) == want["slabs8_sha256"][r], f"slab {r} over the brick box"
                v.set_option("plan", 2)
    finally:
        v.close()


def test_config5_soup10m_512(dxv, configs):
    """config 5: the 10 M-triangle soup at 512^3 (build + traversal stress), the product's default candidate
    structure for it and the plain LBVH walk."""
    key = "soup10m/512/reference"
    v = dxv.Voxelizer(0)
    try:
        init(v, configs, key)
        st = v.stats()
        assert st["num_tris"] == 10_000_000 and st["tree_height"] <= 62
        for lists in (2, 0):
            v.set_option("lists", lists)
            v.Voxelize(512)
            check_whole(v.Grid(), configs[key], f"{key} lists={lists}")
        v.set_option("lists", 2)
        v.set_option("plan", 0)
        v.Voxelize(512)
        assert v.stats()["plan_bricks"] == 0
        check_whole(v.Grid(), configs[key], f"{key} brick box")
        for plan in (1, 1, 2):
            v.set_option("plan", plan)
            v.Voxelize(512)
        if v.stats()["list_entries"]:
            assert v.stats()["plan_bricks"] > 0
            chk = v.plan_check()
            assert chk["violations"] == 0 and chk["duplicates"] == 0, chk
            # The lists' superset claim, exhaustively, on a slab of this very scene: 27 texels per triangle make config 5 the case
            # the margin-based culls of the list build (per-texel radial ranges, no entries outside the outline) matter most for --
            # every (ray, triangle) pair the canonical step accepts must be selectable from the ray's texel list
            accepted, violations, first = v.list_check(512, 240, 32)
            assert accepted > 10_000_000 and violations == 0, (accepted, violations, first)
    finally:
        v.close()
        _cache.clear()


# The two exactness claims that rest on margin arguments, checked exhaustively at the scale of the configurations: (i) every
# triangle the canonical step accepts for a ray can be selected from that ray's texel list (dxv_debug_list_check: LBVH walk
# without distance culling against the lists' integer tests, radial cut and early stop), (ii) a triangle's class of the
# normal test agrees with the predicate for every closest hit on it (dxv_debug_class_check: plain LBVH walk + hlsl:137-138).
@pytest.mark.parametrize("name,N,z0,nz", [("dragon9", 512, 0, 512), ("torus1m", 512, 0, 512), ("bunny16", 512, 0, 512),
                                          ("dragon9", 1024, 384, 128), ("soup1m", 256, 0, 256)])
def test_lists_and_classes_checked_exhaustively_at_config_scale(dxv, configs, name, N, z0, nz):
    v = dxv.Voxelizer(0)
    try:
        key = f"{name}/{N}/reference"
        init(v, configs, key)
        v.set_option("lists", 2)
        v.Voxelize(64)                                         # (builds the lists the product would use for this scene)
        if v.stats()["list_entries"]:
            accepted, violations, first = v.list_check(N, z0, nz)
            assert accepted > 0 and violations == 0, (key, accepted, violations, first)
        classified, wrong, hits, first = v.class_check(N, z0, nz)
        assert hits > 0 and wrong == 0, (key, classified, wrong, first)
        if name != "soup1m":
            assert classified > hits // 2, "most hits of a surface mesh land on classified triangles"
        whole = configs[key]
        if (z0, nz) == (0, N):
            assert hits >= whole["solid"]                      # every solid voxel is a hit
    finally:
        v.close()


# The brick test of the launches over the brick box (tree walks: lists = 0, scenes over the lists' caps, dynamic first launches; and
# plan = 0): a brick none of whose rays can reach a triangle is zeroed without a walk -- against a far-radius map made from the
# triangles' own footprints (no lists needed) or the lists' max-mip.  Exhaustively: every ray of every brick the test calls dead is
# walked through the LBVH and must hit nothing; and the grids with the test on and off are the fixture's.
@pytest.mark.parametrize("key", ["torus1m/512/reference", "dragon9/512/reference", "bunny16/512/reference", "soup1m/256/reference", "bunny/256/reference"])
def test_tree_walk_brick_test_never_drops_a_hit(dxv, configs, key):
    name, N, _ = key.split("/")
    N = int(N)
    v = dxv.Voxelizer(0)
    try:
        vb, ib = make(name)
        v.set_option("lists", 0)
        v.InitFromArrays(vb, ib)                               # (lists = 0: Init builds the LBVH only)
        chk = v.far_check(N)
        assert chk["violations"] == 0 and chk["bricks"] == (N // 4) ** 3, chk
        if name != "soup1m":
            assert chk["dead_bricks"] > chk["bricks"] // 4 and chk["rays_walked"] > 0, chk        # (a surface mesh: most of the grid lies beyond it or outside its box)
        ms = {}
        for far in (1, 1, 0, 1):                               # (a scene's first launch over the brick box makes no map: dxv_policy.h, far_map_build_now)
            v.set_option("farmap", far)
            v.Voxelize(N)
            st = v.stats()
            assert st["list_entries"] == 0 and st["plan_bricks"] == 0
            ms[far] = st["voxelize_ms"]
            check_whole(v.Grid(), configs[key], f"{key} tree walk, farmap={far}")
        # the same test against the lists' max-mip (what a brick-box launch of a scene WITH lists reads: plan = 0, lists = 0 set later)
        v.set_option("lists", 2)
        v.Voxelize(N)
        if v.stats()["list_entries"]:
            chk = v.far_check(N, lists_mip=True)
            assert chk["violations"] == 0 and chk["dead_bricks"] > 0, chk
            v.set_option("plan", 0)
            v.Voxelize(N)
            check_whole(v.Grid(), configs[key], f"{key} lists over the brick box with the brick test")
            v.set_option("lists", 0)
            v.Voxelize(N)
            check_whole(v.Grid(), configs[key], f"{key} tree walk with the lists' mip")
    finally:
        v.close()


def test_headline_partition_of_bench_at_8_ranks(dxv, configs):
    """What `bench.py --gpus 8` runs: torus-1M at 512^3, Z blocks of 4 slices dealt round-robin over 8 ranks, every rank's
    share launched repeatedly (through its work queue; from the second launch on with the kept queue and memset).  One GPU plays the ranks in turn; the
    reassembled grid must be the fixture's."""
    from dxrvoxelizer_amd.slabs import scatter_interleaved
    key = "torus1m/512/reference"
    N, W, blk = 512, 8, 4
    v = dxv.Voxelizer(0)
    try:
        init(v, configs, key)
        # the headline's steps (every rank's share prepared after Init, as bench.py does: queue from Init, grid cleared inside the launch),
        # config.unprepared_step's (queue built inside the launch) and config.kept_step's
        for what, plan, prepare in (("prepared", 2, True), ("unprepared", 2, False), ("kept", 1, False)):
            v.set_option("plan", plan)
            v.set_option("prepared", 1 if prepare else 0)
            parts = []
            for r in range(W):
                if prepare:
                    v.PrepareLaunchInterleaved(N, r, W, blk)
                for _ in range(3):
                    v.VoxelizeInterleaved(N, r, W, blk)
                st = v.stats()
                assert st["plan_bricks"] > 0 and st["plan_prepared"] == (1 if prepare else 0)
                if prepare and r in (0, 5):
                    chk = v.plan_check()
                    assert chk["violations"] == 0 and chk["duplicates"] == 0 and chk["queued_bricks"] == st["plan_bricks"], (r, chk)
                parts.append((r, v.Grid().copy()))
            check_whole(scatter_interleaved(parts, N, W, blk), configs[key], f"8 ranks x blocks of 4 slices, {what}")
        v.set_option("prepared", 1)
    finally:
        v.close()


def test_first_second_and_third_voxelize_after_init_are_the_same_launch(dxv, configs):
    """Init leaves everything the launches trace through finished (LBVH + lists on the static scene's map, like the reference's
    Init, Content/Voxelizer.cpp:73) and every launch builds its queue and clears its grid: a scene's first, second and third
    Voxelize put the same kernels into the stream, take the same time (within 10 %: the library's events around the launch) and
    give the fixture's grid."""
    key = "torus1m/512/reference"
    v = dxv.Voxelizer(0)
    try:
        init(v, configs, key)
        st = v.stats()
        assert st["list_entries"] == 0                         # (launch fields: nothing launched yet)
        v.Voxelize(512)                                        # (the process's first launch at this size: allocations, code loading)
        for cycle in range(2):
            init(v, configs, key)                              # a new scene on the context: Init again
            ms, lists = [], []
            for call in range(3):
                v.Voxelize(512)
                s = v.stats()
                ms.append(s["voxelize_ms"]); lists.append((s["list_res"], s["list_entries"], s["plan_bricks"], s["plan_waves"]))
                check_whole(v.Grid(), configs[key], f"launch {call + 1} after Init")
            assert len(set(lists)) == 1 and lists[0][0] == 512 and lists[0][2] > 0, lists       # same map, same lists, same queue, same launch shape
            assert max(ms) <= 1.10 * min(ms), ms
    finally:
        v.close()
