"""Launches through a PREPARED work queue (include/dxv.h: dxv_prepare_launch; the host mirrors' Init with a grid hint): the queue
of a (static scene's lists, grid, partition) is built once, where the reference builds everything its frames trace through
(Content/Voxelizer.cpp:73), and a launch is the grid's clear plus one hardware-dispatched brick kernel
(Content/Voxelizer.cpp:351-369: one DispatchRays per frame).  Checked here: the grids are the oracle's fixtures whatever the grid
held before (every voxel is written in every launch), for every way of clearing, for slabs, block-cyclic shares, grids whose side is
no multiple of 16 and with the texel image on; the queue's own claim (no live ray in an unqueued brick) exhaustively; and that
whatever changes the scene or its lists drops the queue."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD
from test_gpu_configs import check_whole, init, make, sha

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def configs():
    with open(os.path.join(GOLD, "configs.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def dxv(dxvlib):
    import dxrvoxelizer_amd
    return dxrvoxelizer_amd


def poison(v, value=0xAB):
    """Fill the selected frame's grid with garbage through its device pointer: a launch that relied on zeros (or anything else) an
    earlier launch left there would now give a wrong grid."""
    import torch
    from dxrvoxelizer_amd.slabs import device_grid_tensor
    v.Sync()
    device_grid_tensor(v, "cuda").fill_(value)
    torch.cuda.synchronize()


@pytest.mark.parametrize("key", ["torus1m/512/reference", "dragon9/512/reference", "torus1m/256/reference", "bunny16/512/reference",
                                 "soup1m/256/reference"])
def test_prepared_launch_equals_oracle_fixture_whatever_the_grid_held(dxv, configs, key):
    name, N, _ = key.split("/")
    N = int(N)
    v = dxv.Voxelizer(0)
    try:
        init(v, configs, key)
        v.Voxelize(N)                                          # (unprepared: the launch builds its queue itself)
        base = v.stats()
        if not base["list_entries"]:
            v.PrepareLaunch(N)                                 # a scene without lists: nothing to prepare, not an error
            v.Voxelize(N)
            assert v.stats()["plan_prepared"] == 0
            return
        assert base["plan_prepared"] == 0 and base["plan_bricks"] > 0
        v.PrepareLaunch(N)
        assert v.stats()["prepare_ms"] > 0.0
        for clear in (3, 2, 1, 0, 3):
            v.set_option("prepclear", clear)
            poison(v)
            v.Voxelize(N)
            st = v.stats()
            assert st["plan_prepared"] == 1 and st["plan_ms"] == 0.0
            assert st["plan_bricks"] == base["plan_bricks"]    # the same queue the launch would have built
            assert st["plan_waves"] == 8 * ((st["plan_bricks"] + 7) // 8)
            check_whole(v.Grid(), configs[key], f"{key} prepared, prepclear={clear}")
        chk = v.plan_check()
        assert chk["violations"] == 0 and chk["duplicates"] == 0 and chk["queued_bricks"] == st["plan_bricks"], chk
        # the whole wave scanning a lone lane's long list (option coop, default on) looks at the same entries in another order: same grid without
        v.set_option("coop", 0)
        v.Voxelize(N)
        check_whole(v.Grid(), configs[key], f"{key} prepared, every lane scanning alone")
        v.set_option("coop", 1)
        # preparing again finds the partition prepared; option prepared = 0 goes back to a queue per launch
        v.PrepareLaunch(N)
        v.set_option("prepared", 0)
        v.Voxelize(N)
        assert v.stats()["plan_prepared"] == 0 and v.stats()["plan_ms"] > 0.0
        check_whole(v.Grid(), configs[key], f"{key} prepared = 0")
        v.set_option("prepared", 1)
        # all three frames read the one queue, launched back to back on their own streams
        for f in range(v.FrameCount):
            v.SetFrame(f)
            v.Voxelize(N, sync=False)                          # (allocates the frame's grid)
        v.SyncAll()
        for f in range(v.FrameCount):
            v.SetFrame(f)
            poison(v, 0x5A + f)
        for f in range(v.FrameCount):
            v.Voxelize(N, sync=False, frameIndex=f)
        v.SyncAll()
        for f in range(v.FrameCount):
            v.SetFrame(f)
            assert v.stats()["plan_prepared"] == 1
            check_whole(v.Grid(), configs[key], f"{key} prepared, frame {f}")
    finally:
        v.close()


def test_init_with_grid_hint_prepares_and_launches_are_the_same_launch(dxv, configs):
    """The mirrors' Init with the grid it will be launched at (the reference's GRID_SIZE is known to its Init too,
    Content/Voxelizer.cpp:8): the first, second and third Voxelize are the prepared launch, take the same time and give the fixture."""
    key = "torus1m/512/reference"
    vb, ib = make("torus1m")
    v = dxv.Voxelizer(0)
    try:
        v.InitFromArrays(vb, ib, gridDim=512)
        v.Voxelize(512)
        for cycle in range(2):
            v.InitFromArrays(vb, ib, gridDim=512)
            assert v.stats()["prepare_ms"] > 0.0
            ms, shape = [], []
            for call in range(3):
                poison(v, 0x11 * (call + 1))
                v.Voxelize(512)
                s = v.stats()
                ms.append(s["voxelize_ms"]); shape.append((s["plan_prepared"], s["list_res"], s["plan_bricks"], s["plan_waves"]))
                check_whole(v.Grid(), configs[key], f"prepared launch {call + 1} after Init")
            assert len(set(shape)) == 1 and shape[0][0] == 1 and shape[0][1] == 512, shape
            assert max(ms) <= 1.10 * min(ms), ms
        # another grid size of the same scene was not prepared: its launches build their queue
        v.Voxelize(256)
        assert v.stats()["plan_prepared"] == 0
        check_whole(v.Grid(), configs["torus1m/256/reference"], "unprepared 256^3 beside a prepared 512^3")
        v.Voxelize(512)
        assert v.stats()["plan_prepared"] == 1
    finally:
        v.close()


def test_whatever_changes_the_scene_drops_the_prepared_queue(dxv, orc, bunny, dragon):
    vb, ib, _ = bunny
    N = 128
    v = dxv.Voxelizer(0)
    w = dxv.Voxelizer(0)
    try:
        want = orc.Scene(vb, ib).voxelize(N)
        v.InitFromArrays(vb, ib, gridDim=N)
        v.Voxelize(N)
        assert v.stats()["plan_prepared"] == 1 and np.array_equal(v.Grid(), want)
        # a refit (the vertices moved): the old surface's queue must not be used -- nor is the scene static any more
        moved = vb.copy()
        moved[:, :3] = (moved[:, :3] - moved[:, :3].mean(0)) * np.float32(0.8) + moved[:, :3].mean(0)
        v.UpdateVertices(moved)
        v.Voxelize(N)
        assert v.stats()["plan_prepared"] == 0
        fresh = dxv.Voxelizer(0)
        try:
            fresh.InitFromArrays(vb, ib)
            fresh.UpdateVertices(moved)
            fresh.Voxelize(N)
            assert np.array_equal(v.Grid(), fresh.Grid())
        finally:
            fresh.close()
        # a new mesh: dxv_set_mesh + dxv_build drop it; Init without the hint prepares nothing
        dvb, dib, _ = dragon
        v.InitFromArrays(vb, ib, gridDim=N)
        v.InitFromArrays(dvb, dib)
        v.Voxelize(N)
        assert v.stats()["plan_prepared"] == 0
        assert np.array_equal(v.Grid(), orc.Scene(dvb, dib).voxelize(N))
        # lists rebuilt on another map: the queue was probed against the old ones
        v.InitFromArrays(vb, ib, gridDim=N)
        v.set_option("listres", 64)
        v.Voxelize(N)
        assert v.stats()["plan_prepared"] == 0 and v.stats()["list_res"] == 64 and np.array_equal(v.Grid(), want)
        v.PrepareLaunch(N)
        v.Voxelize(N)
        assert v.stats()["plan_prepared"] == 1 and np.array_equal(v.Grid(), want)
        v.set_option("listres", 0)
        # an import: the importing context prepares its own (the queue does not travel; slabs.py / MultiVoxelizer prepare after it)
        import torch
        v.InitFromArrays(vb, ib, gridDim=N)
        n = v.scene_bytes()
        blob = torch.empty(n, dtype=torch.uint8, device="cuda")
        v.scene_export(blob.data_ptr(), n)
        w.InitFromArrays(dvb, dib, gridDim=N)
        w.scene_import(blob.data_ptr(), n)
        w.Voxelize(N)
        assert w.stats()["plan_prepared"] == 0 and np.array_equal(w.Grid(), want)
        w.PrepareLaunch(N)
        w.Voxelize(N)
        assert w.stats()["plan_prepared"] == 1 and np.array_equal(w.Grid(), want)
        # options that a queue was built with (run length, heavy threshold) are part of its key
        w.set_option("planheavy", 9)
        w.Voxelize(N)
        assert w.stats()["plan_prepared"] == 0 and np.array_equal(w.Grid(), want)
        w.set_option("planheavy", 0)
        w.Voxelize(N)
        assert w.stats()["plan_prepared"] == 1
        # tree walks asked for: Init builds no lists and prepares nothing
        w.set_option("lists", 0)
        w.InitFromArrays(vb, ib, gridDim=N)
        w.Voxelize(N)
        st = w.stats()
        assert st["plan_prepared"] == 0 and st["list_entries"] == 0 and np.array_equal(w.Grid(), want)
    finally:
        v.close()
        w.close()


@pytest.mark.parametrize("N", [64, 100, 250, 48, 34])
def test_prepared_slabs_and_shares_of_grids_of_any_even_side(dxv, orc, bunny, N):
    """Slabs whose height is no multiple of 4, grids whose side is no multiple of 16 (the clear is then a kernel of its own) or of 4
    (bricks hang over the grid's end), block-cyclic shares: prepared launches into a poisoned grid against the oracle."""
    from dxrvoxelizer_amd.slabs import scatter_interleaved
    vb, ib, _ = bunny
    want = orc.Scene(vb, ib).voxelize(N)
    v = dxv.Voxelizer(0)
    try:
        v.InitFromArrays(vb, ib, gridDim=N)
        for clear in (3, 2, 1, 0):
            v.set_option("prepclear", clear)
            for z0, nz in ((0, N), (3, 7), (N // 2 - 1, N // 2 + 1), (N - 5, 5)):
                v.PrepareLaunch(N, z0, nz)
                v.Voxelize(N, 0, z0, nz)
                poison(v)
                v.Voxelize(N, 0, z0, nz)
                assert v.stats()["plan_prepared"] == 1
                assert np.array_equal(v.Grid(), want[z0:z0 + nz]), (N, clear, z0, nz)
        if N % 16 == 0:
            W, blk = 4, 4
            parts = []
            for r in range(W):
                v.PrepareLaunchInterleaved(N, r, W, blk)
                v.VoxelizeInterleaved(N, r, W, blk)
                poison(v)
                v.VoxelizeInterleaved(N, r, W, blk)
                assert v.stats()["plan_prepared"] == 1
                parts.append((r, v.Grid().copy()))
            assert np.array_equal(scatter_interleaved(parts, N, W, blk), want)
    finally:
        v.close()


def test_prepared_texel_image_equals_unprepared(dxv, orc, bunny):
    """The reference's R10G10B10A2 texel (hlsl:83-84) through a prepared launch: the clear zeroes the texels of unqueued bricks too."""
    vb, ib, _ = bunny
    N = 128
    v = dxv.Voxelizer(0)
    try:
        v.EnableTexels(True)
        v.InitFromArrays(vb, ib)
        v.Voxelize(N)
        want_t, want_g = v.Texels(), v.Grid()
        assert v.stats()["plan_prepared"] == 0
        og, ot = orc.Scene(vb, ib).voxelize(N, texels=True)
        assert np.array_equal(want_g, og) and np.array_equal(want_t, ot)
        v.PrepareLaunch(N)
        for clear in (3, 2, 1, 0):
            v.set_option("prepclear", clear)
            poison(v)
            v.Voxelize(N)                                       # (the texel image keeps the last launch's values: the launch must overwrite them all)
            assert v.stats()["plan_prepared"] == 1
            assert np.array_equal(v.Grid(), want_g) and np.array_equal(v.Texels(), want_t), clear
    finally:
        v.close()


def test_more_partitions_than_slots(dxv, orc, bunny):
    """Sixteen prepared partitions per context; the seventeenth takes the least recently used slot, whose launches then build
    their own queue again."""
    vb, ib, _ = bunny
    N = 64
    want = orc.Scene(vb, ib).voxelize(N)
    v = dxv.Voxelizer(0)
    try:
        v.InitFromArrays(vb, ib)
        slabs = [(z, 4) for z in range(0, 64, 4)] + [(0, 8)]
        for z0, nz in slabs[:16]:
            v.PrepareLaunch(N, z0, nz)
        for z0, nz in slabs[:16]:
            v.Voxelize(N, 0, z0, nz)
            assert v.stats()["plan_prepared"] == 1 and np.array_equal(v.Grid(), want[z0:z0 + nz])
        v.PrepareLaunch(N, *slabs[16])                          # takes the slot of slabs[0], the least recently launched
        v.Voxelize(N, 0, *slabs[16])
        assert v.stats()["plan_prepared"] == 1 and np.array_equal(v.Grid(), want[0:8])
        v.Voxelize(N, 0, *slabs[0])
        assert v.stats()["plan_prepared"] == 0 and np.array_equal(v.Grid(), want[0:4])
        v.Voxelize(N, 0, *slabs[5])
        assert v.stats()["plan_prepared"] == 1
    finally:
        v.close()


def test_prepare_launch_refuses_what_voxelize_refuses_and_warmup_is_explicit(dxv, dxvlib, bunny):
    """Argument checks of dxv_prepare_launch* mirror dxv_voxelize*'s (same words in the message); dxv_warmup is idempotent per process and
    device and says what it cost (0 ms once the device is warm -- the first context of this process paid, and reported it in its stats)."""
    import ctypes as C
    vb, ib, _ = bunny
    v = dxv.Voxelizer(0)
    try:
        with pytest.raises(dxv.DxvError, match="no scene"):
            v.PrepareLaunch(64)
        v.InitFromArrays(vb, ib)
        for bad in (63, 0, 4096):
            with pytest.raises(dxv.DxvError, match="grid_dim must be even"):
                v.PrepareLaunch(bad)
        with pytest.raises(dxv.DxvError, match="outside the grid"):
            v.PrepareLaunch(64, 60, 8)
        with pytest.raises(dxv.DxvError, match="zblock a power of two"):
            v.PrepareLaunchInterleaved(64, 0, 4, 3)
        with pytest.raises(dxv.DxvError, match="rank < world"):
            v.PrepareLaunchInterleaved(64, 4, 4, 4)
        v.PrepareLaunch(64)                                     # ... and the context is still good
        v.Voxelize(64)
        assert v.stats()["plan_prepared"] == 1
        ms = C.c_float(-1.0)
        assert dxvlib.dxv_warmup(0, C.byref(ms)) == 0 and ms.value == 0.0       # (warm since this process's first dxv_create)
        assert dxvlib.dxv_warmup(0, None) == 0 and dxvlib.dxv_warmup(-1, None) != 0
        assert v.stats()["warmup_ms"] >= 0.0
    finally:
        v.close()


def test_prepared_kernel_breaks_exact_ties_like_the_oracle_and_at_every_occupancy(dxv):
    """The hardware-dispatched kernel keeps nothing of the closest hit but t and the triangle's slot: on an exact tie of t the incumbent's
    index is read from its record, and the barycentrics of an unclassified hit are computed again behind the walk.  A sphere whose every
    triangle is there TWICE -- the copies with negated normals, so the winner of each tie decides the voxel -- in both orders, against the
    oracle; and the same grid at every value of option listedwaves (workgroups per CU held by unused LDS)."""
    from dxrvoxelizer_amd import meshes
    from oracle import orc
    vb, ib = meshes.uv_sphere(48, 24, 0.7, (0.05, -0.03, 0.02))
    V = len(vb)
    flipped = vb.copy()
    flipped[:, 3:6] *= -1.0
    vb2 = np.concatenate([vb, flipped]).astype(np.float32)
    N = 64
    for first in (0, 1):
        ib2 = np.concatenate([ib + (V if first else 0), ib + (0 if first else V)]).astype(np.uint32)
        want = orc.Scene(vb2, ib2).voxelize(N)
        assert (0 < int(want.sum()) < N ** 3) if first == 0 else int(want.sum()) == 0     # (the outward-normal copies first: a solid ball; the negated ones first: nothing)
        v = dxv.Voxelizer(0)
        try:
            v.set_option("lists", 2)
            v.InitFromArrays(vb2, ib2, gridDim=N)
            v.Voxelize(N)                                      # (the frame's grid exists from the first launch on)
            for waves in (0, 32, 28, 20, 8):
                v.set_option("listedwaves", waves)
                poison(v)
                v.Voxelize(N)
                st = v.stats()
                assert st["plan_prepared"] == 1 and st["list_entries"] > 0
                assert np.array_equal(v.Grid(), want), (first, waves)
            with pytest.raises(Exception):
                v.set_option("listedwaves", 7)
            with pytest.raises(Exception):
                v.set_option("listedwaves", 33)
        finally:
            v.close()
    # the opposite-normal copy first or second must matter (else the tie never decided anything)
    a = orc.Scene(vb2, np.concatenate([ib, ib + V]).astype(np.uint32)).voxelize(N)
    b = orc.Scene(vb2, np.concatenate([ib + V, ib]).astype(np.uint32)).voxelize(N)
    assert not np.array_equal(a, b)
