"""The host side's decisions as a table (dxrvoxelizer_amd/csrc/dxv_policy.h compiled for the CPU by tests/hostcheck -- the very
header libdxv.so compiles): WHEN the candidate lists of the reference rule are built and on WHICH map, and whether a launch builds
its work queue, keeps it or has the hardware deal it out.  Every transition of the launch history is walked here without a GPU:
the static scene of the host mirrors (Init builds the lists: every launch is the same launch), the C-ABI's own rules (dxv_build,
then launches), a mesh that is refitted every frame, explicit options, relaunches."""
import ctypes as C

import pytest

NONE, BUILD_IF_IT_PAYS, MOVE_TO_FINE_MAP, BUILD = range(4)
BUILD_AND_PERSISTENT, KEPT_PERSISTENT, KEPT_HARDWARE, PREPARED_HARDWARE = range(4)


class S(C.Structure):
    _fields_ = [("optLists", C.c_int32), ("optListRes", C.c_int32), ("listOpt", C.c_int32), ("listState", C.c_int32), ("listRes", C.c_uint32),
                ("numTris", C.c_uint32), ("launchesOfScene", C.c_uint32), ("refitted", C.c_uint32), ("floorTried", C.c_uint32), ("listEntries", C.c_uint64)]


@pytest.fixture(scope="module")
def pol(hostcheck):
    import os
    from conftest import ROOT
    L = C.CDLL(os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so"))
    L.hc_lists_step.argtypes = [C.POINTER(S), C.c_uint64, C.c_int]
    L.hc_lists_used.argtypes = [C.POINTER(S), C.c_int]
    L.hc_lists_static_fine.argtypes = [C.POINTER(S), C.c_uint32]
    L.hc_lists_base_map.argtypes = [C.c_uint32, C.c_int]
    L.hc_lists_base_map.restype = C.c_uint32
    L.hc_lists_recount_on.argtypes = [C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.c_int]
    L.hc_lists_recount_on.restype = C.c_uint32
    L.hc_lists_over_the_caps.argtypes = [C.c_uint64, C.c_uint32]
    L.hc_lists_pay.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
    L.hc_queue_policy.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64]
    L.hc_queue_policy_prepared.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64]
    return L


def state(**kw):
    d = dict(optLists=1, optListRes=0, listOpt=0, listState=0, listRes=0, numTris=1_000_000, launchesOfScene=0, refitted=0, floorTried=0, listEntries=0)
    d.update(kw)
    return S(**d)


BIG, SMALL = 512 ** 3, 64 ** 3


def test_static_scene_of_the_mirrors_every_launch_is_the_same(pol):
    # Init: dxv_build_lists_for_grid goes straight to the fine map for a scene of 20,000 triangles or more ...
    s = state()
    assert pol.hc_lists_static_fine(C.byref(s), 0) == 1
    assert pol.hc_lists_static_fine(C.byref(state(numTris=5000)), 0) == 0           # (small scenes: the base map)
    assert pol.hc_lists_static_fine(C.byref(state(optListRes=64)), 0) == 0          # (an explicit map wins)
    assert pol.hc_lists_static_fine(C.byref(state(refitted=1)), 0) == 0             # (a mesh that is being refitted: not static)
    # ... and from then on no launch, however many, however large, finds anything to do to the lists: they are used
    for launches in (0, 1, 2, 100):
        for voxels in (SMALL, BIG):
            s = state(listState=1, listRes=512, listEntries=6_400_000, floorTried=1, launchesOfScene=launches)
            assert pol.hc_lists_step(C.byref(s), voxels, 0) == NONE and pol.hc_lists_used(C.byref(s), 0) == 1
    # ... and every launch builds its queue (plan = 2, the default): nothing carried, whatever a sync has read meanwhile
    for kept, lens in ((0, 0), (7, 0), (7, 7)):
        assert pol.hc_queue_policy(2, 1, 0, kept, lens, 1000, 7, BIG) == BUILD_AND_PERSISTENT
    # ... unless Init was given the grid (dxv_prepare_launch): the queue is then the scene's, like the lists, the launch clears its
    # grid and the hardware deals the bricks out -- whatever the frame's last launch was, whoever holds a pointer to the grid
    for plan in (1, 2):
        for kept, lens, exposed in ((0, 0, 0), (7, 0, 0), (7, 7, 1), (9, 9, 0)):
            assert pol.hc_queue_policy_prepared(plan, 1, 1, 1, exposed, kept, lens, 1000, 7, BIG) == PREPARED_HARDWARE
    # option prepared = 0, no prepared queue for this partition, or no queue at all (plan = 0 is not asked): as before
    assert pol.hc_queue_policy_prepared(2, 1, 0, 1, 0, 0, 0, 0, 7, BIG) == BUILD_AND_PERSISTENT
    assert pol.hc_queue_policy_prepared(2, 1, 1, 0, 0, 0, 0, 0, 7, BIG) == BUILD_AND_PERSISTENT
    assert pol.hc_queue_policy_prepared(1, 1, 1, 0, 0, 7, 7, 1000, 7, BIG) == KEPT_HARDWARE


def test_the_c_abi_own_rules_first_second_third_launch(pol):
    # dxv_build, then launches, lists = 1: a SMALL first launch walks the tree (nothing built), the second builds the base map ...
    s = state()
    assert pol.hc_lists_step(C.byref(s), SMALL, 0) == NONE and pol.hc_lists_used(C.byref(s), 0) == 0
    s = state(launchesOfScene=1)
    assert pol.hc_lists_step(C.byref(s), SMALL, 0) == BUILD
    s = state(launchesOfScene=1, listState=1, listRes=256, listEntries=3_300_000)
    assert pol.hc_lists_used(C.byref(s), 0) == 1
    # ... hm: a scene with lists on the 256 map that is launched AGAIN without a refit is static: once, the fine map
    assert pol.hc_lists_step(C.byref(s), SMALL, 0) == MOVE_TO_FINE_MAP
    s = state(launchesOfScene=2, listState=1, listRes=512, listEntries=6_400_000, floorTried=1)
    assert pol.hc_lists_step(C.byref(s), SMALL, 0) == NONE and pol.hc_lists_used(C.byref(s), 0) == 1
    # a LARGE first launch may build at once -- when the build's estimate agrees; declined, it walks the tree
    s = state()
    assert pol.hc_lists_step(C.byref(s), BIG, 0) == BUILD_IF_IT_PAYS
    assert pol.hc_lists_step(C.byref(s), 0, 0) == NONE and pol.hc_lists_used(C.byref(s), 0) == 0      # (asked once per launch: declined)
    s = state(listState=1, listRes=256, listEntries=3_300_000)                                            # (built by that first launch)
    assert pol.hc_lists_step(C.byref(s), 0, 0) == NONE and pol.hc_lists_used(C.byref(s), 0) == 1
    assert pol.hc_lists_pay(BIG, 3_300_000, 256) == 1 and pol.hc_lists_pay(128 ** 3, 3_300_000, 256) == 0
    # lists = 2: from the first launch; lists = 0: never
    assert pol.hc_lists_step(C.byref(state(optLists=2)), SMALL, 0) == BUILD
    s = state(optLists=0, listState=1, listRes=512, launchesOfScene=3)
    assert pol.hc_lists_step(C.byref(s), BIG, 0) == NONE and pol.hc_lists_used(C.byref(s), 0) == 0
    # over the caps (listState = -1): the tree walk, and no build is tried again
    s = state(listState=-1, launchesOfScene=5)
    assert pol.hc_lists_step(C.byref(s), BIG, 0) == NONE and pol.hc_lists_used(C.byref(s), 0) == 0


def test_a_mesh_that_is_refitted_every_frame_keeps_the_base_map(pol):
    # after a refit the scene's launch count starts over; with lists from the first launch (the refit loop: lists = 2, or a scene that
    # had lists) every frame builds on the base map and never moves to the fine one
    s = state(optLists=2, refitted=1)
    assert pol.hc_lists_step(C.byref(s), BIG, 0) == BUILD
    for launches in (1, 2, 5):
        s = state(optLists=2, refitted=1, listState=1, listRes=256, listEntries=3_300_000, launchesOfScene=launches)
        assert pol.hc_lists_step(C.byref(s), BIG, 0) == NONE and pol.hc_lists_used(C.byref(s), 0) == 1
    assert pol.hc_lists_recount_on(256, 3_300_000, 1, 0, 0) == 0                    # a build that must pay on one launch stays on its map
    assert pol.hc_lists_recount_on(256, 3_300_000, 0, 0, 0) == 512                  # a static scene's build moves to the fine map
    assert pol.hc_lists_recount_on(512, 6_400_000, 0, 0, 0) == 0


def test_maps_and_caps(pol):
    assert [pol.hc_lists_base_map(t, 0) for t in (100, 19_999, 20_000, 2_999_999, 3_000_000, 10_000_000)] == [128, 128, 256, 256, 512, 512]
    assert pol.hc_lists_base_map(1_000_000, 64) == 64
    # deep scenes (soups: over 32 entries per texel) step DOWN: 512 -> 256, and to 128 when the 256 map would still hold over 320 M
    assert pol.hc_lists_recount_on(512, 600_000_000, 0, 0, 0) == 256
    assert pol.hc_lists_recount_on(256, 150_000_000, 0, 0, 1) == 0                  # config 5: 150 M entries on the 256 map stay there
    assert pol.hc_lists_recount_on(256, 400_000_000, 0, 0, 1) == 128
    assert pol.hc_lists_recount_on(256, 400_000_000, 0, 0, 0) == 128
    assert pol.hc_lists_recount_on(256, 3_300_000, 0, 0, 1) == 0                    # (never back up to a finer map after a step down)
    # a scene of millions of triangles is counted on a sample first, and the estimate picks the map the full count is made on:
    # config 5 (10 M triangles, ~600 M entries on the 512 map) goes straight to the 256 map, a 4 M-triangle surface stays
    pol.hc_lists_sample_first.argtypes = [C.c_uint32, C.c_int]
    pol.hc_lists_sample_stride.restype = C.c_uint32
    assert pol.hc_lists_sample_first(10_000_000, 0) == 1 and pol.hc_lists_sample_first(1_000_000, 0) == 0
    assert pol.hc_lists_sample_first(10_000_000, 256) == 0                          # (an explicit listres: nothing to pick)
    k = pol.hc_lists_sample_stride()
    assert pol.hc_lists_recount_on(512, (600_000_000 // k) * k, 0, 0, 0) == 256
    assert pol.hc_lists_recount_on(512, (26_000_000 // k) * k, 0, 0, 0) == 0
    # persistent waves by mesh and grid: all the device holds unless a mesh of many small triangles is launched on a grid much
    # coarser than its map
    pol.hc_queue_waves_sevenths.restype = C.c_uint32
    assert [pol.hc_queue_waves_sevenths(1_000_000, 512, n) for n in (64, 128, 256, 384, 512, 1024)] == [4, 4, 5, 7, 7, 7]
    assert [pol.hc_queue_waves_sevenths(100_000, 512, n) for n in (64, 128, 256, 512)] == [7, 7, 7, 7]
    assert pol.hc_queue_waves_sevenths(10_000_000, 256, 256) == 7 and pol.hc_queue_waves_sevenths(10_000_000, 256, 128) == 5
    assert pol.hc_lists_recount_on(256, 3_300_000, 0, 256, 0) == 0                  # an explicit listres: no recount at all
    assert pol.hc_lists_over_the_caps(256 * 1_000_000 + (64 << 20) + 1, 1_000_000) == 1 and pol.hc_lists_over_the_caps(6_400_000, 1_000_000) == 0
    assert pol.hc_lists_over_the_caps(0x80000000, 100_000_000) == 1


def test_options_and_relaunches(pol):
    # another listres than the lists were made with: rebuilt at the next launch; a relaunch (deeper column, withdrawn lists) builds nothing
    s = state(optListRes=1024, listOpt=0, listState=1, listRes=512, launchesOfScene=2, floorTried=1)
    assert pol.hc_lists_step(C.byref(s), BIG, 0) == BUILD
    assert pol.hc_lists_step(C.byref(s), BIG, 1) == NONE and pol.hc_lists_used(C.byref(s), 1) == 0
    s = state(listState=1, listRes=512, launchesOfScene=2, floorTried=1)
    assert pol.hc_lists_step(C.byref(s), BIG, 1) == NONE and pol.hc_lists_used(C.byref(s), 1) == 1
    # a deep scene on a coarse map is left there (no move to the fine map)
    s = state(numTris=10_000_000, listState=1, listRes=256, listEntries=150_000_000, launchesOfScene=1)
    assert pol.hc_lists_step(C.byref(s), BIG, 0) == NONE


def test_work_queue_kept_only_on_request_and_dealt_out_only_when_its_size_is_known(pol):
    sig = 0x1234
    # plan = 1 (opt-in): first launch of a partition builds; the second keeps (persistent waves: the host does not know the size yet);
    # after a sync has read the lengths the hardware deals it out
    assert pol.hc_queue_policy(1, 1, 0, 0, 0, 0, sig, BIG) == BUILD_AND_PERSISTENT
    assert pol.hc_queue_policy(1, 1, 0, sig, 0, 0, sig, BIG) == KEPT_PERSISTENT
    assert pol.hc_queue_policy(1, 1, 0, sig, sig, 455_584, sig, BIG) == KEPT_HARDWARE
    assert pol.hc_queue_policy(1, 1, 0, sig, sig, 0, sig, BIG) == KEPT_PERSISTENT        # (an empty queue: nothing to size a launch with)
    # another partition / lists / buffers (another signature), or a caller holding a writable pointer to the grid: built again
    assert pol.hc_queue_policy(1, 1, 0, sig, sig, 455_584, sig + 2, BIG) == BUILD_AND_PERSISTENT
    assert pol.hc_queue_policy(1, 1, 1, sig, sig, 455_584, sig, BIG) == BUILD_AND_PERSISTENT
    # dispatch = 0: never by the hardware; 2: only for partitions of up to 2^25 voxels
    assert pol.hc_queue_policy(1, 0, 0, sig, sig, 455_584, sig, BIG) == KEPT_PERSISTENT
    assert pol.hc_queue_policy(1, 2, 0, sig, sig, 455_584, sig, BIG) == KEPT_PERSISTENT
    assert pol.hc_queue_policy(1, 2, 0, sig, sig, 60_000, sig, 1 << 25) == KEPT_HARDWARE
    # the lengths of ANOTHER queue (a sync before the partition changed) do not count
    assert pol.hc_queue_policy(1, 1, 0, sig, sig + 2, 455_584, sig, BIG) == KEPT_PERSISTENT


def test_far_map_of_a_scene_without_lists_is_made_at_its_second_launch(pol):
    """The tree walks' brick test reads a far-radius map made from the triangles (dirmap_far, 0.13 ms at 1 M triangles): not for a scene's
    first launch over the brick box -- a mesh refitted every frame would pay it every frame for 1 - 5 % of one launch -- but from the second on."""
    assert pol.hc_far_map_build_now(0, 0) == 0            # first launch of the scene: no map, every brick is walked
    assert pol.hc_far_map_build_now(0, 1) == 1            # launched again unchanged: made now
    assert pol.hc_far_map_build_now(0, 7) == 1
    assert pol.hc_far_map_build_now(1, 1) == 0            # there already
