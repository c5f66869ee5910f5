// dxv_frames.hip -- frames, launches and work queues: what a dxv_voxelize* call puts into a frame's stream (launch_now), what
// dxv_sync reads back (sync_frame), and the C-ABI entry points around them.  Whether a launch builds its queue, keeps it or has
// the hardware deal it out is dxv_policy.h's queue_policy.
#include "dxv_ctx.h"

using namespace dxv;
using namespace dxvhost;

namespace dxvhost {

// status words, redo list, events and (frames 1..) the stream of a frame, on its first use
int frame_prepare(dxv_ctx* c, uint32_t i)
{
    Frame& f = c->frames[i];
    if (f.ready) return 0;
    if (i && !f.ownStream) DXV_HIP(c, hipStreamCreateWithFlags(&f.ownStream, hipStreamNonBlocking));
    if (!f.ev0) DXV_HIP(c, hipEventCreate(&f.ev0));
    if (!f.ev1) DXV_HIP(c, hipEventCreate(&f.ev1));
    if (!f.evP0) DXV_HIP(c, hipEventCreate(&f.evP0));
    if (!f.evP1) DXV_HIP(c, hipEventCreate(&f.evP1));
    if (!f.evEnd) DXV_HIP(c, hipEventCreateWithFlags(&f.evEnd, hipEventDisableTiming));
    if (!f.dStatus) DXV_HIP(c, hipMalloc(&f.dStatus, 256));
    if (!f.dRedo) DXV_HIP(c, hipMalloc(&f.dRedo, sizeof(uint64_t) * kRedoCap));
    // on the frame's own stream, and finished before anything reads the words: the streams are non-blocking, a memset on the
    // null stream is not ordered with them (a fresh context whose status words landed on recycled memory could read
    // 0x7ff out of them -- seen twice in some fifty runs of the GPU suite)
    DXV_HIP(c, hipMemsetAsync(f.dStatus, 0, 256, frame_stream(c, i)));
    DXV_HIP(c, hipStreamSynchronize(frame_stream(c, i)));
    f.ready = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Prepared work queues (include/dxv.h: dxv_prepare_launch).  A slot is valid while its epoch is the lists' epoch; everything
// that changes the scene calls drop_prepared as well (a refit keeps the epoch until the lists are rebuilt).
// ---------------------------------------------------------------------------------------------
void drop_prepared(dxv_ctx* c, bool freeMemory)
{
    for (auto& q : c->prepared) {
        q.epoch = 0; q.bricks = 0;
        if (freeMemory) { (void)hipFree(q.dMem); (void)hipFree(q.dLive); q.dMem = q.dLive = nullptr; q.words = q.liveWords = 0; }
    }
}
static uint32_t queue_region_bits(const dxv_ctx* c, uint32_t N, uint32_t nz) { return c->optPlanRegion ? (uint32_t)c->optPlanRegion : plan_region_bits(N, nz); }
static int find_prepared(const dxv_ctx* c, uint32_t N, uint32_t z0, uint32_t nz, uint32_t zBlock, uint32_t zPeriod)
{
    if (!c->listEpoch || c->listState != 1) return -1;
    const uint32_t rb = queue_region_bits(c, N, nz);
    for (uint32_t i = 0; i < dxv_ctx::kPreparedSlots; ++i) {
        const auto& q = c->prepared[i];
        if (q.epoch == c->listEpoch && q.N == N && q.z0 == z0 && q.nz == nz && q.zBlock == zBlock && q.zPeriod == zPeriod && q.regionBits == rb &&
            q.planHeavy == (uint32_t)c->optPlanHeavy && q.dMem)
            return (int)i;
    }
    return -1;
}
// slices: local lz in [0, nzLocal) <-> global z0 + (lz / zBlock) * zPeriod + lz % zBlock (as voxelize_common)
static int prepare_partition(dxv_ctx* c, uint32_t N, uint32_t z0, uint32_t nzLocal, uint32_t zBlock, uint32_t zPeriod)
{
    if (!c->haveScene) return fail(c, "dxv_prepare_launch: no scene (call dxv_build or dxv_scene_import first)");
    DXV_HIP(c, hipSetDevice(c->device));
    c->prepareMs = 0.0f;
    if (!c->optLists || !c->optPlan || c->optBrick != 4 || c->optAblate) return 0;     // (launches of this context do not go through a queue)
    if (settle_lists(c)) return 1;
    if (c->listState == 0 || c->listOpt != c->optListRes) {
        if (dxv_build_lists_for_grid(c, 0)) return 1;
    }
    if (c->listState != 1 || !c->dMip) return 0;                       // a scene without lists (over the caps): tree walks, nothing to prepare
    if (find_prepared(c, N, z0, nzLocal, zBlock, zPeriod) >= 0) return 0;
    if (sync_frames(c)) return 1;                                      // (a slot that is reused may still be read by a launch in flight)
    // a slot: one whose key is this partition's (stale epoch), else a free one, else the least recently used
    int slot = -1;
    for (uint32_t i = 0; i < dxv_ctx::kPreparedSlots && slot < 0; ++i) {
        const auto& q = c->prepared[i];
        if (q.dMem && q.N == N && q.z0 == z0 && q.nz == nzLocal && q.zBlock == zBlock && q.zPeriod == zPeriod) slot = (int)i;
    }
    for (uint32_t i = 0; i < dxv_ctx::kPreparedSlots && slot < 0; ++i)
        if (c->prepared[i].epoch != c->listEpoch) slot = (int)i;
    if (slot < 0) {
        slot = 0;
        for (uint32_t i = 1; i < dxv_ctx::kPreparedSlots; ++i)
            if (c->prepared[i].used < c->prepared[slot].used) slot = (int)i;
    }
    auto& q = c->prepared[slot];
    q.epoch = 0; q.bricks = 0;
    uint32_t cap = 0;
    const size_t words = kQueueHeaderWords + (plan_queue_words(N, nzLocal, &cap) - kQueueSlotsAt), liveWords = plan_live_words(N, nzLocal);
    if (words > q.words) {
        (void)hipFree(q.dMem); q.dMem = nullptr; q.words = 0;
        DXV_HIP(c, hipMalloc(&q.dMem, sizeof(uint32_t) * words));
        q.words = words;
    }
    if (liveWords > q.liveWords) {
        (void)hipFree(q.dLive); q.dLive = nullptr; q.liveWords = 0;
        DXV_HIP(c, hipMalloc(&q.dLive, sizeof(uint32_t) * liveWords));
        q.liveWords = liveWords;
    }
    const hipStream_t s = c->stream;
    VoxelizeParams p{};
    memcpy(p.scene.rootLo, c->hdr.rootLo, 12);
    memcpy(p.scene.rootHi, c->hdr.rootHi, 12);
    p.scene.dmCells = c->dListCells; p.scene.dmEntries = c->dListEntries; p.scene.dmR = c->listRes;
    p.N = N; p.z0 = z0; p.nz = nzLocal; p.zBlock = zBlock; p.zPeriod = zPeriod;
    while ((1u << p.zShift) < p.zBlock) ++p.zShift;
    p.queue = q.dMem; p.queueSlots = q.dMem + kQueueHeaderWords; p.queueCap = cap; p.mip = c->dMip;
    p.planRegionBits = queue_region_bits(c, N, nzLocal); p.planHeavy = (uint32_t)c->optPlanHeavy;
    p.planClear = 0u; p.queueZero = nullptr; p.liveMask = q.dLive;
    DXV_HIP(c, hipEventRecord(c->ev[8], s));
    DXV_HIP(c, hipMemsetAsync(q.dMem, 0, sizeof(uint32_t) * kQueueHeaderWords, s));
    DXV_HIP(c, hipMemsetAsync(q.dLive, 0, sizeof(uint32_t) * liveWords, s));
    DXV_HIP(c, plan_build(p, s));
    DXV_HIP(c, hipMemcpyAsync(c->pin->preparedLens, q.dMem + queue_len_word(0), sizeof(c->pin->preparedLens), hipMemcpyDeviceToHost, s));
    DXV_HIP(c, hipEventRecord(c->ev[9], s));
    DXV_HIP(c, hipStreamSynchronize(s));
    const uint32_t* lens = c->pin->preparedLens;
    for (uint32_t x = 0; x < 8u; ++x) {
        q.lens[8u + x] = lens[queue_heavy_word(x) - queue_len_word(0)];
        q.lens[x] = lens[queue_len_word(x) - queue_len_word(0)] + q.lens[8u + x];
        q.bricks += q.lens[x];
    }
    q.N = N; q.z0 = z0; q.nz = nzLocal; q.zBlock = zBlock; q.zPeriod = zPeriod; q.regionBits = p.planRegionBits; q.planHeavy = p.planHeavy; q.cap = cap;
    q.ms = elapsed(c->ev[8], c->ev[9]);
    q.used = ++c->preparedClock;
    q.epoch = c->listEpoch;
    c->prepareMs = q.ms;
    c->stats.prepare_ms = q.ms;
    return 0;
}

// the far-radius map of the current scene (a scene without lists), on the frame's stream and finished before any other stream can use it
int ensure_far_map(dxv_ctx* c, hipStream_t s)
{
    if (c->farEpoch == c->sceneEpoch && c->dFarMip) return 0;
    // (coarse: a 4^3-voxel brick of a 512^3 grid is a texel of the 128 map wide where the map is finest; the test reads a max-mip level
    // that holds the brick's patch in 2 x 2 cells anyway)
    const uint32_t R = c->hdr.numTris < 20000u ? 64u : 128u;
    if (R > c->farCap) {
        (void)hipFree(c->dFar32); (void)hipFree(c->dFarCells); (void)hipFree(c->dFarMip);
        c->dFar32 = nullptr; c->dFarCells = nullptr; c->dFarMip = nullptr; c->farCap = 0;
        DXV_HIP(c, hipMalloc(&c->dFar32, sizeof(uint32_t) * 6u * R * R));
        DXV_HIP(c, hipMalloc(&c->dFarCells, sizeof(DirCell) * 6u * R * R));
        DXV_HIP(c, hipMalloc(&c->dFarMip, sizeof(uint16_t) * (size_t)dm_mip_buffer_words(R)));
        c->farCap = R;
    }
    c->farEpoch = 0;
    DXV_HIP(c, hipEventRecord(c->ev[8], s));
    DXV_HIP(c, dirmap_far(scene_tripos(c), c->hdr.numTris, R, c->dFar32, c->dFarCells, c->dFarMip, s));
    DXV_HIP(c, hipEventRecord(c->ev[9], s));
    DXV_HIP(c, hipStreamSynchronize(s));
    c->farMs = elapsed(c->ev[8], c->ev[9]);
    c->farR = R;
    c->farEpoch = c->sceneEpoch;
    return 0;
}

// Everything that changes what the frames read (mesh, scene, lists, options that rebuild) first lets every
// frame finish -- including the status check and, if a launch asked for it, the relaunch against the OLD scene.
int sync_frames(dxv_ctx* c)
{
    for (uint32_t i = 0; i < DXV_FRAME_COUNT; ++i)
        if (c->frames[i].ready && sync_frame(c, i)) return 1;
    return settle_lists(c);
}

// Stack policy.  The stack never needs more than treeHeight entries, but rays rarely push more
// than a dozen, and LDS (entries * 4 B * threads) is what limits resident waves.  Launches use 20
// entries; the few rays that run out of them are listed and finished by k_voxelize_redo with a
// 64-entry column right behind the launch.  Only when a launch fills that list does it report
// through the status word, and dxv_sync then re-runs it with the next larger depth (up to the
// always-sufficient one) and keeps that depth for this scene.
// (+3: the postponed-leaf traversal keeps room for one push and two queued leaves)
// The wide walk pushes up to three entries per wide level (two binary levels) and keeps room for
// four more slots: 3 * ceil(h / 2) + 5.  Trees too deep for the largest column use the binary walk.
bool use_wide(const dxv_ctx* c, int mode)
{
    const int need = 3 * (((int)c->hdr.treeHeight + 1) / 2) + 5;
    return mode == DXV_MODE_REFERENCE && c->optWide && c->hdr.hasWide && c->optQueue && need <= 64;
}
int safe_stack(const dxv_ctx* c, int mode)
{
    if (use_wide(c, mode)) return stack_round_up(3 * (((int)c->hdr.treeHeight + 1) / 2) + 5);
    return stack_round_up((int)c->hdr.treeHeight + 3);
}

// relaunch: the same launch again with a deeper column (sync_frame, after a walk reported an overflow) -- possibly on behalf of
// a caller that is about to replace the scene (sync_frames): it builds nothing, it takes the candidate structures that exist.
int launch_now(dxv_ctx* c, uint32_t frame, bool relaunch)
{
    Frame& f = c->frames[frame];
    const hipStream_t fs = frame_stream(c, frame);
    VoxelizeParams p{};
    p.scene.nodes = scene_nodes32(c); p.scene.wide = c->hdr.hasWide ? scene_nodes64(c) : nullptr; p.scene.triPos = scene_tripos(c); p.scene.triNrm = scene_trinrm(c);
    memcpy(p.scene.rootLo, c->hdr.rootLo, 12);
    memcpy(p.scene.rootHi, c->hdr.rootHi, 12);
    p.grid = f.dGrid; p.texels = c->texels ? f.dTexels : nullptr; p.status = f.dStatus;
    p.clearSig = &f.clearSig;
    p.redo = f.dRedo; p.redoCap = kRedoCap; p.redoParity = f.redoParity;
    p.N = f.grid_dim; p.z0 = f.z0; p.nz = f.nz; p.mode = f.lastMode;
    p.zBlock = f.lastZBlock; p.zPeriod = f.lastZPeriod;
    p.zShift = 0;
    while ((1u << p.zShift) < p.zBlock) ++p.zShift;
    p.morton = (uint32_t)c->optMorton;
    p.regionBits = (uint32_t)c->optRegion;
    p.queued = (uint32_t)c->optQueue;
    p.subbox = (uint32_t)c->optSubbox;
    p.wide = use_wide(c, p.mode) ? (uint32_t)c->optWide : 0u;      // 1: four-box nodes, 2: on wave-uniform visits only
    int st = c->optStack ? c->optStack : c->stackNow;
    bool queued = false;
    int prep = -1;                                                      // the context's prepared queue this launch runs (-1: none)
    uint32_t cap = 0;                                                   // words per XCD queue of this partition
    f.list_entries = 0; f.list_res = 0;
    f.usedLists = false;
    // WHEN the lists are built and on WHICH map is dxv_policy.h's lists_step (a pure function of the state below: tests/test_policy.py
    // walks its transitions): a first launch that is large enough may build them at once (lists = 1; the build's own estimate
    // decides after its counting pass), a scene launched AGAIN without a refit in between is static and moves to the fine map, once;
    // otherwise they are built when they are wanted and not there.  (A scene whose Init built them -- the host mirrors' -- meets none
    // of this: its first launch is already the launch every later one is.)
    const uint64_t voxels = (uint64_t)p.N * p.N * p.nz;
    auto lists_state = [&]() {
        ListsState s{};
        s.optLists = c->optLists; s.optListRes = c->optListRes; s.listOpt = c->listOpt; s.listState = c->listState; s.listRes = c->listRes;
        s.listEntries = c->listEntries; s.numTris = c->hdr.numTris; s.launchesOfScene = c->launchesOfScene; s.refitted = c->refitted;
        s.floorTried = c->listFloorTried;
        return s;
    };
    bool wantLists = false;
    if (p.mode == DXV_MODE_REFERENCE) {
        ListsStep step = lists_step(lists_state(), voxels, relaunch);
        if (step == ListsStep::build_if_it_pays) {
            if (sync_frames(c)) return 1;
            if (build_lists(c, fs, voxels, true)) return 1;             // (declined: listState stays 0, this launch walks the tree, the second one builds)
            step = lists_step(lists_state(), 0, relaunch);              // (asked once per launch)
        }
        if (step == ListsStep::move_to_fine_map) {
            if (sync_frames(c)) return 1;
            c->listResFloor = kListsFineMap; c->listFloorTried = true;
            if (build_lists(c, fs)) return 1;
            step = lists_step(lists_state(), 0, relaunch);
        }
        if (step == ListsStep::build && build_lists(c, fs, 0, true)) return 1;      // (this launch queues behind the build; its verdict: sync_frame)
        wantLists = lists_used(lists_state(), relaunch);                // (with the scene's launch count as it stood in front of this launch)
        if (!relaunch) ++c->launchesOfScene;
    }
    if (wantLists) {
        if (c->listState == 1) {
            // lists built on another frame's stream whose end nobody has waited for yet: this stream waits for it on the device
            if (c->listCheckPending && c->listCheckStream != fs) DXV_HIP(c, hipStreamWaitEvent(fs, c->evList[3], 0));
            f.usedLists = true; f.listEpochUsed = c->listEpoch;
            p.lists = 1u;
            p.ablate = (uint32_t)c->optAblate;
            p.scene.dmCells = c->dListCells; p.scene.dmEntries = c->dListEntries; p.scene.dmR = c->listRes; p.scene.dmCoop = (uint32_t)c->optCoop; p.listedWaves = (uint32_t)c->optListedWaves;
            st = 16;                                                // no stack: the column is the queue of selected triangles (8 items of two words)
            if (c->optRegion == 6) p.regionBits = 9u;                  // larger XCD regions suit the lists (-4 %); an explicit option wins
            f.list_entries = c->listEntries; f.list_res = c->listRes;
            if (c->optBrick == 4 && !c->optAblate && c->optPlan && c->dMip) {
                // a queue PREPARED for this very launch (dxv_prepare_launch, Init with a grid hint)?  Then the frame needs none of its own.
                prep = c->optPrepared ? find_prepared(c, p.N, p.z0, p.nz, p.zBlock, p.zPeriod) : -1;
                // the frame's work queue: sized for the partition (worst case: every brick live)
                const size_t words = prep >= 0 ? 0 : plan_queue_words(p.N, p.nz, &cap);
                if (words > f.queueWords) {
                    DXV_HIP(c, hipStreamSynchronize(fs));
                    (void)hipFree(f.dQueue); f.dQueue = nullptr; f.queueWords = 0;
                    const hipError_t qe = hipMalloc(&f.dQueue, sizeof(uint32_t) * words);
                    if (qe == hipSuccess) {
                        f.queueWords = words;
                        DXV_HIP(c, hipMemsetAsync(f.dQueue, 0, sizeof(uint32_t) * kQueueSlotsAt, fs));      // both headers
                        f.queueHdr = 0; f.queueOtherClear = true;
                    }
                    else if (qe == hipErrorOutOfMemory) (void)hipGetLastError();       // no queue: the brick-box launch still works
                    else return fail(c, "work queue: hipMalloc failed: %s", hipGetErrorString(qe));
                    f.clearSig = 0;
                }
                if (prep >= 0) {
                    const auto& q = c->prepared[prep];
                    queued = true; p.queue = q.dMem; p.queueSlots = q.dMem + kQueueHeaderWords; p.queueCap = q.cap;
                    p.mip = c->dMip; p.planRegionBits = q.regionBits; p.planHeavy = q.planHeavy;
                }
                else if (f.dQueue) {
                    queued = true; p.queue = f.dQueue + f.queueHdr * kQueueHeaderWords; p.queueSlots = f.dQueue + kQueueSlotsAt; p.queueCap = cap;
                    p.mip = c->dMip; p.queueWaves = (uint32_t)c->optQueueWaves; p.queueSevenths = queue_waves_sevenths(c->hdr.numTris, c->listRes, p.N); p.queueHeads = (uint32_t)c->optQueueHeads; p.queueMinBricks = (uint32_t)c->optQueueMin;
                    p.planRegionBits = c->optPlanRegion ? (uint32_t)c->optPlanRegion : plan_region_bits(p.N, p.nz);
                    p.planClear = c->optFuse ? 1u : 0u;
                    p.planHeavy = (uint32_t)c->optPlanHeavy;
                }
            }
        }
    }
    if (!queued) { f.plan_bricks = 0; f.plan_waves = 0; f.plan_ms = 0.0f; }
    f.lastQueued = false; f.lastPrepared = -1;
    if (f.ptrExposed) p.clearSig = nullptr;                            // the caller may have written into the grid: clear it every time
    st = stack_for_brick(c->optBrick, st);                             // (shapes other than the shipped one are compiled for three depths)
    f.stack_entries = (uint32_t)st;
    f.lastCanFail = true;
    if (p.mode == DXV_MODE_PARITY && c->optRows && !c->optRowBlock) {
        // parity rule: row lists from the scene's second parity launch on (their build, two passes of atomic additions per
        // entry, costs 2 ms at 1 M triangles -- as much as three launches through the tree at 512^3, five with what the lists
        // save: a mesh refitted every frame stays on the tree); plists = 2: from the first
        // ... and only while triangles are small in voxels: a row's candidates are set up per row, and where a triangle spans
        // many rows the 4 x 4 row blocks of the walk share that work (mean box extent in voxels, lists / walk in ms: torus-1M
        // at 1024^3 1.7: 1.31 / 2.02; dragon x9 2.3: 1.26 / 1.49; dragon at 512^3 3.5: 0.16 / 0.36; bunny 5: 0.20 / 0.27;
        // dragon at 1024^3 7: 1.18 / 0.88; bunny 10: 1.36 / 0.98)
        const bool small = c->hdr.triExtent * 0.5f * (float)p.N <= 6.0f;
        const bool want = c->optPlists && (relaunch ? c->plState == 1 : (c->optPlists == 2 || (small && (c->parityLaunchesOfScene > 0 || c->plState != 0))));
        if (!relaunch) ++c->parityLaunchesOfScene;
        if (want && c->plState == 0) {
            if (sync_frames(c)) return 1;
            if (build_plists(c, fs)) return 1;
        }
        if (want && c->plState == 1) {
            p.scene.plCells = c->dPlCells; p.scene.plEntries = c->dPlEntries; p.scene.plR = c->plRes;
            f.list_entries = c->plEntries; f.list_res = c->plRes;
        }
    }
    if (!p.lists && !p.scene.plCells && ensure_nodes(c, fs)) return 1;  // a tree walk after a refit: its copies of the hierarchy first
    if (p.mode == DXV_MODE_REFERENCE && !queued && c->optFarMap && c->optBrick == 4 && !c->optAblate) {
        // a launch over the brick box (tree walk, or the lists under plan = 0): every workgroup makes the queue's brick test itself --
        // against the lists' max-mip when the scene has (settled) lists, else against the far-radius map of the triangles' own
        // footprints (dirmap_far: 0.13 ms at 1 M triangles), made at the scene's SECOND such launch -- a mesh refitted every frame
        // goes without (dxv_policy.h, far_map_build_now)
        if (c->boxLaunchEpoch != c->sceneEpoch) { c->boxLaunchEpoch = c->sceneEpoch; c->boxLaunchesOfScene = 0; }
        const bool haveFar = c->farEpoch == c->sceneEpoch && c->dFarMip;
        if (c->listState == 1 && c->dMip && !c->listCheckPending) { p.mip = c->dMip; p.mipR = c->listRes; }
        else if (haveFar || far_map_build_now(haveFar, c->boxLaunchesOfScene)) {
            if (ensure_far_map(c, fs)) return 1;
            p.mip = c->dFarMip; p.mipR = c->farR;
        }
        if (!relaunch) ++c->boxLaunchesOfScene;
    }
    if ((p.mode == DXV_MODE_REFERENCE && p.lists) || (p.mode == DXV_MODE_PARITY && c->optRows && p.scene.plCells)) f.lastCanFail = false;
    if (c->optEvents) DXV_HIP(c, hipEventRecord(f.ev0, fs));
    if (p.mode == DXV_MODE_PARITY && c->optRows) {
        // rows whose triangles span several voxels share a walk: 4 x 4 rows per wave above 1.5 voxels of
        // mean triangle extent, 2 x 2 above 1.2 -- as long as the launch still has enough waves to fill
        // the GPU twice (blocks of a small grid or a thin slab leave it idle).  Measured crossovers:
        // profiles/r01/final/rowblock.jsonl; voxel-sized triangles are 1.2-2x slower in blocks, 4-7
        // voxel ones 3-5x faster.
        const float voxels = c->hdr.triExtent * 0.5f * (float)p.N;
        const uint64_t nseg = (p.N + 511u) / 512u;
        auto waves = [&](uint32_t rb) { return (uint64_t)((p.N + rb - 1u) / rb) * ((p.nz + rb - 1u) / rb) * nseg; };
        int rowBlock = 1;
        if (voxels > 1.5f && waves(4) >= 12288u) rowBlock = 4;
        else if (voxels > 1.2f && waves(2) >= 12288u) rowBlock = 2;
        if (c->optRowBlock) rowBlock = c->optRowBlock;
        if (p.scene.plCells) rowBlock = 1;                             // row lists: one row per wave
        f.row_block = (uint32_t)rowBlock;
        f.clearSig = 0;                                                // (the row kernel writes every voxel of the grid)
        DXV_HIP(c, launch_parity_rows(p, rowBlock, fs));
        f.lastRedoParity = -1;
    } else {
        if (queued) {
            // The grid's zeros outside the queued bricks and the queue itself are still good when the frame's last writer was this
            // very launch -- same lists, partition and buffers (the kernel writes the same bricks every time): the frame's signature
            // word says so, every other writer of the grid resets it.  plan = 2, or a grid whose pointer the caller holds: never.
            uint64_t sig = 0;
            auto mix = [&](uint64_t v) { sig = (sig ^ v) * 0x9E3779B97F4A7C15ull; sig ^= sig >> 29; };
            mix(0x7175657565ull); mix(c->listEpoch); mix(p.N); mix(p.nz); mix(p.z0); mix(p.zBlock); mix(p.zPeriod);
            mix(reinterpret_cast<uint64_t>(p.grid)); mix(reinterpret_cast<uint64_t>(p.texels)); mix(reinterpret_cast<uint64_t>(f.dQueue));
            mix(p.planRegionBits); mix(p.planHeavy);
            sig |= 1ull;
            // (dxv_policy.h: every launch builds its queue under plan = 2; a kept queue whose lengths an earlier dxv_sync has read is
            // dealt out by the hardware -- option dispatch: 1 = whenever known, 2 = for partitions of up to 2^25 voxels)
            QueueState qs{};
            qs.optPlan = c->optPlan; qs.optDispatch = c->optDispatch; qs.ptrExposed = f.ptrExposed; qs.keptSig = f.clearSig; qs.lensSig = f.queueLenSig;
            qs.queuedBricks = f.plan_bricks;
            qs.optPrepared = c->optPrepared; qs.prepared = prep >= 0;
            const QueueLaunch how = queue_policy(qs, sig, voxels);
            if (how == QueueLaunch::prepared_hardware) {
                // queue from Init; the grid cleared and every queued brick written inside this launch; the frame keeps nothing
                auto& q = c->prepared[prep];
                q.used = ++c->preparedClock;
                f.clearSig = 0; f.queueLenSig = 0;
                DXV_HIP(c, launch_voxelize_prepared(p, q.lens, q.dLive, c->optPrepClear, &f.plan_waves, fs));
                f.plan_bricks = q.bricks; f.plan_ms = 0.0f;
                f.lastPrepared = prep;
            } else {
                const bool rebuild = how == QueueLaunch::build_and_persistent;
                hipEvent_t pe[2] = {f.evP0, f.evP1};
                const uint32_t* listed = how == QueueLaunch::kept_hardware ? f.queueLens : nullptr;
                if (rebuild) {
                    // the new queue goes into the frame's other header, which the last build left cleared; this build clears the one it leaves
                    const uint32_t target = f.queueHdr ^ 1u;
                    p.queue = f.dQueue + target * kQueueHeaderWords;
                    p.queueZero = f.dQueue + f.queueHdr * kQueueHeaderWords;
                    if (!f.queueOtherClear) DXV_HIP(c, hipMemsetAsync(p.queue, 0, sizeof(uint32_t) * kQueueHeaderWords, fs));
                    f.queueOtherClear = false;                                 // (until this launch is in the stream)
                    f.clearSig = 0; f.queueLenSig = 0;
                }
                DXV_HIP(c, launch_voxelize_queue(p, rebuild, &f.plan_waves, rebuild && c->optEvents ? pe : nullptr, listed, fs));
                if (rebuild) { f.queueHdr ^= 1u; f.queueOtherClear = true; }
                f.clearSig = f.ptrExposed ? 0 : sig;
                f.lastQueued = true; f.lastRebuilt = rebuild;
            }
        } else DXV_HIP(c, launch_voxelize(p, c->optBrick, st, fs));
        if (p.lists) f.lastRedoParity = -1;                        // no column to run out of, nothing to redo
        else {
            DXV_HIP(c, launch_voxelize_redo(p, fs));
            f.lastRedoParity = (int)f.redoParity;
            f.redoParity ^= 1u;
        }
    }
    if (c->optEvents) DXV_HIP(c, hipEventRecord(f.ev1, fs));
    DXV_HIP(c, hipEventRecord(f.evEnd, fs));
    f.timed = c->optEvents != 0;
    f.pending = true;
    return 0;
}


// slices this launch writes: local lz in [0, nzLocal) <-> global z0 + (lz / zBlock) * zPeriod + lz % zBlock
int voxelize_common(dxv_ctx* c, uint32_t N, int mode, uint32_t z0, uint32_t nzLocal, uint32_t zBlock, uint32_t zPeriod)
{
    if (!c->haveScene) return fail(c, "dxv_voxelize: no scene (call dxv_build or dxv_scene_import first)");
    if (mode != DXV_MODE_REFERENCE && mode != DXV_MODE_PARITY) return fail(c, "dxv_voxelize: unknown mode %d", mode);
    if (c->texels && mode != DXV_MODE_REFERENCE) return fail(c, "dxv_voxelize: texel output exists in reference mode only");
    DXV_HIP(c, hipSetDevice(c->device));
    Frame& f = cur_frame(c);
    const hipStream_t fs = cur_stream(c);
    // the frame's previous launch is checked before its grid is reused -- when it can have anything to report: a launch
    // through the lists has no column to run out of, and the next launch simply queues behind it on the frame's stream
    // (no host round trip between back-to-back launches: 20 us of a 0.15 ms launch at 8 ranks)
    if (f.pending && f.lastCanFail && sync_frame(c, c->cur)) return 1;
    const size_t bytes = (size_t)N * N * nzLocal;
    if (bytes > f.gridCap) {
        DXV_HIP(c, hipStreamSynchronize(fs));
        (void)hipFree(f.dGrid); f.dGrid = nullptr; f.gridCap = 0;
        DXV_HIP(c, hipMalloc(&f.dGrid, align256(bytes)));
        f.gridCap = bytes;
        f.clearSig = 0;
        f.ptrExposed = false;                                           // (pointers handed out before are dead)
    }
    if (c->texels && bytes > f.texelCap) {
        DXV_HIP(c, hipStreamSynchronize(fs));
        (void)hipFree(f.dTexels); f.dTexels = nullptr; f.texelCap = 0;
        DXV_HIP(c, hipMalloc(&f.dTexels, align256(bytes * 4)));
        f.texelCap = bytes;
        f.clearSig = 0;
    }
    f.gridBytes = bytes;
    f.grid_dim = N; f.z0 = z0; f.nz = nzLocal;
    f.lastMode = mode; f.lastZBlock = zBlock; f.lastZPeriod = zPeriod;
    return launch_now(c, c->cur);
}

// dxv_sync of one frame: wait for its stream, read its status words, redo the launch with a deeper column if asked to
int sync_frame(dxv_ctx* c, uint32_t i)
{
    Frame& f = c->frames[i];
    const hipStream_t fs = frame_stream(c, i);
    for (int attempt = 0; attempt < 8; ++attempt) {
        // status words and the queue's header in one round trip, into page-locked words
        uint32_t* words = c->pin->status[i];
        const uint32_t* lens = c->pin->queueLens[i];
        const bool readQueue = f.pending && f.lastQueued;
        DXV_HIP(c, hipMemcpyAsync(words, f.dStatus, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, fs));
        if (readQueue) DXV_HIP(c, hipMemcpyAsync(c->pin->queueLens[i], f.dQueue + f.queueHdr * kQueueHeaderWords + queue_len_word(0), sizeof(c->pin->queueLens[i]), hipMemcpyDeviceToHost, fs));
        DXV_HIP(c, hipStreamSynchronize(fs));
        // lists this launch was queued behind without waiting for their verdict: withdrawn -> the launch again, through the tree
        if (settle_lists(c)) return 1;
        if (f.pending && f.usedLists && f.listEpochUsed == c->withdrawnEpoch && c->haveScene && f.grid_dim) {
            f.usedLists = false;
            if (launch_now(c, i, true)) return 1;
            continue;
        }
        const uint32_t status = words[0];
        if (f.pending) {
            f.voxelize_ms = f.timed ? elapsed(f.ev0, f.ev1) : 0.0f;
            f.redo_rays = f.lastRedoParity < 0 ? 0u : words[1 + f.lastRedoParity];
            if (readQueue) {
                f.plan_bricks = 0;
                for (uint32_t x = 0; x < 8u; ++x) {
                    f.queueLens[8u + x] = lens[queue_heavy_word(x) - queue_len_word(0)];
                    f.queueLens[x] = lens[queue_len_word(x) - queue_len_word(0)] + f.queueLens[8u + x];
                    f.plan_bricks += f.queueLens[x];
                }
                f.queueLenSig = f.clearSig;                             // (the queue of this signature: 0 = none kept)
                if (f.lastRebuilt) f.plan_ms = f.timed ? elapsed(f.evP0, f.evP1) : 0.0f;
            }
        }
        f.pending = false;
        if (!status) return 0;
        DXV_HIP(c, hipMemsetAsync(f.dStatus, 0, sizeof(uint32_t), fs));
        if (!c->optStack && c->stackNow < safe_stack(c, f.lastMode) && c->haveScene && f.grid_dim) {
            // grow to the next instantiated depth (at most up to the depth that cannot overflow) and redo
            const int next = stack_round_up(c->stackNow + 1);
            c->stackNow = next < safe_stack(c, f.lastMode) ? next : safe_stack(c, f.lastMode);
            if (launch_now(c, i, true)) return 1;
            continue;
        }
        return fail(c, "voxelize kernel reported status 0x%x (traversal stack overflow: tree height %u, stack %u)",
                    status, c->hdr.treeHeight, f.stack_entries);
    }
    return 0;
}

} // namespace dxvhost

extern "C" {

int dxv_voxelize_async(dxv_ctx* c, uint32_t N, int mode, uint32_t z0, uint32_t nz)
{
    if (!c) return 1;
    if (N < 2 || (N & 1u) || N > 2048) return fail(c, "dxv_voxelize: grid_dim must be even and in [2, 2048], got %u", N);
    if (nz == 0 || z0 >= N || nz > N - z0) return fail(c, "dxv_voxelize: slab [%u, %u+%u) outside the grid (N=%u)", z0, z0, nz, N);
    return voxelize_common(c, N, mode, z0, nz, nz, nz);
}

int dxv_voxelize_interleaved_async(dxv_ctx* c, uint32_t N, int mode, uint32_t rank, uint32_t world, uint32_t zblock)
{
    if (!c) return 1;
    if (N < 2 || (N & 1u) || N > 2048) return fail(c, "dxv_voxelize: grid_dim must be even and in [2, 2048], got %u", N);
    if (!world || rank >= world || !zblock || (zblock & (zblock - 1u)) || N % (zblock * world))
        return fail(c, "dxv_voxelize_interleaved: need rank < world, zblock a power of two and grid_dim %% (zblock * world) == 0 "
                       "(N=%u, world=%u, zblock=%u)", N, world, zblock);
    return voxelize_common(c, N, mode, rank * zblock, N / world, zblock, zblock * world);
}

int dxv_prepare_launch(dxv_ctx* c, uint32_t N, uint32_t z0, uint32_t nz)
{
    if (!c) return 1;
    if (N < 2 || (N & 1u) || N > 2048) return fail(c, "dxv_prepare_launch: grid_dim must be even and in [2, 2048], got %u", N);
    if (nz == 0 || z0 >= N || nz > N - z0) return fail(c, "dxv_prepare_launch: slab [%u, %u+%u) outside the grid (N=%u)", z0, z0, nz, N);
    return prepare_partition(c, N, z0, nz, nz, nz);
}

int dxv_prepare_launch_interleaved(dxv_ctx* c, uint32_t N, uint32_t rank, uint32_t world, uint32_t zblock)
{
    if (!c) return 1;
    if (N < 2 || (N & 1u) || N > 2048) return fail(c, "dxv_prepare_launch: grid_dim must be even and in [2, 2048], got %u", N);
    if (!world || rank >= world || !zblock || (zblock & (zblock - 1u)) || N % (zblock * world))
        return fail(c, "dxv_prepare_launch_interleaved: need rank < world, zblock a power of two and grid_dim %% (zblock * world) == 0 "
                       "(N=%u, world=%u, zblock=%u)", N, world, zblock);
    return prepare_partition(c, N, rank * zblock, N / world, zblock, zblock * world);
}

int dxv_voxelize_interleaved(dxv_ctx* c, uint32_t N, int mode, uint32_t rank, uint32_t world, uint32_t zblock)
{
    if (dxv_voxelize_interleaved_async(c, N, mode, rank, world, zblock)) return 1;
    return dxv_sync(c);
}

int dxv_sync(dxv_ctx* c)
{
    if (!c) return 1;
    DXV_HIP(c, hipSetDevice(c->device));
    return sync_frame(c, c->cur);
}

int dxv_sync_all(dxv_ctx* c)
{
    if (!c) return 1;
    DXV_HIP(c, hipSetDevice(c->device));
    return sync_frames(c);
}

int dxv_voxelize(dxv_ctx* c, uint32_t N, int mode, uint32_t z0, uint32_t nz)
{
    if (dxv_voxelize_async(c, N, mode, z0, nz)) return 1;
    return dxv_sync(c);
}

} // extern "C"
