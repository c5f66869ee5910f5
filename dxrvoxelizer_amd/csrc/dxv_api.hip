// dxv_api.hip -- the C-ABI of libdxv.so (include/dxv.h): context, device memory, build and
// voxelize orchestration on one HIP stream.  There is no CPU fallback anywhere in this file:
// without a HIP device dxv_create fails.
#include "../../include/dxv.h"
#include "dxv_device.h"
#include "dxv_raycast.h"
#include "dxv_dirmap.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

using namespace dxv;

namespace {
#if defined(DXV_QUEUE_TIMES)
constexpr uint32_t kRedoCap = 1u << 21;   // (diagnostic build: the list doubles as the buffer of per-workgroup time stamps)
#else
constexpr uint32_t kRedoCap = 1u << 16;   // rays per launch the redo pass takes before the column is grown instead
#endif
thread_local std::string g_createError;

size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
}

struct dxv_ctx {
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t copyStream = nullptr;    // dxv_update_vertices: the upload runs beside the frames' launches (made at its first call)
    bool vbCopyQueued = false;           // dxv_update_vertices_device left a copy into the vertex buffer on `stream` (until the next refit / build)
    std::string err;

    // mesh (caller's layout)
    float* dVb = nullptr;
    uint32_t* dIb = nullptr;
    uint32_t T = 0, V = 0;
    float bound[4] = {0, 0, 0, 0};
    bool haveMesh = false;

    // scene blob
    uint8_t* dScene = nullptr;
    size_t sceneBytes = 0;
    SceneHeader hdr{};
    bool haveScene = false;
    bool haveHierarchy = false;      // dxv_build ran for the resident mesh: keys, links and parent words are in place for dxv_refit
                                     // (stays true when a refit fails on bad vertices: the next good update refits again)

    // build scratch
    uint64_t* dKeys = nullptr;
    uint64_t* dKeysTmp = nullptr;
    uint32_t* dHist = nullptr;
    uint32_t* dParents = nullptr;
    void* dPyramid = nullptr;        // min/max pyramid over the leaf boxes (refit = 1: dxv_build and dxv_refit)
    uint32_t* dFlags = nullptr;
    uint32_t* dFlags2 = nullptr;
    uint32_t* dRootInfo = nullptr;
    uint32_t scratchT = 0;

    // outputs: FrameCount sets of grid / texel image / status words / redo list / stream, the way the reference's
    // Voxelizer owns FrameCount grids (Content/Voxelizer.h:24, :110); one scene and one set of lists serve them all
    struct Frame {
        hipStream_t ownStream = nullptr; // frames 1.. launch on a stream of their own; frame 0 on the context's stream
        uint8_t* dGrid = nullptr;
        size_t gridCap = 0, gridBytes = 0;
        uint32_t* dTexels = nullptr;
        size_t texelCap = 0;
        uint32_t* dStatus = nullptr;     // [0] status bits, [1], [2] redo-list counters (alternating launches)
        uint64_t* dRedo = nullptr;       // voxels whose LDS column was too small, finished by the redo pass
        uint32_t redoParity = 0;
        int lastRedoParity = -1;         // counter of the last launch (-1: that launch has none)
        hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the frame's last launch
        int lastMode = 0;
        uint32_t lastZBlock = 1, lastZPeriod = 1;
        bool pending = false;            // a voxelize launch has not been checked by dxv_sync yet
        bool timed = true;               // ... and it was bracketed by the frame's two events (option events)
        bool lastCanFail = true;         // ... and it can report something (a walk's column can run out; the lists have no column)
        bool ready = false;              // status words, redo list, events and stream exist
        uint64_t clearSig = 0;           // the partial launch whose memset this grid still carries (launch_shape, traverse.hip); 0 = none
        bool ptrExposed = false;         // dxv_grid_device_ptr handed this grid out for writing: the caller may write through the pointer at any
                                         // time, so no memset is ever kept for it again (until the grid is reallocated)
        // launch fields of dxv_stats
        float voxelize_ms = 0.0f;
        uint32_t grid_dim = 0, z0 = 0, nz = 0, stack_entries = 0, redo_rays = 0, row_block = 0, list_entries = 0, list_res = 0;
        uint32_t plan_bricks = 0, plan_waves = 0;
        float plan_ms = 0.0f;
        // work queue of the lists kernel (traverse.hip): the frame's own, written and read on the frame's stream only
        uint32_t* dQueue = nullptr;      // two headers, then the slots (dxv_device.h)
        size_t queueWords = 0;           // allocated 32-bit words
        uint32_t queueHdr = 0;           // the header (0 / 1) of the frame's current queue; the next build takes the other one ...
        bool queueOtherClear = false;    // ... which is all zero (cleared at the allocation, then by every build's k_plan_bricks)
        bool lastQueued = false;         // the frame's last launch went through the queue (dxv_sync reads its lengths for the stats)
        bool lastRebuilt = false;        // ... and built it (plan_ms is that build's)
        hipEvent_t evP0 = nullptr, evP1 = nullptr;   // around the queue build of the frame's last launch (option events)
        uint32_t queueLens[16] = {};     // the lengths of the frame's eight queues and how many of each are heavy, as last read by dxv_sync ...
        uint64_t queueLenSig = 0;        // ... for the queue of this signature (clearSig); 0: not known
        hipEvent_t evEnd = nullptr;      // behind the frame's last launch, always recorded: what a refit on another stream waits for on the device
        bool usedLists = false;          // the frame's last launch went through the direction-space lists ...
        uint64_t listEpochUsed = 0;      // ... of this build (a build whose deferred check fails is withdrawn: settle_lists, sync_frame)
    };
    Frame frames[DXV_FRAME_COUNT];
    uint32_t cur = 0;                    // dxv_set_frame
    bool texels = false;
    unsigned long long* dCount = nullptr;
    uint8_t* dPacked = nullptr;
    size_t packedCap = 0;
    uint32_t* dImage = nullptr;
    size_t imageCap = 0;
    // direction-space lists of the reference rule (dxv_dirmap.h), built lazily from the scene's triangle records
    DirCell* dListCells = nullptr;
    DirEntry* dListEntries = nullptr;
    size_t listCellCap = 0, listEntryCap = 0;
    uint32_t listEntries = 0, listRes = 0;
    int listState = 0;               // 0: not built for this scene, 1: built, -1: over the cap for this scene (tree walk)
    int optLists = 1;                // reference rule through the lists (-40...-60 % against the tree walk, profiles/r01/final/ab_lists.jsonl):
                                     // 1 = from a scene's second launch on (from the first when that launch is large: build_lists), 2 = from the first, 0 = tree walk
    int optListRes = 0;              // texels per face side; 0 = by triangle count (list_resolution)
    uint32_t listResFloor = 0;       // automatic resolution: not below this (512 once a scene of 20 k triangles or more that was not refitted
    bool listFloorTried = false;     // is launched AGAIN: a static scene -- the finer map is 10 - 20 % faster at every grid size since texels
                                     // outside a triangle's outline get no entry, and costs a build of 1.5 - 2 x)
    bool refitted = false;           // dxv_refit has run since dxv_build: the mesh is being animated, its lists are built for one launch
    float listMs = 0.0f;
    uint8_t *dListScratchA = nullptr, *dListScratchB = nullptr;   // scratch of the list build, kept between builds (a refit rebuilds them)
    size_t listScratchACap = 0, listScratchBCap = 0;
    // The dynamic case (a mesh refitted every frame, XUSGRayTracing.h:13-22) with ONE host round trip per frame instead of four:
    //  * dxv_refit queues the lists' counting pass behind its own kernels when the scene had lists (specRes: the map it counted
    //    on) and reads root box and entry total in one synchronisation;
    //  * a build made inside a launch does not wait for its own end: the launch is queued behind it, and the one thing the host
    //    must still look at -- a texel with more entries than its 16-bit count holds -- is looked at when the frame is
    //    synchronised (settle_lists); lists that fail there are withdrawn and the frame is launched again through the tree.
    // Everything the device reports goes through page-locked words (a copy into pageable memory blocks the host until the
    // stream has drained: 30 us of idle GPU per copy in the refit loop's trace).
    struct Pinned {
        uint32_t rootInfo[16];
        unsigned long long listTotal;
        uint32_t listLongest, pad;
        uint32_t status[DXV_FRAME_COUNT][4];
        uint32_t queueLens[DXV_FRAME_COUNT][16 * 64];    // the sixteen count words of a frame's queue (light and heavy bricks of the eight queues; each in a 256-byte line of its own)
    };
    Pinned* pin = nullptr;
    hipEvent_t evList[4] = {};       // around the counting pass, around the rest of the build
    uint32_t specRes = 0;            // the counting pass for the current scene has run on this map (records, counts, total in place)
    bool listCheckPending = false;   // lists in use whose longest texel has not been looked at yet
    hipStream_t listCheckStream = nullptr;
    uint64_t withdrawnEpoch = 0;     // listEpoch of the last build that failed its deferred check
    uint32_t launchesOfScene = 0;    // reference-rule launches since the scene last changed (build / refit / import)
    // max-mip of the lists' far radii (dxv_dirmap.h): made with the lists, what a launch's work queue is probed against
    uint16_t* dMip = nullptr;
    size_t mipCap = 0;               // 16-bit words
    uint64_t listEpoch = 0;          // counts list builds / imports: a frame's queue belongs to the lists it was probed against
    int optPlan = 2;                 // work queue of the lists kernel (live bricks only, built on the device inside the stream): 0 = none (brick box
                                     // in Morton order), 1 = built when lists, partition or buffers differ from the frame's last launch (opt-in), 2 = on every launch (default: nothing carried)
    int optQueueWaves = 0;           // persistent waves of a queue launch; 0 = what the device holds at once
    int optQueueHeads = 8;           // heads per queue (persistent waves): 1, 2, 4, 8
    int optPlanRegion = 0;           // log2 of the run of Morton bricks dealt to one queue: 6, 7, 8; 0 = by the partition's size (plan_region_bits)
    int optPlanHeavy = 0;            // list length beyond which a brick starts early; 0 = long for this scene (k_dm_heavy_thresholds), 65535: no brick does
    int optFuse = 1;                 // 1: the queue build clears the grid as well (one kernel in front of the brick kernel); 0: memsets in front of it
    int optDispatch = 1;             // a kept queue whose lengths the host knows: 0 = persistent waves all the same, 1 = one workgroup per
                                     // queued brick dealt out by the hardware (-1 ... -10 % per launch, and back-to-back launches overlap
                                     // their ends: profiles/r04/ab_dispatch_kept_queue.jsonl), 2 = that for partitions of up to 2^25 voxels only
    int optEvents = 1;               // bracket every launch with two HIP events (stats.voxelize_ms); 0: none (a caller timing its own loop)
    // row lists of the parity rule (dirmap.hip): built like the direction-space lists, on a scene's second parity launch or on
    // a large first one; not part of the scene blob (an importing context builds its own from the triangle records: 0.2 ms)
    uint32_t* dPlCells = nullptr;
    uint32_t* dPlEntries = nullptr;
    uint32_t* dPlScratch = nullptr;  // counts, offsets, block sums of the build
    size_t plCellCap = 0, plEntryCap = 0, plScratchCap = 0;
    uint32_t plEntries = 0, plRes = 0;
    int plState = 0;                 // 0: not built for this scene, 1: built, -1: over the cap (tree walk)
    int optPlistRes = 0;             // texels per side of the row lists' grid; 0 = by triangle count
    int optPlists = 1;               // 1 = from a scene's second parity launch, 2 = from the first, 0 = tree walk
    uint32_t parityLaunchesOfScene = 0;
    float plMs = 0.0f;
    int nodesStale = 0;              // what a build / refit left behind (ensure_nodes brings it up to date before anything reads it):
                                     // 1 = the four-box copy (nodes64); 2 = every node box (dxv_refit stopped at the pyramid: deferBoxes)
    int listOpt = 0;                 // the listres option the current lists (or the decision against them) were made with
    uint8_t* dEmpty = nullptr;       // display pass: empty-brick flags of the grid
    size_t emptyCap = 0;
    int optSkipEmpty = 1;    // display pass: skip the samples of empty 8^3 bricks (same image)
    float renderMs = 0.0f;

    hipEvent_t ev[10] = {};
    dxv_stats stats{};

    // options
    int optBrick = 4;        // 4x4x4 voxels = one wavefront per workgroup (fastest in the r01 sweeps)
    int optStack = 0;        // 0 = adaptive (start small, grow on overflow), else forced depth
    int optDeferBoxes = 1;   // dxv_refit with lists wanted: node boxes only when a tree walk asks for them (0: always, as dxv_build does)
    int optRefit = 1;        // box merge of build and refit: 1 = min/max pyramid (default), 2 = level sweeps, 0 = atomic one-pass climb (17-30x slower, cross-check)
    int optMorton = 1;       // Morton brick order
    int optQueue = 1;        // postponed-leaf traversal
    int optSubbox = 1;       // launch only the bricks around the scene's root box, memset the rest
    int optWide = 2;         // reference rule: 2 = four-box nodes on wave-uniform visits (-2...-7 % everywhere measured),
                             // 1 = on every visit (-8 % on low-poly meshes, +10 % on 1 M triangles at 256^3), 0 = binary only
    int optRows = 1;         // parity mode: one tree walk per grid row (k_parity_rows) instead of per voxel
    int optRowBlock = 0;     // rows per side of a wave's block of rows: 0 = by triangle size, 1, 2
    int optAblate = 0;       // timing-only variants of the lists kernel (results are wrong by design; tools/ablate.py)
    int optRegion = 6;       // log2 bricks per XCD region (64 bricks: balanced and L2 friendly in the r01 sweeps)
    int optStack0 = 20;      // adaptive mode starts with this many entries (stack + leaf queue share them)
    int stackNow = 20;       // adaptive: LDS stack entries per thread currently in use for this scene
};

namespace {

int fail(dxv_ctx* c, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_createError = buf;
    return 1;
}

#define DXV_HIP(c, call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail((c), "%s failed: %s", #call, hipGetErrorString(e_));    \
    } while (0)

using Frame = dxv_ctx::Frame;
Frame& cur_frame(dxv_ctx* c) { return c->frames[c->cur]; }
hipStream_t frame_stream(dxv_ctx* c, uint32_t i) { return i == 0 ? c->stream : c->frames[i].ownStream; }
hipStream_t cur_stream(dxv_ctx* c) { return frame_stream(c, c->cur); }

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

int sync_frame(dxv_ctx* c, uint32_t i);
int settle_lists(dxv_ctx* c);

// Everything that changes what the frames read (mesh, scene, lists, options that rebuild) first lets every
// frame finish -- including the status check and, if a launch asked for it, the relaunch against the OLD scene.
int sync_frames(dxv_ctx* c)
{
    for (uint32_t i = 0; i < DXV_FRAME_COUNT; ++i)
        if (c->frames[i].ready && sync_frame(c, i)) return 1;
    return settle_lists(c);
}

Node* scene_nodes(dxv_ctx* c) { return reinterpret_cast<Node*>(c->dScene + c->hdr.offNodes); }
Node32* scene_nodes32(dxv_ctx* c) { return reinterpret_cast<Node32*>(c->dScene + c->hdr.offNodes32); }
Node64* scene_nodes64(dxv_ctx* c) { return reinterpret_cast<Node64*>(c->dScene + c->hdr.offNodes64); }
TriPos* scene_tripos(dxv_ctx* c) { return reinterpret_cast<TriPos*>(c->dScene + c->hdr.offTriPos); }
TriNrm* scene_trinrm(dxv_ctx* c) { return reinterpret_cast<TriNrm*>(c->dScene + c->hdr.offTriNrm); }

void layout_scene(SceneHeader& h, uint32_t T, uint32_t V, bool wide)
{
    memset(&h, 0, sizeof(h));
    h.magic = kSceneMagic;
    h.version = kSceneVersion;
    h.numTris = T;
    h.numVerts = V;
    h.numNodes = T > 1 ? T - 1 : 1;
    h.offNodes = align256(sizeof(SceneHeader));
    h.offNodes32 = align256(h.offNodes + sizeof(Node) * (size_t)h.numNodes);
    h.offNodes64 = align256(h.offNodes32 + sizeof(Node32) * (size_t)h.numNodes);
    h.hasWide = wide ? 1u : 0u;
    h.offTriPos = align256(h.offNodes64 + (wide ? sizeof(Node64) * (size_t)h.numNodes : 0));
    h.offTriNrm = align256(h.offTriPos + sizeof(TriPos) * (size_t)T);
    h.totalBytes = align256(h.offTriNrm + sizeof(TriNrm) * (size_t)T);
}

int alloc_scene(dxv_ctx* c, uint32_t T, uint32_t V, bool wide)
{
    SceneHeader h;
    layout_scene(h, T, V, wide);
    if (c->dScene && c->sceneBytes != h.totalBytes) { (void)hipFree(c->dScene); c->dScene = nullptr; }
    if (!c->dScene) DXV_HIP(c, hipMalloc(&c->dScene, h.totalBytes));
    c->sceneBytes = h.totalBytes;
    c->hdr = h;
    return 0;
}

void free_scratch(dxv_ctx* c)
{
    (void)hipFree(c->dKeys); (void)hipFree(c->dKeysTmp); (void)hipFree(c->dHist); (void)hipFree(c->dParents);
    (void)hipFree(c->dPyramid); c->dPyramid = nullptr;
    (void)hipFree(c->dFlags); (void)hipFree(c->dFlags2);
    c->dKeys = c->dKeysTmp = nullptr; c->dHist = c->dParents = c->dFlags = c->dFlags2 = nullptr;
    c->scratchT = 0;
}

int alloc_scratch(dxv_ctx* c, uint32_t T)
{
    if (c->scratchT == T) return 0;
    free_scratch(c);
    DXV_HIP(c, hipMalloc(&c->dKeys, sizeof(uint64_t) * (size_t)T));
    DXV_HIP(c, hipMalloc(&c->dKeysTmp, sizeof(uint64_t) * (size_t)T));
    DXV_HIP(c, hipMalloc(&c->dHist, sizeof(uint32_t) * (size_t)radix_sort_hist_words(T)));
    DXV_HIP(c, hipMalloc(&c->dParents, sizeof(uint32_t) * (2 * (size_t)T)));
    DXV_HIP(c, hipMalloc(&c->dFlags, sizeof(uint32_t) * (size_t)T));
    DXV_HIP(c, hipMalloc(&c->dFlags2, sizeof(uint32_t) * (size_t)T));
    c->scratchT = T;
    return 0;
}

float elapsed(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return -1.0f;
    return ms;
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

// Build the direction-space lists of the current scene (one-off per scene; synchronous).  Scenes whose
// lists would exceed 256 entries per triangle + 64 M (triangles through the grid centre cover whole
// faces) keep the tree walk: listState = -1.
// Texels per face side.  Measured optimum (tools/ab_lists.py): 5-10 entries per texel -- coarser maps
// have long lists, finer ones stop fitting the caches: 128 below 20 k triangles, 256 up to 3 M (512 when the
// 256 map holds more than 10 entries per texel and the scene is presumed static: build_lists), 512 beyond.
uint32_t list_resolution(const dxv_ctx* c)
{
    if (c->optListRes) return (uint32_t)c->optListRes;
    return c->hdr.numTris < 20000u ? 128u : c->hdr.numTris < 3000000u ? 256u : 512u;
}

// firstLaunchVoxels != 0: called for the FIRST launch of a scene (option lists=1), which may be its only one -- a mesh that
// is refitted every frame.  The build then has to pay for itself on this launch: after the counting pass (0.1 ms) it goes
// on only when what the lists save over the tree walk (about 10 ps per voxel; more in deep scenes, in proportion to the
// mean list length) exceeds what the rest of the build costs (0.1 ms + 0.15 ns per entry: 0.65 ms for 3.8 M entries).
// Declined: listState stays 0, the launch walks the tree, the second launch builds the lists.
int ensure_nodes(dxv_ctx* c, hipStream_t stream);      // (below, with the build)
int build_lists_into(dxv_ctx* c, hipStream_t stream, uint64_t firstLaunchVoxels, bool defer);
int settle_lists(dxv_ctx* c);
// A context that HAS working lists (the one-time move to the 512 map for launches at 1024^3 and beyond, an explicit listres)
// builds the new ones beside them and swaps only when the build succeeded: out of memory, or lists over the caps on the new map,
// leave the scene on the lists it had instead of on the tree walk (three times slower).
// defer: the caller queues its launch behind the build and lets the frame's synchronisation look at the build's verdict
// (settle_lists); otherwise the build is finished and checked when this returns.
int build_lists(dxv_ctx* c, hipStream_t stream, uint64_t firstLaunchVoxels = 0, bool defer = false)
{
    if (settle_lists(c)) return 1;
    if (c->listState != 1) return build_lists_into(c, stream, firstLaunchVoxels, defer);
    DirCell* oldCells = c->dListCells; DirEntry* oldEntries = c->dListEntries; uint16_t* oldMip = c->dMip;
    const size_t oldCellCap = c->listCellCap, oldEntryCap = c->listEntryCap, oldMipCap = c->mipCap;
    const uint32_t oldN = c->listEntries, oldRes = c->listRes;
    const int oldOpt = c->listOpt;
    const float oldMs = c->listMs;
    c->dListCells = nullptr; c->dListEntries = nullptr; c->dMip = nullptr; c->listCellCap = c->listEntryCap = c->mipCap = 0;
    c->listState = 0;
    const int rc = build_lists_into(c, stream, firstLaunchVoxels, false);
    if (rc == 0 && c->listState == 1) {                                 // the new lists stand: the old ones go
        (void)hipFree(oldCells); (void)hipFree(oldEntries); (void)hipFree(oldMip);
        return 0;
    }
    (void)hipFree(c->dListCells); (void)hipFree(c->dListEntries); (void)hipFree(c->dMip);
    c->dListCells = oldCells; c->dListEntries = oldEntries; c->dMip = oldMip;
    c->listCellCap = oldCellCap; c->listEntryCap = oldEntryCap; c->mipCap = oldMipCap;
    c->listEntries = oldN; c->listRes = oldRes; c->listState = 1; c->listMs = oldMs;
    // (the option the caller asked for counts as answered: the next launch does not try the same build again; the epoch moves on
    // because the frames' queues were probed against buffers that may have moved -- they have not, but a rebuild is cheap)
    c->listOpt = rc == 0 ? c->optListRes : oldOpt;
    ++c->listEpoch;
    return rc;
}

// scratch of the counting pass (records, counts, offsets, block sums, total), kept with the context up to 16 GiB: an allocation
// costs ~0.1 ms, as much as a pass of the build -- and hundreds of ms for the gigabytes of a 10 M-triangle scene
struct ListScratchA { DirRecord* rec; uint32_t *counts, *offsets, *pairs, *sums; unsigned long long* total; size_t bytes; };
ListScratchA list_scratch_a(uint8_t* base, uint32_t T)
{
    const size_t n6 = 6 * (size_t)T, nb = (n6 + 1023) / 1024;
    const size_t offCounts = align256(n6 * sizeof(DirRecord)), offOffsets = offCounts + align256(n6 * 4), offPairs = offOffsets + align256(n6 * 4),
                 offSums = offPairs + align256(n6 * 4), offTotal = offSums + align256((nb + 1) * 4);
    return {reinterpret_cast<DirRecord*>(base), reinterpret_cast<uint32_t*>(base + offCounts), reinterpret_cast<uint32_t*>(base + offOffsets),
            reinterpret_cast<uint32_t*>(base + offPairs), reinterpret_cast<uint32_t*>(base + offSums), reinterpret_cast<unsigned long long*>(base + offTotal), offTotal + 256};
}

// The verdict of a build whose caller did not wait for it: time, and the one thing only the host can act on -- a texel with
// more entries than its 16-bit count holds.  Such lists are withdrawn (tree walk for this scene); frames launched with them
// are launched again when they are synchronised (sync_frame).
int settle_lists(dxv_ctx* c)
{
    if (!c->listCheckPending) return 0;
    DXV_HIP(c, hipEventSynchronize(c->evList[3]));
    c->listCheckPending = false;
    c->listMs = elapsed(c->evList[0], c->evList[1]) + elapsed(c->evList[2], c->evList[3]);
    if (c->pin->listLongest > 0xffffu) {
        c->withdrawnEpoch = c->listEpoch;
        c->listState = -1; c->listEntries = 0; c->listOpt = c->optListRes;
    }
    return 0;
}

int build_lists_into(dxv_ctx* c, hipStream_t stream, uint64_t firstLaunchVoxels, bool defer)
{
    const uint32_t T = c->hdr.numTris;
    uint32_t R = list_resolution(c);
    if (!c->optListRes && c->listResFloor > R) R = c->listResFloor;
    // dxv_refit has run the counting pass already (and read its total with the root box)?
    const bool counted = c->specRes != 0 && c->dListScratchA && (c->optListRes ? (uint32_t)c->optListRes == c->specRes : c->listResFloor <= c->specRes);
    if (counted) R = c->specRes;
    c->specRes = 0;
    const size_t n6 = 6 * (size_t)T, nb = (n6 + 1023) / 1024;
    // scratch in two allocations (an allocation costs ~0.1 ms, as much as a pass): per-(triangle, face)
    // arrays now, the key buffers once the number of entries is known
    // (kept with the context up to 16 GiB each: an allocation costs ~0.1 ms, as much as a pass of the build -- and hundreds of ms
    // for the gigabytes of a 10 M-triangle scene)
    uint8_t *scratchA = nullptr, *scratchB = nullptr;
    auto scratch = [&](uint8_t*& keep, size_t& cap, size_t bytes, uint8_t*& out) -> hipError_t {
        if (bytes <= cap && (cap < (256ull << 20) || bytes >= cap / 4)) { out = keep; return hipSuccess; }    // (a much smaller scene gives the gigabytes back)
        (void)hipFree(keep); keep = nullptr; cap = 0;
        const hipError_t err = hipMalloc(&out, bytes);
        if (err == hipSuccess && bytes <= (16ull << 30)) { keep = out; cap = bytes; }    // (kept: a multi-GB hipMalloc is 0.1 - 0.3 s, ten builds' worth)
        return err;
    };
    auto release = [&]() {
        if (scratchA != c->dListScratchA) (void)hipFree(scratchA);
        if (scratchB != c->dListScratchB) (void)hipFree(scratchB);
        scratchA = scratchB = nullptr;
    };
    // The lists are an optional accelerator: when their memory cannot be had the scene keeps the tree walk
    // (listState = -1, like a scene whose lists would be too long); only launch and sync errors are errors.
    auto bail = [&](hipError_t e, const char* what) {
        release();
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();                   // clear the sticky allocation error
            c->listState = -1; c->listEntries = 0; c->listOpt = c->optListRes;
            return 0;
        }
        return fail(c, "lists: %s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    (void)nb;
    if (counted) scratchA = c->dListScratchA;
    else if ((e = scratch(c->dListScratchA, c->listScratchACap, list_scratch_a(nullptr, T).bytes, scratchA)) != hipSuccess) return bail(e, "hipMalloc");
    const ListScratchA sa = list_scratch_a(scratchA, T);
    DirRecord* rec = sa.rec;
    uint32_t *counts = sa.counts, *offsets = sa.offsets, *sums = sa.sums;
    unsigned long long* dTotal = sa.total;
    unsigned long long total = 0;
    auto recount = [&](uint32_t res) -> int {
        R = res;
        if ((e = dirmap_count(scene_tripos(c), T, R, rec, counts, sa.pairs, dTotal, stream)) != hipSuccess) return bail(e, "dirmap_count");
        (void)hipEventRecord(c->evList[1], stream);
        if ((e = hipMemcpyAsync(&c->pin->listTotal, dTotal, sizeof(total), hipMemcpyDeviceToHost, stream)) != hipSuccess) return bail(e, "hipMemcpyAsync");
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
        total = c->pin->listTotal;
        return 0;
    };
    if (counted) total = c->pin->listTotal;
    else {
        (void)hipEventRecord(c->evList[0], stream);
        if (recount(R)) return 1;
    }
    // automatic resolution, from the mean list length A = entries per texel (it hardly depends on the map: it is the
    // number of triangles a direction meets, at any depth):
    //  * 10 < A <= 32 on the 256 map: the 512 map is faster for some scenes (bunny x16 1.49 -> 1.42 ms; dragon x9 0.79 ->
    //    0.82, torus-1M with 9.6 per texel the same) at 2 - 3 x the build time and memory -- taken when the scene is
    //    presumed static (not on a first-launch build, which must pay for itself at once);
    //  * A > 32: the scene is deep in every direction (soups: hundreds of triangles behind one another).  A ray still
    //    reads only the part of its list between its start and its first hit (the lists are sorted by far radius and the
    //    scan stops behind the hit, dxv_dirmap.h), so what matters is the size of the structure, which grows with the
    //    square of the map: the 256 map while it stays below 320 M entries, else the 128 map.
    auto perTexel = [&]() { return (double)total / (6.0 * R * R); };
    if (firstLaunchVoxels) {
        const double depth = perTexel() > 10.0 ? perTexel() / 10.0 : 1.0;
        const double gainMs = (double)firstLaunchVoxels * 1e-8 * depth, buildMs = 0.1 + 0.15e-6 * (double)total;
        if (gainMs < buildMs) { release(); return 0; }
    }
    // (round 4: with no entries for texels outside a triangle's outline the 512 map beats the 256 map at every grid size measured,
    // 128^3 to 1024^3, by 5 - 22 % -- profiles/r04/ab_texels_outside_the_outline.jsonl -- so every scene that is presumed static
    // takes it; a mesh that is being refitted, or a first launch that must pay for its build at once, keeps the base map)
    const bool oneLaunch = firstLaunchVoxels != 0 || c->refitted;
    if (!c->optListRes && !oneLaunch && R == 256u && perTexel() <= 32.0) {
        if (recount(512u)) return 1;
    } else if (!c->optListRes && perTexel() > 32.0) {
        if (R != 256u && recount(256u)) return 1;
        if (total > (320ull << 20) && recount(128u)) return 1;
    }
    const unsigned long long cap = 256ull * T + (64ull << 20);
    if (total > cap || total > 0x7fffffffull || (unsigned long long)T > (1ull << dm_key_layout(R).triBits) || T > kDmTriMask) {
        release();
        c->listState = -1;
        c->listEntries = 0;
        c->listOpt = c->optListRes;
        return 0;
    }
    const uint32_t n = (uint32_t)total;
    const size_t cells = 6 * (size_t)R * R;
    if (cells > c->listCellCap) {
        (void)hipFree(c->dListCells); c->dListCells = nullptr; c->listCellCap = 0;
        if ((e = hipMalloc(&c->dListCells, cells * sizeof(DirCell))) != hipSuccess) return bail(e, "hipMalloc");
        c->listCellCap = cells;
    }
    if ((size_t)n > c->listEntryCap) {
        (void)hipFree(c->dListEntries); c->dListEntries = nullptr; c->listEntryCap = 0;
        if ((e = hipMalloc(&c->dListEntries, ((size_t)n + 4) * sizeof(DirEntry))) != hipSuccess) return bail(e, "hipMalloc");   // (+ spare ones: a scan round loads four)
        c->listEntryCap = n;
    }
    const size_t keyBytes = align256(((size_t)n + 1) * 8);
    if ((e = scratch(c->dListScratchB, c->listScratchBCap, 2 * keyBytes + sizeof(uint32_t) * (size_t)radix_sort_hist_words(n ? n : 1), scratchB)) != hipSuccess) return bail(e, "hipMalloc");
    uint64_t* keys = reinterpret_cast<uint64_t*>(scratchB);
    uint64_t* keysTmp = reinterpret_cast<uint64_t*>(scratchB + keyBytes);
    uint32_t* hist = reinterpret_cast<uint32_t*>(scratchB + 2 * keyBytes);
    (void)hipEventRecord(c->evList[2], stream);
    c->pin->listLongest = 0;
    if ((e = dirmap_fill(T, R, rec, counts, sa.pairs, dTotal, offsets, sums, keys, keysTmp, hist, n, c->dListCells, c->dListEntries, &c->pin->listLongest, stream)) != hipSuccess)
        return bail(e, "dirmap_fill");

    // the max-mip of the texels' far radii goes with the lists (a launch's work queue is probed against it)
    if (dm_mip_words(R) > c->mipCap) {
        (void)hipFree(c->dMip); c->dMip = nullptr; c->mipCap = 0;
        if ((e = hipMalloc(&c->dMip, sizeof(uint16_t) * (size_t)dm_mip_buffer_words(R))) != hipSuccess) return bail(e, "hipMalloc");   // (far radii, entry counts)
        c->mipCap = dm_mip_words(R);
    }
    if ((e = dirmap_mip(c->dListCells, R, c->dMip, stream)) != hipSuccess) return bail(e, "dirmap_mip");
    if ((e = hipEventRecord(c->evList[3], stream)) != hipSuccess) return bail(e, "hipEventRecord");
    // scratch that is not kept (over 16 GiB) is freed here: hipFree waits for the device
    release();
    c->listEntries = n;
    c->listRes = R;
    c->listState = 1;
    c->listOpt = c->optListRes;
    ++c->listEpoch;                                       // (work queues probed against older lists are stale)
    // a texel with more entries than its 16-bit count holds: tree walk -- decided by settle_lists, now or when the frame
    // that is launched behind this build is synchronised
    c->listCheckPending = true; c->listCheckStream = stream;
    return defer ? 0 : settle_lists(c);
}

// Row lists of the parity rule (dirmap.hip).  Resolution: the finest grid, from 512 (below 20 k triangles), 2048 (up to 3 M) or
// 4096 texels per side downwards, whose lists stay within 24 entries per triangle + 8 M (an entry is 4 bytes; measured at
// 512^3, 1 M triangles: 256 -> 0.62 ms, 512 -> 0.34, 1024 -> 0.24, 2048 -> 0.20; the walk over the tree: 0.65); scenes over that
// cap on every grid (big triangles cover many texels) or with more than 256 entries per texel keep the tree walk
// (plState = -1), as does a context that cannot allocate the lists.
int build_plists(dxv_ctx* c, hipStream_t stream)
{
    const uint32_t T = c->hdr.numTris;
    uint32_t R = c->optPlistRes ? (uint32_t)c->optPlistRes : T < 20000u ? 512u : T < 3000000u ? 2048u : 4096u;
    const size_t n = (size_t)R * R, nb = (n + 1023) / 1024;             // (scratch for the finest grid tried)
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess) (void)hipEventRecord(t0, stream);
    auto done = [&](int state) {
        if (t0) (void)hipEventDestroy(t0);
        if (t1) (void)hipEventDestroy(t1);
        c->plState = state;
        return 0;
    };
    auto oom = [&](hipError_t e, const char* what) {
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); c->plEntries = 0; return done(-1); }
        (void)done(0);
        return fail(c, "row lists: %s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    const size_t scratchWords = 2 * n + nb + 1 + 6;                     // counts, offsets, sums, two 64-bit words (total, largest rectangle)
    if (scratchWords > c->plScratchCap) {
        (void)hipFree(c->dPlScratch); c->dPlScratch = nullptr; c->plScratchCap = 0;
        if ((e = hipMalloc(&c->dPlScratch, scratchWords * sizeof(uint32_t) + 8)) != hipSuccess) return oom(e, "hipMalloc");
        c->plScratchCap = scratchWords;
    }
    uint32_t* counts = c->dPlScratch;
    uint32_t* offsets = counts + n;
    uint32_t* sums = offsets + n;
    unsigned long long* dTotal = reinterpret_cast<unsigned long long*>(c->dPlScratch + ((2 * n + nb + 1 + 1) & ~(size_t)1));     // two words
    unsigned long long tot[2] = {0, 0};
    const unsigned long long cap = 24ull * T + (8ull << 20), rectCap = 16384;    // (a thread of the fill walks its triangle's rectangle alone)
    for (;;) {
        if ((e = parity_lists_total(scene_tripos(c), T, R, dTotal, stream)) != hipSuccess) return oom(e, "parity_lists_total");
        if ((e = hipMemcpyAsync(tot, dTotal, sizeof(tot), hipMemcpyDeviceToHost, stream)) != hipSuccess) return oom(e, "hipMemcpyAsync");
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return oom(e, "hipStreamSynchronize");
        // the finest grid that fits the caps: a finer grid has more entries but shorter lists (fewer false candidates per row)
        if ((tot[0] <= cap && tot[1] <= rectCap) || R <= 256u) break;
        if (c->optPlistRes) break;
        R >>= 1;
    }
    const unsigned long long total = tot[0];
    if (tot[1] > rectCap) { c->plEntries = 0; return done(-1); }       // a triangle facing the rays covers the plane: the tree walk stays
    // over the cap even on the coarsest grid, or deep in every row (soups: hundreds of triangles behind one another -- the
    // row's work is the triangles themselves, and a coarse grid only adds false candidates to them): the tree walk stays
    if (total > cap || total > 0x7ffffff0ull || (double)total > 256.0 * (double)R * (double)R) { c->plEntries = 0; return done(-1); }
    const size_t cellWords = 2 * (size_t)R * R;
    if (cellWords > c->plCellCap) {
        (void)hipFree(c->dPlCells); c->dPlCells = nullptr; c->plCellCap = 0;
        if ((e = hipMalloc(&c->dPlCells, cellWords * sizeof(uint32_t))) != hipSuccess) return oom(e, "hipMalloc");
        c->plCellCap = cellWords;
    }
    if ((size_t)total + 8 > c->plEntryCap) {
        (void)hipFree(c->dPlEntries); c->dPlEntries = nullptr; c->plEntryCap = 0;
        if ((e = hipMalloc(&c->dPlEntries, ((size_t)total + 8) * sizeof(uint32_t))) != hipSuccess) return oom(e, "hipMalloc");
        c->plEntryCap = (size_t)total + 8;
    }
    // (the kernel fetches up to three slots behind the end of a list: spare words, slot 0)
    if ((e = hipMemsetAsync(c->dPlEntries + total, 0, 8 * sizeof(uint32_t), stream)) != hipSuccess) return oom(e, "hipMemsetAsync");
    if ((e = parity_lists_fill(scene_tripos(c), T, R, counts, offsets, sums, c->dPlCells, c->dPlEntries, stream)) != hipSuccess)
        return oom(e, "parity_lists_fill");
    if (t1) (void)hipEventRecord(t1, stream);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return oom(e, "hipStreamSynchronize");
    if (t0 && t1) c->plMs = elapsed(t0, t1);
    c->plEntries = (uint32_t)total;
    c->plRes = R;
    return done(1);
}

// relaunch: the same launch again with a deeper column (sync_frame, after a walk reported an overflow) -- possibly on behalf of
// a caller that is about to replace the scene (sync_frames): it builds nothing, it takes the candidate structures that exist.
int launch_now(dxv_ctx* c, uint32_t frame, bool relaunch = false)
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
    uint32_t cap = 0;                                                   // words per XCD queue of this partition
    f.list_entries = 0; f.list_res = 0;
    f.usedLists = false;
    // The lists cost 0.3-2.7 ms to build: a scene pays for them on its second launch (lists=1), so a
    // mesh that is refitted every frame and voxelized once per refit stays on the tree walk; lists=2
    // builds them at the first launch.
    // (... unless the launch is large enough for the build to pay for itself at once: build_lists decides after its
    // counting pass -- 1 M triangles at 512^3: 0.65 ms of build + 1.0 ms against 2.7 ms through the tree)
    const uint64_t voxels = (uint64_t)p.N * p.N * p.nz;
    if (!relaunch && p.mode == DXV_MODE_REFERENCE && c->optLists == 1 && c->launchesOfScene == 0 && c->listState == 0 && voxels >= (1ull << 26)) {
        if (sync_frames(c)) return 1;
        if (build_lists(c, fs, voxels, true)) return 1;
    }
    // A scene that has lists on the 256 map (a first launch that had to pay for its build at once) and is now launched AGAIN without
    // having been refitted: a static scene -- once, the 512 map instead (faster at every grid size, build_lists_into).  Deep scenes
    // fall back to their coarse map inside build_lists.
    if (!relaunch && p.mode == DXV_MODE_REFERENCE && c->optLists && !c->optListRes && c->listState == 1 && c->listRes < 512u && !c->refitted &&
        !c->listFloorTried && c->launchesOfScene > 0 && c->hdr.numTris >= 20000u &&
        (double)c->listEntries <= 32.0 * 6.0 * (double)c->listRes * (double)c->listRes) {       // (deep scenes keep their coarse map: build_lists_into)
        if (sync_frames(c)) return 1;
        c->listResFloor = 512u; c->listFloorTried = true;
        if (build_lists(c, fs)) return 1;
    }
    const bool wantLists = p.mode == DXV_MODE_REFERENCE && c->optLists &&
                           (relaunch ? c->listState == 1 && c->listOpt == c->optListRes : (c->optLists == 2 || c->launchesOfScene > 0 || c->listState != 0));
    if (p.mode == DXV_MODE_REFERENCE && !relaunch) ++c->launchesOfScene;
    if (wantLists) {
        if (c->listState == 0 || (c->listState != 0 && c->listOpt != c->optListRes)) {
            if (build_lists(c, fs, 0, true)) return 1;              // (this launch queues behind the build; its verdict: sync_frame)
        }
        if (c->listState == 1) {
            // lists built on another frame's stream whose end nobody has waited for yet: this stream waits for it on the device
            if (c->listCheckPending && c->listCheckStream != fs) DXV_HIP(c, hipStreamWaitEvent(fs, c->evList[3], 0));
            f.usedLists = true; f.listEpochUsed = c->listEpoch;
            p.lists = 1u;
            p.ablate = (uint32_t)c->optAblate;
            p.scene.dmCells = c->dListCells; p.scene.dmEntries = c->dListEntries; p.scene.dmR = c->listRes;
            st = 16;                                                // no stack: the column is the queue of selected triangles (8 items of two words)
            if (c->optRegion == 6) p.regionBits = 9u;                  // larger XCD regions suit the lists (-4 %); an explicit option wins
            f.list_entries = c->listEntries; f.list_res = c->listRes;
            if (c->optBrick == 4 && !c->optAblate && c->optPlan && c->dMip) {
                // the frame's work queue: sized for the partition (worst case: every brick live)
                const size_t words = plan_queue_words(p.N, p.nz, &cap);
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
                if (f.dQueue) {
                    queued = true; p.queue = f.dQueue + f.queueHdr * kQueueHeaderWords; p.queueSlots = f.dQueue + kQueueSlotsAt; p.queueCap = cap;
                    p.mip = c->dMip; p.queueWaves = (uint32_t)c->optQueueWaves; p.queueHeads = (uint32_t)c->optQueueHeads;
                    p.planRegionBits = c->optPlanRegion ? (uint32_t)c->optPlanRegion : plan_region_bits(p.N, p.nz);
                    p.planClear = c->optFuse ? 1u : 0u;
                    p.planHeavy = (uint32_t)c->optPlanHeavy;
                }
            }
        }
    }
    if (!queued) { f.plan_bricks = 0; f.plan_waves = 0; f.plan_ms = 0.0f; }
    f.lastQueued = false;
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
            const bool rebuild = c->optPlan == 2 || f.ptrExposed || f.clearSig != sig;
            hipEvent_t pe[2] = {f.evP0, f.evP1};
            // a queue launched again whose lengths an earlier dxv_sync has read: its size is known, the hardware can deal it out
            // (option dispatch: 1 = whenever known, 2 = for partitions of up to 2^25 voxels)
            const uint32_t* listed = nullptr;
            if (!rebuild && f.queueLenSig == sig && f.plan_bricks && (c->optDispatch == 1 || (c->optDispatch == 2 && voxels <= (1ull << 25)))) listed = f.queueLens;
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

} // namespace

extern "C" {

int dxv_create(dxv_ctx** out, int device)
{
    if (!out) return fail(nullptr, "dxv_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, "dxv_create: no HIP device available (%s); this library has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= n) return fail(nullptr, "dxv_create: device %d out of range [0,%d)", device, n);
    dxv_ctx* c = new dxv_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(nullptr, "dxv_create: cannot initialise device %d", device);
    }
    c->stream = c->ownStream;
    for (auto& ev : c->ev) {
        if (hipEventCreate(&ev) != hipSuccess) { delete c; return fail(nullptr, "dxv_create: hipEventCreate failed"); }
    }
    for (auto& ev : c->evList) {
        if (hipEventCreate(&ev) != hipSuccess) { dxv_destroy(c); return fail(nullptr, "dxv_create: hipEventCreate failed"); }
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&c->pin), sizeof(dxv_ctx::Pinned), hipHostMallocDefault) != hipSuccess) {
        c->pin = nullptr;
        dxv_destroy(c);
        return fail(nullptr, "dxv_create: hipHostMalloc failed");
    }
    memset(c->pin, 0, sizeof(dxv_ctx::Pinned));
    if (hipMalloc(&c->dCount, 256) != hipSuccess || hipMalloc(&c->dRootInfo, 256) != hipSuccess || frame_prepare(c, 0)) {
        dxv_destroy(c);
        return fail(nullptr, "dxv_create: hipMalloc failed");
    }
    *out = c;
    return 0;
}

void dxv_destroy(dxv_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (uint32_t i = 0; i < DXV_FRAME_COUNT; ++i) {
        Frame& f = c->frames[i];
        if (frame_stream(c, i)) (void)hipStreamSynchronize(frame_stream(c, i));
        (void)hipFree(f.dGrid); (void)hipFree(f.dTexels); (void)hipFree(f.dStatus); (void)hipFree(f.dRedo); (void)hipFree(f.dQueue);
        if (f.ev0) (void)hipEventDestroy(f.ev0);
        if (f.ev1) (void)hipEventDestroy(f.ev1);
        if (f.evP0) (void)hipEventDestroy(f.evP0);
        if (f.evP1) (void)hipEventDestroy(f.evP1);
        if (f.evEnd) (void)hipEventDestroy(f.evEnd);
        if (f.ownStream) (void)hipStreamDestroy(f.ownStream);
    }
    free_scratch(c);
    (void)hipFree(c->dMip);
    (void)hipFree(c->dVb); (void)hipFree(c->dIb); (void)hipFree(c->dScene);
    (void)hipFree(c->dImage); (void)hipFree(c->dEmpty); (void)hipFree(c->dListCells); (void)hipFree(c->dListEntries); (void)hipFree(c->dPlCells); (void)hipFree(c->dPlEntries); (void)hipFree(c->dPlScratch); (void)hipFree(c->dListScratchA); (void)hipFree(c->dListScratchB);
    (void)hipFree(c->dCount); (void)hipFree(c->dPacked); (void)hipFree(c->dRootInfo);
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : c->evList) if (ev) (void)hipEventDestroy(ev);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->copyStream) (void)hipStreamDestroy(c->copyStream);
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    delete c;
}

int dxv_api_version(void) { return DXV_API_VERSION; }

int dxv_trim(dxv_ctx* c)
{
    if (!c) return 1;
    DXV_HIP(c, hipSetDevice(c->device));
    if (sync_frames(c)) return 1;
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->dListScratchA); (void)hipFree(c->dListScratchB);
    c->dListScratchA = c->dListScratchB = nullptr; c->listScratchACap = c->listScratchBCap = 0;
    c->specRes = 0;
    if (!c->haveHierarchy) free_scratch(c);                             // (a built scene keeps keys and links: dxv_refit reads them)
    return 0;
}

const char* dxv_last_error(const dxv_ctx* c) { return c ? c->err.c_str() : g_createError.c_str(); }

int dxv_set_stream(dxv_ctx* c, void* s)
{
    if (!c) return 1;
    DXV_HIP(c, hipSetDevice(c->device));
    if (sync_frames(c)) return 1;
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = s ? static_cast<hipStream_t>(s) : c->ownStream;
    return 0;
}

int dxv_set_frame(dxv_ctx* c, uint32_t frame)
{
    if (!c) return 1;
    if (frame >= DXV_FRAME_COUNT) return fail(c, "dxv_set_frame: frame %u out of range [0, %d)", frame, DXV_FRAME_COUNT);
    DXV_HIP(c, hipSetDevice(c->device));
    if (frame_prepare(c, frame)) return 1;
    c->cur = frame;
    return 0;
}

int dxv_set_mesh(dxv_ctx* c, const float* vb, uint32_t V, const uint32_t* ib, uint32_t T)
{
    if (!c) return 1;
    if (!vb || !ib || !V || !T) return fail(c, "dxv_set_mesh: empty mesh (V=%u, T=%u)", V, T);
    if (T > 0x7fffffffu / 2) return fail(c, "dxv_set_mesh: too many triangles (%u)", T);
    for (size_t i = 0; i < 3 * (size_t)T; ++i)
        if (ib[i] >= V) return fail(c, "dxv_set_mesh: index %u at position %zu out of range (V=%u)", ib[i], i, V);
    // bound: AABB over every VB position (XUSGObjLoader.cpp:386-416), centre and half max extent
    // (Content/Voxelizer.cpp:52-57)
    float mn[3] = {vb[0], vb[1], vb[2]}, mx[3] = {vb[0], vb[1], vb[2]};
    for (uint32_t i = 0; i < V; ++i) {
        const float* p = vb + 6 * (size_t)i;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2]))   // (a NaN would slip through both comparisons below)
            return fail(c, "dxv_set_mesh: vertex %u has a non-finite position (%g, %g, %g)", i, (double)p[0], (double)p[1], (double)p[2]);
        for (int a = 0; a < 3; ++a) {
            if (p[a] < mn[a]) mn[a] = p[a];
            else if (p[a] > mx[a]) mx[a] = p[a];
        }
    }
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    c->bound[0] = (mx[0] + mn[0]) / 2.0f;
    c->bound[1] = (mx[1] + mn[1]) / 2.0f;
    c->bound[2] = (mx[2] + mn[2]) / 2.0f;
    const float eyz = ey > ez ? ey : ez;
    c->bound[3] = (ex > eyz ? ex : eyz) / 2.0f;
    if (!(c->bound[3] > 0.0f) || !std::isfinite(c->bound[3]) || !std::isfinite(c->bound[0]) ||
        !std::isfinite(c->bound[1]) || !std::isfinite(c->bound[2]))
        return fail(c, "dxv_set_mesh: degenerate or non-finite bound (half extent %g)", (double)c->bound[3]);

    DXV_HIP(c, hipSetDevice(c->device));
    if (sync_frames(c)) return 1;
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    c->vbCopyQueued = false;
    (void)hipFree(c->dVb); (void)hipFree(c->dIb);
    c->dVb = nullptr; c->dIb = nullptr;
    c->haveMesh = false; c->haveScene = false; c->haveHierarchy = false; c->listState = 0; c->specRes = 0; c->listResFloor = 0; c->listFloorTried = false; c->refitted = false; c->launchesOfScene = 0; c->plState = 0; c->parityLaunchesOfScene = 0; c->nodesStale = 0;
    DXV_HIP(c, hipMalloc(&c->dVb, sizeof(float) * 6 * (size_t)V));
    DXV_HIP(c, hipMalloc(&c->dIb, sizeof(uint32_t) * 3 * (size_t)T));
    DXV_HIP(c, hipEventRecord(c->ev[8], c->stream));
    DXV_HIP(c, hipMemcpyAsync(c->dVb, vb, sizeof(float) * 6 * (size_t)V, hipMemcpyHostToDevice, c->stream));
    DXV_HIP(c, hipMemcpyAsync(c->dIb, ib, sizeof(uint32_t) * 3 * (size_t)T, hipMemcpyHostToDevice, c->stream));
    DXV_HIP(c, hipEventRecord(c->ev[9], c->stream));
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    c->T = T; c->V = V;
    c->haveMesh = true;
    c->stats.num_tris = T; c->stats.num_verts = V;
    memcpy(c->stats.bound, c->bound, sizeof(c->bound));
    c->stats.upload_ms = elapsed(c->ev[8], c->ev[9]);
    return 0;
}

namespace {
void fill_build_buffers(dxv_ctx* c, BuildBuffers& b)
{
    b.vb = c->dVb; b.ib = c->dIb; b.T = c->T; b.V = c->V;
    memcpy(b.bound, c->bound, sizeof(c->bound));
    b.keys = c->dKeys; b.keysTmp = c->dKeysTmp; b.hist = c->dHist; b.parents = c->dParents;
    b.flags = c->dFlags; b.flags2 = c->dFlags2; b.rootInfo = c->dRootInfo; b.pyramid = c->dPyramid;
    b.nodes = scene_nodes(c); b.nodes32 = scene_nodes32(c); b.nodes64 = c->hdr.hasWide ? scene_nodes64(c) : nullptr; b.triPos = scene_tripos(c); b.triNrm = scene_trinrm(c);
}

// The half-float / four-box copies of the hierarchy after a refit that skipped them (dxv_refit): made now, on `stream`, and
// finished before anyone else can launch a walk on another stream.
int ensure_nodes(dxv_ctx* c, hipStream_t stream)
{
    if (!c->nodesStale) return 0;
    BuildBuffers b{};
    fill_build_buffers(c, b);
    if (c->nodesStale == 2) DXV_HIP(c, lbvh_refit_boxes(b, stream));
    else DXV_HIP(c, lbvh_traversal_copies(b, stream));
    DXV_HIP(c, hipStreamSynchronize(stream));
    c->nodesStale = 0;
    return 0;
}

// min/max pyramid of the box merge (refit = 1): 24 B box + 4 B deepest leaf per slot
int alloc_pyramid(dxv_ctx* c)
{
    if (!c->dPyramid && c->optRefit == 1 && c->T > 1)      // refit=2 keeps the level sweeps, refit=0 the atomic pass
        DXV_HIP(c, hipMalloc(&c->dPyramid, 28 * (size_t)pyramid_slots(c->T)));
    return 0;
}

// headerToDevice: the resident copy of the header (nothing on the device reads it -- launches get their words through kernel
// arguments, an export writes the blob's header from the host's copy): dxv_build keeps it current, a refit saves the round trip
int finish_build(dxv_ctx* c, const char* who, bool headerToDevice = true)
{
    uint32_t* rootInfo = c->pin->rootInfo;
    DXV_HIP(c, hipMemcpyAsync(rootInfo, c->dRootInfo, sizeof(c->pin->rootInfo), hipMemcpyDeviceToHost, c->stream));
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    c->vbCopyQueued = false;
    if (rootInfo[7] != 1) return fail(c, "%s: did not complete", who);
    if (rootInfo[11]) return fail(c, "%s: %u triangle(s) have a non-finite vertex position (NaN / Inf in the vertex buffer)", who, rootInfo[11]);
    memcpy(c->hdr.rootLo, &rootInfo[0], 12);
    memcpy(c->hdr.rootHi, &rootInfo[3], 12);
    c->hdr.treeHeight = rootInfo[6];
    if (rootInfo[10]) {   // k_tri_keys: sum over the sampled triangles of (extent y + extent z) in 2^-20 units, and their number
        unsigned long long ext;
        memcpy(&ext, &rootInfo[8], sizeof(ext));
        c->hdr.triExtent = (float)((double)ext / 1048576.0 / 2.0 / (double)rootInfo[10]);
    }                     // (a refit keeps the figure of the build)
    for (int a = 0; a < 3; ++a)
        if (!(c->hdr.rootLo[a] <= c->hdr.rootHi[a]))
            return fail(c, "%s: refit produced an invalid root box (axis %d: %g > %g)", who, a,
                        (double)c->hdr.rootLo[a], (double)c->hdr.rootHi[a]);
    if (c->hdr.treeHeight == 0 || c->hdr.treeHeight > 64) return fail(c, "%s: implausible tree height %u", who, c->hdr.treeHeight);
    if (headerToDevice) {
        DXV_HIP(c, hipMemcpyAsync(c->dScene, &c->hdr, sizeof(SceneHeader), hipMemcpyHostToDevice, c->stream));
        DXV_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->haveScene = true;
    c->stackNow = stack_round_up((int)(c->hdr.treeHeight + 3 < (uint32_t)c->optStack0 ? c->hdr.treeHeight + 3 : (uint32_t)c->optStack0));
    c->stats.num_nodes = c->hdr.numNodes;
    c->stats.tree_height = c->hdr.treeHeight;
    c->stats.tri_extent = c->hdr.triExtent;
    return 0;
}
} // namespace

int dxv_update_vertices(dxv_ctx* c, const float* vb, uint32_t V)
{
    if (!c) return 1;
    if (!c->haveMesh || !c->dVb) return fail(c, "dxv_update_vertices: no mesh resident on this context");
    if (!vb || V != c->V) return fail(c, "dxv_update_vertices: vertex count must stay %u, got %u", c->V, V);
    DXV_HIP(c, hipSetDevice(c->device));
    // No wait for the frames: their launches read the scene's triangle records and lists, never the vertex buffer -- its only
    // readers are dxv_build and dxv_refit, which return after their work is done.  The upload therefore runs on a stream of its
    // own, beside whatever the frames still have in flight (12 MB over PCIe at 1 M triangles: 0.25 ms hidden behind a launch).
    if (!c->copyStream) DXV_HIP(c, hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
    if (c->vbCopyQueued) { DXV_HIP(c, hipStreamSynchronize(c->stream)); c->vbCopyQueued = false; }    // (an earlier device update lands first)
    DXV_HIP(c, hipMemcpyAsync(c->dVb, vb, sizeof(float) * 6 * (size_t)V, hipMemcpyHostToDevice, c->copyStream));
    DXV_HIP(c, hipStreamSynchronize(c->copyStream));
    return 0;
}

int dxv_update_vertices_device(dxv_ctx* c, const void* dvb, uint32_t V)
{
    if (!c) return 1;
    if (!c->haveMesh || !c->dVb) return fail(c, "dxv_update_vertices_device: no mesh resident on this context");
    if (!dvb || V != c->V) return fail(c, "dxv_update_vertices_device: vertex count must stay %u, got %u", c->V, V);
    DXV_HIP(c, hipSetDevice(c->device));
    // No wait for the frames (as in dxv_update_vertices): their launches never read the vertex buffer, and its readers --
    // dxv_build, dxv_refit -- run on this same stream.
    DXV_HIP(c, hipMemcpyAsync(c->dVb, dvb, sizeof(float) * 6 * (size_t)V, hipMemcpyDeviceToDevice, c->stream));
    c->vbCopyQueued = true;
    return 0;                                                          // (dxv_refit, on the same stream, comes next)
}

int dxv_refit(dxv_ctx* c)
{
    if (!c) return 1;
    if (!c->haveMesh || !c->haveHierarchy || c->scratchT != c->T || !c->T)
        return fail(c, "dxv_refit: needs a scene built on this context by dxv_build (imported scenes carry no build state)");
    DXV_HIP(c, hipSetDevice(c->device));
    // The frames' launches read what the refit is about to write.  A launch that may still have something to say (a tree walk whose
    // column can run out; lists that failed their deferred check) is synchronised on the host and, if need be, run again against
    // the OLD scene; every other launch is waited for ON THE DEVICE -- frame 0 shares this stream, the other frames' end events
    // are waited for by it -- so that the refit's kernels, the lists' counting pass and the host's one synchronisation of the
    // frame (finish_build) queue up behind a launch that is still running.
    if (settle_lists(c)) return 1;                                     // (waits for a build's end, not for the launch behind it)
    for (uint32_t i = 0; i < DXV_FRAME_COUNT; ++i) {
        Frame& f = c->frames[i];
        if (!f.ready || !f.pending) continue;
        if (f.lastCanFail || (f.usedLists && f.listEpochUsed == c->withdrawnEpoch)) { if (sync_frame(c, i)) return 1; }
        else if (frame_stream(c, i) != c->stream) DXV_HIP(c, hipStreamWaitEvent(c->stream, f.evEnd, 0));
    }
    const uint32_t hadListsOn = c->listState == 1 ? c->listRes : 0u;
    c->haveScene = false; c->listState = 0; c->listResFloor = 0; c->listFloorTried = false; c->launchesOfScene = 0; c->plState = 0; c->parityLaunchesOfScene = 0;
    c->refitted = true;
    c->specRes = 0;
    if (alloc_pyramid(c)) return 1;
    BuildBuffers b{};
    fill_build_buffers(c, b);
    if (c->optRefit != 1) b.pyramid = nullptr;
    // the wide walks' four-box copy of the hierarchy waits until a walk needs it: the next launch of a refitted mesh usually
    // goes through the lists, which are built from the triangle records alone (the half-float copy comes out of the box
    // merge's registers and is always current)
    b.deferCopies = c->optLists != 0 && c->hdr.hasWide;
    // ... and so do the node boxes themselves (half of the refit's time at 1 M triangles): the refit stops at the min/max
    // pyramid, whose top is the root box the launch needs, and ensure_nodes finishes it in front of the first tree walk
    b.deferBoxes = c->optLists != 0 && c->optDeferBoxes && b.pyramid && c->T > 1;
    DXV_HIP(c, lbvh_refit(b, c->optRefit, c->hdr.treeHeight, c->stream, c->ev + 3));
    c->nodesStale = b.deferBoxes ? 2 : b.deferCopies ? 1 : 0;
    // A scene that had lists (or asks for them from its first launch) will have them rebuilt by its next launch: their counting
    // pass needs the new triangle records only, so it runs here, behind the refit, and its total comes back with the root box.
    uint32_t spec = 0;
    if (c->optLists && (hadListsOn || c->optLists == 2) && c->dListScratchA && c->listScratchACap >= list_scratch_a(nullptr, c->hdr.numTris).bytes) {
        spec = c->optListRes ? (uint32_t)c->optListRes : list_resolution(c);     // (the base map: a mesh that is being refitted gets its lists built for one launch)
        const ListScratchA sa = list_scratch_a(c->dListScratchA, c->hdr.numTris);
        DXV_HIP(c, hipEventRecord(c->evList[0], c->stream));
        DXV_HIP(c, dirmap_count(scene_tripos(c), c->hdr.numTris, spec, sa.rec, sa.counts, sa.pairs, sa.total, c->stream));
        DXV_HIP(c, hipEventRecord(c->evList[1], c->stream));
        DXV_HIP(c, hipMemcpyAsync(&c->pin->listTotal, sa.total, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    }
    if (finish_build(c, "dxv_refit", false)) return 1;
    c->specRes = spec;
    c->stats.refit_ms = elapsed(c->ev[3], c->ev[4]);
    return 0;
}

int dxv_build(dxv_ctx* c)
{
    if (!c) return 1;
    if (!c->haveMesh) return fail(c, "dxv_build: no mesh (call dxv_set_mesh first)");
    DXV_HIP(c, hipSetDevice(c->device));
    if (sync_frames(c)) return 1;
    c->haveScene = false; c->haveHierarchy = false; c->listState = 0; c->specRes = 0; c->listResFloor = 0; c->listFloorTried = false; c->refitted = false; c->launchesOfScene = 0; c->plState = 0; c->parityLaunchesOfScene = 0; c->nodesStale = 0;
    if (alloc_scene(c, c->T, c->V, c->optWide != 0)) return 1;
    if (alloc_scratch(c, c->T)) return 1;
    if (alloc_pyramid(c)) return 1;
    memcpy(c->hdr.bound, c->bound, sizeof(c->bound));

    BuildBuffers b{};
    fill_build_buffers(c, b);
    if (c->optRefit != 1) b.pyramid = nullptr;
    b.deferCopies = c->optLists != 0 && c->hdr.hasWide;                // (as in dxv_refit: 0.6 ms of a 10 M-triangle build that most scenes never need)
    DXV_HIP(c, lbvh_build(b, c->optRefit, c->stream, c->ev));
    c->nodesStale = b.deferCopies ? 1 : 0;
    if (finish_build(c, "dxv_build")) return 1;
    c->haveHierarchy = true;
    c->stats.prep_ms = elapsed(c->ev[0], c->ev[1]);
    c->stats.sort_ms = elapsed(c->ev[1], c->ev[2]);
    c->stats.hierarchy_ms = elapsed(c->ev[2], c->ev[3]);
    c->stats.refit_ms = elapsed(c->ev[3], c->ev[4]);
    c->stats.build_ms = elapsed(c->ev[0], c->ev[4]);
    return 0;
}

namespace {
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
} // namespace

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

int dxv_render(dxv_ctx* c, const float eye[3], const float viewProj[16], const float posScale[4], uint32_t width,
               uint32_t height, uint8_t* rgbaHost)
{
    if (!c) return 1;
    Frame& f = cur_frame(c);
    const hipStream_t fs = cur_stream(c);
    (void)fs;
    if (!eye || !viewProj || !rgbaHost || !width || !height || width > 16384 || height > 16384)
        return fail(c, "dxv_render: bad arguments");
    const uint32_t N = f.grid_dim;
    if (!f.dGrid || !N || f.z0 != 0 || f.nz != N || f.lastZBlock != N)
        return fail(c, "dxv_render: needs the whole grid of the last dxv_voxelize (z0 = 0, nz = grid_dim) on this context");
    const float unit[4] = {0.0f, 0.0f, 0.0f, 1.0f};                 // DXRVoxelizer.cpp:37
    RayCastCB cb;
    if (!update_frame(c->bound, posScale ? posScale : unit, eye, viewProj, (float)width, (float)height, cb))
        return fail(c, "dxv_render: singular view/projection chain");
    DXV_HIP(c, hipSetDevice(c->device));
    const size_t pixels = (size_t)width * height;
    if (pixels > c->imageCap) {
        DXV_HIP(c, hipStreamSynchronize(fs));
        (void)hipFree(c->dImage); c->dImage = nullptr; c->imageCap = 0;
        DXV_HIP(c, hipMalloc(&c->dImage, pixels * 4));
        c->imageCap = pixels;
    }
    if (c->optSkipEmpty && empty_brick_bytes(N) > c->emptyCap) {
        DXV_HIP(c, hipStreamSynchronize(fs));
        (void)hipFree(c->dEmpty); c->dEmpty = nullptr; c->emptyCap = 0;
        DXV_HIP(c, hipMalloc(&c->dEmpty, align256(empty_brick_bytes(N))));
        c->emptyCap = empty_brick_bytes(N);
    }
    if (dxv_sync(c)) return 1;                                       // the grid must be complete and valid
    DXV_HIP(c, hipEventRecord(c->ev[8], fs));
    DXV_HIP(c, launch_raycast(cb, f.dGrid, N, width, height, c->dImage, c->optSkipEmpty ? c->dEmpty : nullptr, fs));
    DXV_HIP(c, hipEventRecord(c->ev[9], fs));
    DXV_HIP(c, hipMemcpyAsync(rgbaHost, c->dImage, pixels * 4, hipMemcpyDeviceToHost, fs));
    DXV_HIP(c, hipStreamSynchronize(fs));
    c->renderMs = elapsed(c->ev[8], c->ev[9]);
    c->stats.render_ms = c->renderMs;
    return 0;
}

void* dxv_grid_device_ptr(dxv_ctx* c)
{
    if (!c) return nullptr;
    // the caller may write through the pointer now or at any later time (it stays valid until the grid is reallocated): from
    // here on every launch into this frame clears the grid itself instead of trusting what it wrote there last
    cur_frame(c).clearSig = 0;
    cur_frame(c).ptrExposed = true;
    return cur_frame(c).dGrid;
}
const void* dxv_grid_device_ptr_ro(const dxv_ctx* c) { return c ? c->frames[c->cur].dGrid : nullptr; }
size_t dxv_grid_bytes(const dxv_ctx* c) { return c ? c->frames[c->cur].gridBytes : 0; }

int dxv_grid_download(dxv_ctx* c, uint8_t* host, size_t bytes)
{
    if (!c) return 1;
    Frame& f = cur_frame(c);
    const hipStream_t fs = cur_stream(c);
    (void)fs;
    if (!host || bytes != f.gridBytes || !f.gridBytes) return fail(c, "dxv_grid_download: expected %zu bytes, got %zu", f.gridBytes, bytes);
    if (f.pending && dxv_sync(c)) return 1;          // an unchecked launch: finish it (and its redo, if any) first
    DXV_HIP(c, hipSetDevice(c->device));
    DXV_HIP(c, hipMemcpyAsync(host, f.dGrid, bytes, hipMemcpyDeviceToHost, fs));
    DXV_HIP(c, hipStreamSynchronize(fs));
    return 0;
}

size_t dxv_grid_packed_bytes(const dxv_ctx* c) { return c ? (c->frames[c->cur].gridBytes + 7) / 8 : 0; }

int dxv_grid_download_packed(dxv_ctx* c, uint8_t* host, size_t bytes)
{
    if (!c) return 1;
    Frame& f = cur_frame(c);
    const hipStream_t fs = cur_stream(c);
    (void)fs;
    const size_t want = (f.gridBytes + 7) / 8;
    if (!host || !want || bytes != want) return fail(c, "dxv_grid_download_packed: expected %zu bytes, got %zu", want, bytes);
    if (f.pending && dxv_sync(c)) return 1;          // an unchecked launch: finish it (and its redo, if any) first
    DXV_HIP(c, hipSetDevice(c->device));
    if (want > c->packedCap) {
        DXV_HIP(c, hipStreamSynchronize(fs));
        (void)hipFree(c->dPacked); c->dPacked = nullptr; c->packedCap = 0;
        DXV_HIP(c, hipMalloc(&c->dPacked, align256(want)));
        c->packedCap = want;
    }
    DXV_HIP(c, launch_pack_bits(f.dGrid, f.gridBytes, c->dPacked, fs));
    DXV_HIP(c, hipMemcpyAsync(host, c->dPacked, want, hipMemcpyDeviceToHost, fs));
    DXV_HIP(c, hipStreamSynchronize(fs));
    return 0;
}

int dxv_grid_count(dxv_ctx* c, uint64_t* solid)
{
    if (!c) return 1;
    Frame& f = cur_frame(c);
    const hipStream_t fs = cur_stream(c);
    (void)fs;
    if (!solid || !f.gridBytes) return fail(c, "dxv_grid_count: no grid");
    if (f.pending && dxv_sync(c)) return 1;          // an unchecked launch: finish it (and its redo, if any) first
    DXV_HIP(c, hipSetDevice(c->device));
    DXV_HIP(c, launch_count(f.dGrid, f.gridBytes, c->dCount, fs));
    unsigned long long v = 0;
    DXV_HIP(c, hipMemcpyAsync(&v, c->dCount, sizeof(v), hipMemcpyDeviceToHost, fs));
    DXV_HIP(c, hipStreamSynchronize(fs));
    *solid = v;
    return 0;
}

int dxv_enable_texels(dxv_ctx* c, int enable)
{
    if (!c) return 1;
    c->texels = enable != 0;
    return 0;
}

int dxv_texels_download(dxv_ctx* c, uint32_t* host, size_t bytes)
{
    if (!c) return 1;
    Frame& f = cur_frame(c);
    const hipStream_t fs = cur_stream(c);
    (void)fs;
    if (!c->texels || !f.dTexels) return fail(c, "dxv_texels_download: texel output not enabled");
    if (!host || bytes != f.gridBytes * 4) return fail(c, "dxv_texels_download: expected %zu bytes, got %zu", f.gridBytes * 4, bytes);
    if (f.pending && dxv_sync(c)) return 1;          // an unchecked launch: finish it (and its redo, if any) first
    DXV_HIP(c, hipSetDevice(c->device));
    DXV_HIP(c, hipMemcpyAsync(host, f.dTexels, bytes, hipMemcpyDeviceToHost, fs));
    DXV_HIP(c, hipStreamSynchronize(fs));
    return 0;
}

namespace {
// an exported blob = the scene as it lies in memory + (when the context has built them) the lists' two arrays
struct BlobLayout { size_t offCells, cellBytes, offEntries, entryBytes, offPlCells, plCellBytes, offPlEntries, plEntryBytes, total; };
BlobLayout blob_layout(size_t sceneBytes, uint32_t listRes, uint32_t listCount, uint32_t plRes, uint32_t plCount)
{
    BlobLayout b{0, 0, 0, 0, 0, 0, 0, 0, sceneBytes};
    if (listRes) {
        b.offCells = align256(b.total);
        b.cellBytes = sizeof(DirCell) * 6 * (size_t)listRes * listRes;
        b.offEntries = align256(b.offCells + b.cellBytes);
        b.entryBytes = sizeof(DirEntry) * (size_t)listCount;
        b.total = align256(b.offEntries + b.entryBytes);
    }
    if (plRes) {
        b.offPlCells = align256(b.total);
        b.plCellBytes = sizeof(uint32_t) * 2 * (size_t)plRes * plRes;
        b.offPlEntries = align256(b.offPlCells + b.plCellBytes);
        b.plEntryBytes = sizeof(uint32_t) * (size_t)plCount;
        b.total = align256(b.offPlEntries + b.plEntryBytes);
    }
    return b;
}
bool lists_exportable(const dxv_ctx* c) { return c->listState == 1 && c->listOpt == c->optListRes; }
bool plists_exportable(const dxv_ctx* c) { return c->plState == 1 && (c->optPlistRes == 0 || (uint32_t)c->optPlistRes == c->plRes); }
BlobLayout export_layout(const dxv_ctx* c)
{
    const bool l = lists_exportable(c), pl = plists_exportable(c);
    return blob_layout(c->sceneBytes, l ? c->listRes : 0u, l ? c->listEntries : 0u, pl ? c->plRes : 0u, pl ? c->plEntries : 0u);
}
} // namespace

size_t dxv_scene_bytes(const dxv_ctx* c)
{
    if (!c || !c->haveScene) return 0;
    return export_layout(c).total;
}

int dxv_build_parity_lists(dxv_ctx* c)
{
    if (!c) return 1;
    if (!c->haveScene) return fail(c, "dxv_build_parity_lists: no scene");
    DXV_HIP(c, hipSetDevice(c->device));
    if (c->plState != 0) return 0;
    if (sync_frames(c)) return 1;
    return build_plists(c, c->stream);
}

int dxv_build_lists(dxv_ctx* c)
{
    if (!c) return 1;
    if (!c->haveScene) return fail(c, "dxv_build_lists: no scene");
    DXV_HIP(c, hipSetDevice(c->device));
    if (settle_lists(c)) return 1;
    if (c->listState != 0 && c->listOpt == c->optListRes) return 0;
    if (sync_frames(c)) return 1;
    return build_lists(c, c->stream);
}

int dxv_build_lists_for_grid(dxv_ctx* c, uint32_t N)
{
    if (!c) return 1;
    if (!c->haveScene) return fail(c, "dxv_build_lists_for_grid: no scene");
    DXV_HIP(c, hipSetDevice(c->device));
    // the map the launches of a static scene move to (launch_now: the 512 map, at every grid size)
    (void)N;
    if (!c->optListRes && c->hdr.numTris >= 20000u && !c->refitted && !c->listFloorTried && c->listResFloor < 512u) {
        if (sync_frames(c)) return 1;
        c->listResFloor = 512u; c->listFloorTried = true;
        if (c->listState == 1 && c->listRes >= 512u) return 0;
        return build_lists(c, c->stream);
    }
    return dxv_build_lists(c);
}

int dxv_scene_export(dxv_ctx* c, void* dst, size_t bytes)
{
    if (!c) return 1;
    if (!c->haveScene) return fail(c, "dxv_scene_export: no scene");
    if (settle_lists(c)) return 1;
    const bool withLists = lists_exportable(c), withPl = plists_exportable(c);
    const BlobLayout b = export_layout(c);
    if (!dst || bytes != b.total) return fail(c, "dxv_scene_export: expected %zu bytes, got %zu", b.total, bytes);
    DXV_HIP(c, hipSetDevice(c->device));
    if (ensure_nodes(c, c->stream)) return 1;
    DXV_HIP(c, hipMemcpyAsync(dst, c->dScene, c->sceneBytes, hipMemcpyDeviceToDevice, c->stream));
    SceneHeader h = c->hdr;
    h.offListCells = h.offListEntries = 0; h.listRes = h.listCount = 0;
    h.offPlCells = h.offPlEntries = 0; h.plRes = h.plCount = 0;
    uint8_t* out = static_cast<uint8_t*>(dst);
    if (withLists) {
        DXV_HIP(c, hipMemcpyAsync(out + b.offCells, c->dListCells, b.cellBytes, hipMemcpyDeviceToDevice, c->stream));
        if (b.entryBytes) DXV_HIP(c, hipMemcpyAsync(out + b.offEntries, c->dListEntries, b.entryBytes, hipMemcpyDeviceToDevice, c->stream));
        h.offListCells = b.offCells; h.offListEntries = b.offEntries; h.listRes = c->listRes; h.listCount = c->listEntries;
    }
    if (withPl) {
        DXV_HIP(c, hipMemcpyAsync(out + b.offPlCells, c->dPlCells, b.plCellBytes, hipMemcpyDeviceToDevice, c->stream));
        if (b.plEntryBytes) DXV_HIP(c, hipMemcpyAsync(out + b.offPlEntries, c->dPlEntries, b.plEntryBytes, hipMemcpyDeviceToDevice, c->stream));
        h.offPlCells = b.offPlCells; h.offPlEntries = b.offPlEntries; h.plRes = c->plRes; h.plCount = c->plEntries;
    }
    h.totalBytes = b.total;
    DXV_HIP(c, hipMemcpyAsync(dst, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));     // the blob's own header (the resident one keeps the scene's size)
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int dxv_scene_checksum(dxv_ctx* c, const void* device_blob, size_t bytes, uint64_t* sum)
{
    if (!c || !sum) return 1;
    if (!device_blob || bytes < 8) return fail(c, "dxv_scene_checksum: no blob");
    DXV_HIP(c, hipSetDevice(c->device));
    DXV_HIP(c, launch_checksum(device_blob, bytes, c->dCount, c->stream));
    unsigned long long v = 0;
    DXV_HIP(c, hipMemcpyAsync(&v, c->dCount, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    *sum = v;
    return 0;
}

int dxv_scene_import(dxv_ctx* c, const void* src, size_t bytes)
{
    if (!c) return 1;
    if (!src || bytes < sizeof(SceneHeader)) return fail(c, "dxv_scene_import: blob too small (%zu bytes)", bytes);
    DXV_HIP(c, hipSetDevice(c->device));
    SceneHeader h;
    DXV_HIP(c, hipMemcpyAsync(&h, src, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    if (h.magic != kSceneMagic || h.version != kSceneVersion) return fail(c, "dxv_scene_import: bad magic/version");
    SceneHeader want;
    layout_scene(want, h.numTris, h.numVerts, h.hasWide != 0);
    const bool withLists = h.listRes != 0;
    if (withLists && (h.listRes < 16u || h.listRes > 4096u || (h.listRes & (h.listRes - 1u)) || h.listCount > 0x7fffffffu))
        return fail(c, "dxv_scene_import: inconsistent list section (res=%u, entries=%u)", h.listRes, h.listCount);
    const bool withPl = h.plRes != 0;
    if (withPl && (h.plRes < 16u || h.plRes > 4096u || (h.plRes & (h.plRes - 1u)) || h.plCount > 0x7ffffff0u))
        return fail(c, "dxv_scene_import: inconsistent row-list section (res=%u, entries=%u)", h.plRes, h.plCount);
    const BlobLayout b = blob_layout(want.totalBytes, withLists ? h.listRes : 0u, withLists ? h.listCount : 0u, withPl ? h.plRes : 0u, withPl ? h.plCount : 0u);
    if (!h.numTris || b.total != bytes || h.totalBytes != bytes || h.offNodes != want.offNodes ||
        h.offTriPos != want.offTriPos || h.offTriNrm != want.offTriNrm || h.offNodes32 != want.offNodes32 || h.offNodes64 != want.offNodes64 || h.hasWide > 1u || h.treeHeight == 0 || h.treeHeight > 64 ||
        (withLists && (h.offListCells != b.offCells || h.offListEntries != b.offEntries)) || (!withLists && (h.offListCells || h.offListEntries || h.listCount)) ||
        (withPl && (h.offPlCells != b.offPlCells || h.offPlEntries != b.offPlEntries)) || (!withPl && (h.offPlCells || h.offPlEntries || h.plCount)))
        return fail(c, "dxv_scene_import: inconsistent header (T=%u, bytes=%zu)", h.numTris, bytes);
    if (sync_frames(c)) return 1;
    c->haveScene = false; c->listState = 0; c->specRes = 0; c->listResFloor = 0; c->listFloorTried = false; c->refitted = false; c->launchesOfScene = 0; c->plState = 0; c->parityLaunchesOfScene = 0; c->nodesStale = 0;
    // An imported scene carries no mesh and no build state: drop what an earlier dxv_set_mesh / dxv_build left on this
    // context, so that dxv_build, dxv_refit and dxv_update_vertices fail cleanly instead of running the imported
    // triangle count over the old, smaller buffers.
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->dVb); (void)hipFree(c->dIb);
    c->dVb = nullptr; c->dIb = nullptr; c->haveMesh = false; c->haveHierarchy = false;
    free_scratch(c);
    if (alloc_scene(c, h.numTris, h.numVerts, h.hasWide != 0)) return 1;
    DXV_HIP(c, hipMemcpyAsync(c->dScene, src, want.totalBytes, hipMemcpyDeviceToDevice, c->stream));
    if (withLists) {
        // the lists travel with the scene: adopt them instead of building them again (1-5 ms per rank at 1 M triangles)
        const size_t cells = 6 * (size_t)h.listRes * h.listRes;
        if (cells > c->listCellCap) {
            (void)hipFree(c->dListCells); c->dListCells = nullptr; c->listCellCap = 0;
            DXV_HIP(c, hipMalloc(&c->dListCells, cells * sizeof(DirCell)));
            c->listCellCap = cells;
        }
        if ((size_t)h.listCount > c->listEntryCap) {
            (void)hipFree(c->dListEntries); c->dListEntries = nullptr; c->listEntryCap = 0;
            DXV_HIP(c, hipMalloc(&c->dListEntries, ((size_t)h.listCount + 4) * sizeof(DirEntry)));
            c->listEntryCap = h.listCount;
        }
        const uint8_t* in = static_cast<const uint8_t*>(src);
        DXV_HIP(c, hipMemcpyAsync(c->dListCells, in + b.offCells, b.cellBytes, hipMemcpyDeviceToDevice, c->stream));
        if (b.entryBytes) DXV_HIP(c, hipMemcpyAsync(c->dListEntries, in + b.offEntries, b.entryBytes, hipMemcpyDeviceToDevice, c->stream));
        // The kernel indexes the entries with what the cells say and the triangle records with what the entries say: a blob
        // whose header is consistent but whose payload is not (cut short, corrupted, another version's) must not get that far.
        uint32_t bad[2] = {0, 0};
        DXV_HIP(c, dirmap_validate(c->dListCells, h.listRes, c->dListEntries, h.listCount, h.numTris, c->dRootInfo, c->stream));
        DXV_HIP(c, hipMemcpyAsync(bad, c->dRootInfo, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
        DXV_HIP(c, hipStreamSynchronize(c->stream));
        if (bad[0] || bad[1])
            return fail(c, "dxv_scene_import: the list section is inconsistent (%u texels point outside the %u entries, %u entries name a triangle >= %u)",
                        bad[0], h.listCount, bad[1], h.numTris);
        // the max-mip of the far radii is a function of the cells: made here, not carried in the blob
        if (dm_mip_words(h.listRes) > c->mipCap) {
            (void)hipFree(c->dMip); c->dMip = nullptr; c->mipCap = 0;
            DXV_HIP(c, hipMalloc(&c->dMip, sizeof(uint16_t) * (size_t)dm_mip_buffer_words(h.listRes)));
            c->mipCap = dm_mip_words(h.listRes);
        }
        DXV_HIP(c, dirmap_mip(c->dListCells, h.listRes, c->dMip, c->stream));
        DXV_HIP(c, hipStreamSynchronize(c->stream));
    }
    if (withPl) {
        // ... and so do the row lists of the parity rule (1.6 ms per rank at 1 M triangles)
        const size_t cellWords = 2 * (size_t)h.plRes * h.plRes;
        if (cellWords > c->plCellCap) {
            (void)hipFree(c->dPlCells); c->dPlCells = nullptr; c->plCellCap = 0;
            DXV_HIP(c, hipMalloc(&c->dPlCells, cellWords * sizeof(uint32_t)));
            c->plCellCap = cellWords;
        }
        if ((size_t)h.plCount + 8 > c->plEntryCap) {
            (void)hipFree(c->dPlEntries); c->dPlEntries = nullptr; c->plEntryCap = 0;
            DXV_HIP(c, hipMalloc(&c->dPlEntries, ((size_t)h.plCount + 8) * sizeof(uint32_t)));
            c->plEntryCap = (size_t)h.plCount + 8;
        }
        const uint8_t* in = static_cast<const uint8_t*>(src);
        DXV_HIP(c, hipMemcpyAsync(c->dPlCells, in + b.offPlCells, b.plCellBytes, hipMemcpyDeviceToDevice, c->stream));
        if (b.plEntryBytes) DXV_HIP(c, hipMemcpyAsync(c->dPlEntries, in + b.offPlEntries, b.plEntryBytes, hipMemcpyDeviceToDevice, c->stream));
        DXV_HIP(c, hipMemsetAsync(c->dPlEntries + h.plCount, 0, 8 * sizeof(uint32_t), c->stream));       // (the kernel fetches up to three slots behind a list)
        uint32_t bad[2] = {0, 0};
        DXV_HIP(c, parity_lists_validate(c->dPlCells, h.plRes, c->dPlEntries, h.plCount, h.numTris, c->dRootInfo, c->stream));
        DXV_HIP(c, hipMemcpyAsync(bad, c->dRootInfo, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
        DXV_HIP(c, hipStreamSynchronize(c->stream));
        if (bad[0] || bad[1])
            return fail(c, "dxv_scene_import: the row-list section is inconsistent (%u texels point outside the %u entries, %u entries name a triangle >= %u)",
                        bad[0], h.plCount, bad[1], h.numTris);
    }
    const uint32_t listRes = h.listRes, listCount = h.listCount, plRes = h.plRes, plCount = h.plCount;
    h.offListCells = h.offListEntries = 0; h.listRes = h.listCount = 0; h.totalBytes = want.totalBytes;   // the resident header describes the resident scene
    h.offPlCells = h.offPlEntries = 0; h.plRes = h.plCount = 0;
    DXV_HIP(c, hipMemcpyAsync(c->dScene, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    c->hdr = h;
    c->T = h.numTris; c->V = h.numVerts;
    memcpy(c->bound, h.bound, sizeof(c->bound));
    c->haveScene = true;
    if (withLists && (c->optListRes == 0 || (uint32_t)c->optListRes == listRes)) {   // (an explicit listres of another size: built here as asked)
        c->listEntries = listCount; c->listRes = listRes; c->listState = 1; c->listOpt = c->optListRes; c->listMs = 0.0f;
        ++c->listEpoch;
    }
    if (withPl && (c->optPlistRes == 0 || (uint32_t)c->optPlistRes == plRes)) {
        c->plEntries = plCount; c->plRes = plRes; c->plState = 1; c->plMs = 0.0f;
    }
    c->stackNow = stack_round_up((int)(h.treeHeight + 3 < (uint32_t)c->optStack0 ? h.treeHeight + 3 : (uint32_t)c->optStack0));
    c->stats.num_tris = h.numTris; c->stats.num_verts = h.numVerts; c->stats.num_nodes = h.numNodes;
    c->stats.tree_height = h.treeHeight;
    c->stats.tri_extent = h.triExtent;
    memcpy(c->stats.bound, h.bound, sizeof(h.bound));
    return 0;
}

int dxv_get_stats(const dxv_ctx* c, dxv_stats* out)
{
    if (!c || !out) return 1;
    *out = c->stats;
    const Frame& f = c->frames[c->cur];
    out->voxelize_ms = f.voxelize_ms; out->grid_dim = f.grid_dim; out->z0 = f.z0; out->nz = f.nz;
    out->stack_entries = f.stack_entries; out->redo_rays = f.redo_rays; out->row_block = f.row_block;
    out->list_entries = f.list_entries; out->list_res = f.list_res; out->list_ms = f.lastMode == DXV_MODE_PARITY ? c->plMs : c->listMs;
    out->plan_bricks = f.plan_bricks; out->plan_waves = f.plan_waves; out->plan_ms = f.plan_ms;
    return 0;
}

int dxv_set_option(dxv_ctx* c, const char* key, int64_t value)
{
    if (!c || !key) return 1;
    if (!strcmp(key, "brick")) {
        if (value < 0 || value >= num_brick_shapes()) return fail(c, "option brick: %lld out of range", (long long)value);
        c->optBrick = (int)value;
    } else if (!strcmp(key, "stack")) {
        if (value != 0 && (value < 0 || value > 64 || stack_round_up((int)value) != (int)value))
            return fail(c, "option stack: %lld not in {0,8,12,16,24,32,48,64}", (long long)value);
        c->optStack = (int)value;
    } else if (!strcmp(key, "refit")) {
        if (value < 0 || value > 2) return fail(c, "option refit: %lld not in {0,1,2}", (long long)value);
        c->optRefit = (int)value;
    } else if (!strcmp(key, "deferboxes")) {
        if (value < 0 || value > 1) return fail(c, "option deferboxes: %lld not in {0,1}", (long long)value);
        c->optDeferBoxes = (int)value;
    } else if (!strcmp(key, "subbox")) {
        if (value != 0 && value != 1) return fail(c, "option subbox: %lld not in {0,1}", (long long)value);
        c->optSubbox = (int)value;
    } else if (!strcmp(key, "wide")) {
        if (value < 0 || value > 2) return fail(c, "option wide: %lld not in {0,1,2}", (long long)value);
        c->optWide = (int)value;
        // the wide copy is a section of the scene: a scene built without it is built again
        if (value && c->haveScene && !c->hdr.hasWide) {
            if (!c->haveMesh) return fail(c, "option wide: this scene was imported without wide nodes; set the option on the exporting context before dxv_build");
            return dxv_build(c);
        }
    } else if (!strcmp(key, "lists")) {
        if (value < 0 || value > 2) return fail(c, "option lists: %lld not in {0,1,2}", (long long)value);
        c->optLists = (int)value;
    } else if (!strcmp(key, "plan")) {
        if (value < 0 || value > 2) return fail(c, "option plan: %lld not in {0,1,2}", (long long)value);
        c->optPlan = (int)value;
    } else if (!strcmp(key, "queuewaves")) {
        if (value < 0 || value > (1 << 20)) return fail(c, "option queuewaves: %lld not in [0, 2^20]", (long long)value);
        c->optQueueWaves = (int)value;
    } else if (!strcmp(key, "queueheads")) {
        if (value != 1 && value != 2 && value != 4 && value != 8) return fail(c, "option queueheads: %lld not in {1,2,4,8}", (long long)value);
        c->optQueueHeads = (int)value;
    } else if (!strcmp(key, "planregion")) {
        if (value != 0 && (value < 6 || value > 8)) return fail(c, "option planregion: %lld not in {0,6,7,8}", (long long)value);
        c->optPlanRegion = (int)value;
    } else if (!strcmp(key, "planheavy")) {
        if (value < 0 || value > 65535) return fail(c, "option planheavy: %lld not in [0, 65535]", (long long)value);
        c->optPlanHeavy = (int)value;
    } else if (!strcmp(key, "fuse")) {
        if (value != 0 && value != 1) return fail(c, "option fuse: %lld not in {0,1}", (long long)value);
        c->optFuse = (int)value;
    } else if (!strcmp(key, "events")) {
        if (value != 0 && value != 1) return fail(c, "option events: %lld not in {0,1}", (long long)value);
        c->optEvents = (int)value;
    } else if (!strcmp(key, "plistres")) {
        if (value != 0 && (value < 16 || value > 4096 || (value & (value - 1)))) return fail(c, "option plistres: %lld is not 0 or a power of two in [16, 4096]", (long long)value);
        if (c->optPlistRes != (int)value) { if (sync_frames(c)) return 1; c->plState = 0; }     // the next parity launch rebuilds the row lists
        c->optPlistRes = (int)value;
    } else if (!strcmp(key, "plists")) {
        if (value < 0 || value > 2) return fail(c, "option plists: %lld not in {0,1,2}", (long long)value);
        c->optPlists = (int)value;
    } else if (!strcmp(key, "listres")) {
        if (value != 0 && (value < 16 || value > 4096 || (value & (value - 1)))) return fail(c, "option listres: %lld is not 0 or a power of two in [16, 4096]", (long long)value);
        if (c->optListRes != (int)value && sync_frames(c)) return 1;     // the next launch rebuilds the lists: nothing may still read them
        c->optListRes = (int)value;
    } else if (!strcmp(key, "dispatch")) {
        if (value < 0 || value > 2) return fail(c, "option dispatch: %lld not in {0,1,2}", (long long)value);
        c->optDispatch = (int)value;
    } else if (!strcmp(key, "ablate")) {
#if defined(DXV_ABLATE)
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 6 && value != 8 && value != 16 && value != 18) return fail(c, "option ablate: %lld not in {0,1,2,4,6,8,16,18}", (long long)value);
        c->optAblate = (int)value;
#else
        // the timing-only variants of the lists kernel write wrong grids by design: they exist only in the library that
        // tools/ablate.py builds for itself (python -m dxrvoxelizer_amd.build --ablate -> libdxv_ablate.so)
        if (value != 0) return fail(c, "option ablate: this library was built without the ablation kernels (-DDXV_ABLATE)");
#endif
    } else if (!strcmp(key, "skipempty")) {
        if (value != 0 && value != 1) return fail(c, "option skipempty: %lld not in {0,1}", (long long)value);
        c->optSkipEmpty = (int)value;
    } else if (!strcmp(key, "rowblock")) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return fail(c, "option rowblock: %lld not in {0,1,2,4}", (long long)value);
        c->optRowBlock = (int)value;
    } else if (!strcmp(key, "rows")) {
        if (value != 0 && value != 1) return fail(c, "option rows: %lld not in {0,1}", (long long)value);
        c->optRows = (int)value;
    } else if (!strcmp(key, "queue")) {
        if (value != 0 && value != 1) return fail(c, "option queue: %lld not in {0,1}", (long long)value);
        c->optQueue = (int)value;
    } else if (!strcmp(key, "stack0")) {
        if (value < 8 || value > 64 || stack_round_up((int)value) != (int)value) return fail(c, "option stack0: bad depth %lld", (long long)value);
        c->optStack0 = (int)value;
        if (c->haveScene) c->stackNow = stack_round_up((int)(c->hdr.treeHeight + 3 < (uint32_t)value ? c->hdr.treeHeight + 3 : (uint32_t)value));
    } else if (!strcmp(key, "region")) {
        if (value < 0 || value > 24) return fail(c, "option region: %lld not in [0,24]", (long long)value);
        c->optRegion = (int)value;
    } else if (!strcmp(key, "morton")) {
        if (value != 0 && value != 1) return fail(c, "option morton: %lld not in {0,1}", (long long)value);
        c->optMorton = (int)value;
    } else return fail(c, "unknown option '%s'", key);
    return 0;
}

int dxv_debug_list_check(dxv_ctx* c, uint32_t N, uint32_t z0, uint32_t nz, uint64_t out[34])
{
    if (!c || !out) return 1;
    if (!c->haveScene) return fail(c, "dxv_debug_list_check: no scene");
    if (N < 2 || (N & 1u) || N > 2048) return fail(c, "dxv_debug_list_check: grid_dim must be even and in [2, 2048], got %u", N);
    if (nz == 0 || z0 >= N || nz > N - z0) return fail(c, "dxv_debug_list_check: slab [%u, %u+%u) outside the grid (N=%u)", z0, z0, nz, N);
    if (c->hdr.treeHeight + 1 > 64) return fail(c, "dxv_debug_list_check: tree too deep for the checker's stack");
    DXV_HIP(c, hipSetDevice(c->device));
    if (sync_frames(c)) return 1;
    if (c->listState == 0 || c->listOpt != c->optListRes) {
        if (build_lists(c, c->stream)) return 1;
    }
    if (c->listState != 1) return fail(c, "dxv_debug_list_check: this scene has no lists (over the caps)");
    if (ensure_nodes(c, c->stream)) return 1;
    unsigned long long* dOut = nullptr;
    DXV_HIP(c, hipMalloc(&dOut, 34 * sizeof(unsigned long long)));
    VoxelizeParams p{};
    p.scene.nodes = scene_nodes32(c); p.scene.triPos = scene_tripos(c); p.scene.triNrm = scene_trinrm(c);
    memcpy(p.scene.rootLo, c->hdr.rootLo, 12);
    memcpy(p.scene.rootHi, c->hdr.rootHi, 12);
    p.scene.dmCells = c->dListCells; p.scene.dmEntries = c->dListEntries; p.scene.dmR = c->listRes;
    p.N = N; p.z0 = z0; p.nz = nz;
    hipError_t e = hipMemsetAsync(dOut, 0, 34 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = launch_list_check(p, dOut, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dOut, 34 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dOut);
    if (e != hipSuccess) return fail(c, "dxv_debug_list_check failed: %s", hipGetErrorString(e));
    return 0;
}

int dxv_debug_class_check(dxv_ctx* c, uint32_t N, uint32_t z0, uint32_t nz, uint64_t out[34])
{
    if (!c || !out) return 1;
    if (!c->haveScene) return fail(c, "dxv_debug_class_check: no scene");
    if (N < 2 || (N & 1u) || N > 2048) return fail(c, "dxv_debug_class_check: grid_dim must be even and in [2, 2048], got %u", N);
    if (nz == 0 || z0 >= N || nz > N - z0) return fail(c, "dxv_debug_class_check: slab [%u, %u+%u) outside the grid (N=%u)", z0, z0, nz, N);
    if (c->hdr.treeHeight + 1 > 64) return fail(c, "dxv_debug_class_check: tree too deep for the checker's stack");
    DXV_HIP(c, hipSetDevice(c->device));
    if (sync_frames(c)) return 1;
    if (ensure_nodes(c, c->stream)) return 1;
    unsigned long long* dOut = nullptr;
    DXV_HIP(c, hipMalloc(&dOut, 34 * sizeof(unsigned long long)));
    VoxelizeParams p{};
    p.scene.nodes = scene_nodes32(c); p.scene.triPos = scene_tripos(c); p.scene.triNrm = scene_trinrm(c);
    memcpy(p.scene.rootLo, c->hdr.rootLo, 12);
    memcpy(p.scene.rootHi, c->hdr.rootHi, 12);
    p.N = N; p.z0 = z0; p.nz = nz;
    hipError_t e = hipMemsetAsync(dOut, 0, 34 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = launch_class_check(p, dOut, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dOut, 34 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dOut);
    if (e != hipSuccess) return fail(c, "dxv_debug_class_check failed: %s", hipGetErrorString(e));
    return 0;
}

int dxv_debug_plan_check(dxv_ctx* c, uint64_t out[16])
{
    if (!c || !out) return 1;
    Frame& f = cur_frame(c);
    if (settle_lists(c)) return 1;
    if (!c->haveScene || c->listState != 1 || !f.lastQueued || !f.dQueue || !f.grid_dim)
        return fail(c, "dxv_debug_plan_check: the current frame's last launch did not go through a work queue");
    DXV_HIP(c, hipSetDevice(c->device));
    if (sync_frames(c)) return 1;
    const hipStream_t fs = cur_stream(c);
    VoxelizeParams p{};
    memcpy(p.scene.rootLo, c->hdr.rootLo, 12);
    memcpy(p.scene.rootHi, c->hdr.rootHi, 12);
    p.scene.dmCells = c->dListCells; p.scene.dmEntries = c->dListEntries; p.scene.dmR = c->listRes;
    p.N = f.grid_dim; p.z0 = f.z0; p.nz = f.nz; p.zBlock = f.lastZBlock; p.zPeriod = f.lastZPeriod;
    while ((1u << p.zShift) < p.zBlock) ++p.zShift;
    uint32_t cap = 0;
    (void)plan_queue_words(p.N, p.nz, &cap);
    p.queue = f.dQueue + f.queueHdr * kQueueHeaderWords; p.queueSlots = f.dQueue + kQueueSlotsAt; p.queueCap = cap; p.mip = c->dMip;
    VoxelizeParams q = p;
    const uint32_t nb = plan_layout(q);
    uint32_t* bits = nullptr;
    unsigned long long* dOut = nullptr;
    DXV_HIP(c, hipMalloc(&bits, sizeof(uint32_t) * (((size_t)nb + 31u) / 32u)));
    hipError_t e = hipMalloc(&dOut, 16 * sizeof(unsigned long long));
    if (e == hipSuccess) e = launch_plan_check(p, bits, dOut, fs);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dOut, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, fs);
    if (e == hipSuccess) e = hipStreamSynchronize(fs);
    (void)hipFree(bits); (void)hipFree(dOut);
    if (e != hipSuccess) return fail(c, "dxv_debug_plan_check failed: %s", hipGetErrorString(e));
    return 0;
}

int dxv_debug_download(dxv_ctx* c, int what, void* host, size_t bytes)
{
    if (!c || !host) return 1;
    const void* src = nullptr;
    size_t want = 0;
    const size_t T = c->T;
    if (settle_lists(c)) return 1;
    if ((what == DXV_DBG_NODES || what == DXV_DBG_NODES32 || what == DXV_DBG_NODES64) && c->haveScene && ensure_nodes(c, c->stream)) return 1;
    switch (what) {
    case DXV_DBG_SORTED_KEYS: src = c->dKeys; want = sizeof(uint64_t) * T; if (c->scratchT != c->T) src = nullptr; break;
    case DXV_DBG_PARENTS: src = c->dParents; want = sizeof(uint32_t) * (2 * T - 1); if (c->scratchT != c->T) src = nullptr; break;
    case DXV_DBG_NODES: if (c->haveScene) { src = scene_nodes(c); want = sizeof(Node) * (size_t)c->hdr.numNodes; } break;
    case DXV_DBG_NODES32: if (c->haveScene) { src = scene_nodes32(c); want = sizeof(Node32) * (size_t)c->hdr.numNodes; } break;
    case DXV_DBG_NODES64: if (c->haveScene && c->hdr.hasWide) { src = scene_nodes64(c); want = sizeof(Node64) * (size_t)c->hdr.numNodes; } break;
    case DXV_DBG_TRI_POS: if (c->haveScene) { src = scene_tripos(c); want = sizeof(TriPos) * T; } break;
    case DXV_DBG_TRI_NRM: if (c->haveScene) { src = scene_trinrm(c); want = sizeof(TriNrm) * T; } break;
    case DXV_DBG_LIST_CELLS: if (c->haveScene && c->listState == 1) { src = c->dListCells; want = sizeof(DirCell) * 6 * (size_t)c->listRes * c->listRes; } break;
    case DXV_DBG_LIST_ENTRIES: if (c->haveScene && c->listState == 1) { src = c->dListEntries; want = sizeof(DirEntry) * (size_t)c->listEntries; } break;
    case DXV_DBG_LIST_MIP: if (c->haveScene && c->listState == 1 && c->dMip) { src = c->dMip; want = sizeof(uint16_t) * (size_t)dm_mip_words(c->listRes); } break;
#if defined(DXV_QUEUE_TIMES)
    case 100: src = c->frames[c->cur].dRedo; want = sizeof(uint64_t) * kRedoCap; break;      // per-wave start / end ticks of the last queue launch
#endif
    default: return fail(c, "dxv_debug_download: unknown selector %d", what);
    }
    if (!src) return fail(c, "dxv_debug_download: selector %d not available", what);
    if (bytes != want) return fail(c, "dxv_debug_download: expected %zu bytes, got %zu", want, bytes);
    DXV_HIP(c, hipSetDevice(c->device));
    DXV_HIP(c, hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, c->stream));
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

} // extern "C"
