// dxv_ctx.h -- the context of libdxv.so and what its translation units share: dxv_api.hip (context, mesh, build, options,
// results), dxv_lists.hip (the candidate lists' policy and builds), dxv_frames.hip (frames, launches, work queues),
// dxv_blob.hip (the scene blob that travels between GPUs), dxv_debug.hip (test hooks).  Nothing here is exported.
#pragma once
#include "../../include/dxv.h"
#include "dxv_device.h"
#include "dxv_raycast.h"
#include "dxv_dirmap.h"
#include "dxv_policy.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

namespace dxvhost {
#if defined(DXV_QUEUE_TIMES)
constexpr uint32_t kRedoCap = 1u << 21;   // (diagnostic build: the list doubles as the buffer of per-workgroup time stamps)
#else
constexpr uint32_t kRedoCap = 1u << 16;   // rays per launch the redo pass takes before the column is grown instead
#endif
inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
}

using namespace dxv;
using dxvhost::kRedoCap;

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
    size_t vbCap = 0, ibCap = 0;     // bytes allocated behind them (dxv_set_mesh moves a mesh of about the same size into the same buffers)
    uint32_t T = 0, V = 0;
    float bound[4] = {0, 0, 0, 0};
    bool haveMesh = false;

    // scene blob
    uint8_t* dScene = nullptr;
    size_t sceneBytes = 0, sceneCap = 0;   // bytes of the scene / of the allocation behind it
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
    uint32_t scratchT = 0;           // triangles the build scratch is in use for (0: none)
    uint32_t scratchCap = 0;         // ... and the number it was allocated for (alloc_scratch keeps it for meshes of half to all of that)
    size_t histCapWords = 0;
    uint32_t pyramidSlots = 0;       // slots dPyramid holds

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
        int lastPrepared = -1;           // the frame's last launch ran this PREPARED queue of the context (-1: none)
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
        uint32_t preparedLens[16 * 64];                  // ... of a queue that is being prepared
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
    // far-radius map of a scene WITHOUT lists (dirmap_far): the brick test of its tree walks; made at the scene's first tree walk
    uint64_t sceneEpoch = 0;         // counts builds / refits / imports
    uint64_t farEpoch = 0;           // ... the one the far map was made for (0: none)
    uint64_t boxLaunchEpoch = 0;     // the scene epoch boxLaunchesOfScene counts for
    uint32_t boxLaunchesOfScene = 0; // reference-rule launches over the brick box (tree walks, plan = 0) since the scene last changed
    uint32_t* dFar32 = nullptr;
    DirCell* dFarCells = nullptr;
    uint16_t* dFarMip = nullptr;
    uint32_t farR = 0, farCap = 0;   // the map it is on / was allocated for
    float farMs = 0.0f;
    int optListedWaves = 0;          // workgroups per CU of the hardware-dispatched lists kernel (8 .. 32), or 0 = by grid and map (traverse.hip: listed_lds_pad)
    int optCoop = 1;                 // 1: the lists kernel scans a lone lane's long list with its whole wave (dxv_dirmap.h: trace_reference_dm_from)
    int optFarMap = 1;               // 1: tree walks and brick-box launches of the reference rule skip the bricks none of whose rays can reach a triangle
    int optPlan = 2;                 // work queue of the lists kernel (live bricks only, built on the device inside the stream): 0 = none (brick box
                                     // in Morton order), 1 = built when lists, partition or buffers differ from the frame's last launch (opt-in), 2 = on every launch (default: nothing carried)
    int optQueueWaves = 0;           // persistent waves of a queue launch; 0 = what the device holds at once
    int optQueueMin = 0;             // persistent waves: at least this many bricks per wave (surplus waves leave at once); 0 (default) = every wave stays.
                                     // 12: torus-1M / bunny x16 at 256^3 -13 / -11 %, but dragon x9 +11 % at 256^3 and +35 % on a rank's share: not a
                                     // rule a launch can apply blind (profiles/r05/ab_surplus_waves_leave.jsonl, short_launches_queuemin12.jsonl)
    int optQueueHeads = 8;           // heads per queue (persistent waves): 1, 2, 4, 8
    int optPlanRegion = 0;           // log2 of the run of Morton bricks dealt to one queue: 6, 7, 8; 0 = by the partition's size (plan_region_bits)
    int optPlanHeavy = 0;            // list length beyond which a brick starts early; 0 = long for this scene (k_dm_heavy_thresholds), 65535: no brick does
    int optFuse = 1;                 // 1: the queue build clears the grid as well (one kernel in front of the brick kernel); 0: memsets in front of it
    int optDispatch = 1;             // a kept queue whose lengths the host knows: 0 = persistent waves all the same, 1 = one workgroup per
                                     // queued brick dealt out by the hardware (-1 ... -10 % per launch, and back-to-back launches overlap
                                     // their ends: profiles/r04/ab_dispatch_kept_queue.jsonl), 2 = that for partitions of up to 2^25 voxels only
    int optEvents = 1;               // bracket every launch with two HIP events (stats.voxelize_ms); 0: none (a caller timing its own loop)
    // PREPARED work queues (dxv_prepare_launch; the host mirrors' Init with a grid hint): the queue of a (lists, grid, partition) is a pure
    // function of them, like the lists are of the scene -- built once when they are fixed, kept with the context (not with a frame:
    // every frame's launches read it), dropped by whatever changes the scene or its lists.  A launch of a prepared partition clears
    // its grid and has the hardware deal out the queued bricks; any other launch builds its queue itself (plan = 2).
    struct Prepared {
        uint64_t epoch = 0;              // listEpoch of the lists it was probed against (0: the slot is free)
        uint32_t N = 0, z0 = 0, nz = 0, zBlock = 0, zPeriod = 0, regionBits = 0, planHeavy = 0;
        uint32_t* dMem = nullptr;        // header (the build's counters), then 8 x cap brick words
        size_t words = 0;
        uint32_t cap = 0;
        uint32_t* dLive = nullptr;       // one bit per brick of the partition: queued or not (what the launch's clear reads)
        size_t liveWords = 0;
        uint32_t lens[16] = {};          // the eight lengths and how many of each are heavy
        uint32_t bricks = 0;
        float ms = 0.0f;                 // its build on the device
        uint64_t used = 0;               // (the least recently used slot goes when all are taken)
    };
    static constexpr uint32_t kPreparedSlots = 16;   // (eight shares of a looped 8-rank partition and a few whole grids)
    Prepared prepared[kPreparedSlots];
    uint64_t preparedClock = 0;
    float prepareMs = 0.0f;          // device time of the last dxv_prepare_launch* (0: it found the partition prepared)
    int optPrepared = 1;             // launches of a prepared partition use its queue (1, default); 0: they build their own like any other launch
    int optPrepClear = 2;            // how such a launch clears: 0 = a clear kernel in front of the brick kernel, 1 / 2 = only the bricks nobody runs,
                                     // by workgroups in front of / behind the bricks' in the SAME dispatch
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

namespace dxvhost {

int fail(dxv_ctx* c, const char* fmt, ...);          // dxv_api.hip: message into the context (or the create error), returns 1

#define DXV_HIP(c, call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail((c), "%s failed: %s", #call, hipGetErrorString(e_));    \
    } while (0)

using Frame = dxv_ctx::Frame;
inline Frame& cur_frame(dxv_ctx* c) { return c->frames[c->cur]; }
inline hipStream_t frame_stream(dxv_ctx* c, uint32_t i) { return i == 0 ? c->stream : c->frames[i].ownStream; }
inline hipStream_t cur_stream(dxv_ctx* c) { return frame_stream(c, c->cur); }
inline Node* scene_nodes(dxv_ctx* c) { return reinterpret_cast<Node*>(c->dScene + c->hdr.offNodes); }
inline Node32* scene_nodes32(dxv_ctx* c) { return reinterpret_cast<Node32*>(c->dScene + c->hdr.offNodes32); }
inline Node64* scene_nodes64(dxv_ctx* c) { return reinterpret_cast<Node64*>(c->dScene + c->hdr.offNodes64); }
inline TriPos* scene_tripos(dxv_ctx* c) { return reinterpret_cast<TriPos*>(c->dScene + c->hdr.offTriPos); }
inline TriNrm* scene_trinrm(dxv_ctx* c) { return reinterpret_cast<TriNrm*>(c->dScene + c->hdr.offTriNrm); }
inline float elapsed(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return -1.0f;
    return ms;
}

// dxv_api.hip
void layout_scene(SceneHeader& h, uint32_t T, uint32_t V, bool wide);
int alloc_scene(dxv_ctx* c, uint32_t T, uint32_t V, bool wide);
void free_scratch(dxv_ctx* c);
int alloc_scratch(dxv_ctx* c, uint32_t T);
void fill_build_buffers(dxv_ctx* c, BuildBuffers& b);
int ensure_nodes(dxv_ctx* c, hipStream_t stream);       // the hierarchy's traversal copies after a refit that skipped them
// dxv_frames.hip
int ensure_far_map(dxv_ctx* c, hipStream_t s);             // the far-radius map of a scene without lists (dirmap_far), current for the scene when this returns 0
void drop_prepared(dxv_ctx* c, bool freeMemory = false);   // whatever changes the scene or its lists calls this (the slots keep their memory unless told otherwise)
int frame_prepare(dxv_ctx* c, uint32_t i);
int sync_frame(dxv_ctx* c, uint32_t i);
int sync_frames(dxv_ctx* c);
bool use_wide(const dxv_ctx* c, int mode);
int safe_stack(const dxv_ctx* c, int mode);
int launch_now(dxv_ctx* c, uint32_t frame, bool relaunch = false);
// dxv_lists.hip
struct ListScratchA { DirRecord* rec; uint32_t *counts, *offsets, *pairs, *sums; unsigned long long* total; size_t bytes; };
ListScratchA list_scratch_a(uint8_t* base, uint32_t T);
uint32_t list_resolution(const dxv_ctx* c);
int settle_lists(dxv_ctx* c);
int build_lists(dxv_ctx* c, hipStream_t stream, uint64_t firstLaunchVoxels = 0, bool defer = false);
int build_plists(dxv_ctx* c, hipStream_t stream);

} // namespace dxvhost
