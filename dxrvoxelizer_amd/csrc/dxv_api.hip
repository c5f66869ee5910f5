// dxv_api.hip -- the C-ABI of libdxv.so (include/dxv.h): context, device memory, mesh, build and refit, results,
// options.  The other entry points: dxv_frames.hip (launches), dxv_lists.hip (candidate lists), dxv_blob.hip (scene blob),
// dxv_debug.hip (test hooks).  There is no CPU fallback anywhere in this library: without a HIP device dxv_create fails.
#include "dxv_ctx.h"
#include <chrono>

using namespace dxv;
using namespace dxvhost;

namespace {
thread_local std::string g_createError;
}

namespace dxvhost {

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
    // (a scene of up to the size of the last one, and not under half of it, moves into its allocation: a free and an allocation
    // of a hundred megabytes less on the way to the first launch)
    if (c->dScene && (h.totalBytes > c->sceneCap || c->sceneCap > 2 * h.totalBytes)) { (void)hipFree(c->dScene); c->dScene = nullptr; c->sceneCap = 0; }
    if (!c->dScene) { DXV_HIP(c, hipMalloc(&c->dScene, h.totalBytes)); c->sceneCap = h.totalBytes; }
    c->sceneBytes = h.totalBytes;
    c->hdr = h;
    return 0;
}

void free_scratch(dxv_ctx* c)
{
    (void)hipFree(c->dKeys); (void)hipFree(c->dKeysTmp); (void)hipFree(c->dHist); (void)hipFree(c->dParents);
    (void)hipFree(c->dPyramid); c->dPyramid = nullptr; c->pyramidSlots = 0;
    (void)hipFree(c->dFlags); (void)hipFree(c->dFlags2);
    c->dKeys = c->dKeysTmp = nullptr; c->dHist = c->dParents = c->dFlags = c->dFlags2 = nullptr;
    c->scratchT = 0; c->scratchCap = 0; c->histCapWords = 0;
}

int alloc_scratch(dxv_ctx* c, uint32_t T)
{
    if (c->scratchT == T) return 0;
    // (scratch made for a mesh of up to twice the triangles serves this one too)
    if (c->scratchCap >= T && c->scratchCap / 2 <= T && radix_sort_hist_words(T) <= c->histCapWords) {
        if (c->dPyramid && c->pyramidSlots < pyramid_slots(T)) { (void)hipFree(c->dPyramid); c->dPyramid = nullptr; c->pyramidSlots = 0; }   // (alloc_pyramid makes the larger one)
        c->scratchT = T;
        return 0;
    }
    free_scratch(c);
    DXV_HIP(c, hipMalloc(&c->dKeys, sizeof(uint64_t) * (size_t)T));
    DXV_HIP(c, hipMalloc(&c->dKeysTmp, sizeof(uint64_t) * (size_t)T));
    DXV_HIP(c, hipMalloc(&c->dHist, sizeof(uint32_t) * (size_t)radix_sort_hist_words(T)));
    DXV_HIP(c, hipMalloc(&c->dParents, sizeof(uint32_t) * (2 * (size_t)T)));
    DXV_HIP(c, hipMalloc(&c->dFlags, sizeof(uint32_t) * (size_t)T));
    DXV_HIP(c, hipMalloc(&c->dFlags2, sizeof(uint32_t) * (size_t)T));
    c->scratchT = T; c->scratchCap = T; c->histCapWords = radix_sort_hist_words(T);
    return 0;
}

// dxv_set_mesh's look at the caller's arrays: largest index, position bounds, whether every position is finite.  Chunks of the
// arrays go to up to eight threads; the chunks' results are merged in order with the strict comparisons of a sequential sweep, so
// the bounds are the sequential sweep's bit for bit (signs of zero included).
struct MeshScan { uint32_t maxIndex; float mn[3], mx[3]; bool finite; };
static void scan_mesh_range(const float* vb, size_t v0, size_t v1, const uint32_t* ib, size_t i0, size_t i1, MeshScan* out)
{
    uint32_t m = 0;
    for (size_t i = i0; i < i1; ++i) m = ib[i] > m ? ib[i] : m;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, probe = 0.0f;
    for (size_t v = v0; v < v1; ++v) {
        const float* p = vb + 6 * v;
        for (int a = 0; a < 3; ++a) {
            mn[a] = p[a] < mn[a] ? p[a] : mn[a];
            mx[a] = p[a] > mx[a] ? p[a] : mx[a];
            probe += p[a] - p[a];                 // 0 for every finite value, NaN for NaN and +-Inf
        }
    }
    out->maxIndex = m;
    for (int a = 0; a < 3; ++a) { out->mn[a] = mn[a]; out->mx[a] = mx[a]; }
    out->finite = probe == 0.0f;
}
static MeshScan scan_mesh(const float* vb, uint32_t V, const uint32_t* ib, uint32_t T)
{
    const size_t nIdx = 3 * (size_t)T;
    unsigned hw = std::thread::hardware_concurrency();
    // (a thread per quarter of a million elements, up to eight: starting and joining one costs ~50 us, as much as it sweeps in that time --
    // the bunny's 245 k elements went from 0.2 to 0.36 ms with eight)
    unsigned nt = (unsigned)((nIdx + V) / 250000u);
    nt = nt < 1u ? 1u : nt > 8u ? 8u : nt;
    if (hw && nt > hw) nt = hw;
    if (nt > V) nt = 1;
    MeshScan part[8];
    std::thread th[8];
    unsigned started = 1;
    auto lo = [&](size_t n, unsigned k) { return n * k / nt; };
    for (unsigned k = 1; k < nt; ++k) {
        try { th[k] = std::thread(scan_mesh_range, vb, lo(V, k), lo(V, k + 1), ib, lo(nIdx, k), lo(nIdx, k + 1), &part[k]); ++started; }
        catch (...) { break; }                 // (no thread to be had: the caller's thread does the rest)
    }
    if (started < nt) {                        // what the threads that did not start would have covered
        scan_mesh_range(vb, lo(V, started), V, ib, lo(nIdx, started), nIdx, &part[started]);
        for (unsigned k = started + 1; k < nt; ++k) { part[k] = part[started]; }
    }
    scan_mesh_range(vb, 0, lo(V, 1), ib, 0, lo(nIdx, 1), &part[0]);
    for (unsigned k = 1; k < started; ++k) th[k].join();
    MeshScan r = part[0];
    for (unsigned k = 1; k < nt; ++k) {
        r.maxIndex = part[k].maxIndex > r.maxIndex ? part[k].maxIndex : r.maxIndex;
        for (int a = 0; a < 3; ++a) {
            if (part[k].mn[a] < r.mn[a]) r.mn[a] = part[k].mn[a];
            if (part[k].mx[a] > r.mx[a]) r.mx[a] = part[k].mx[a];
        }
        r.finite = r.finite && part[k].finite;
    }
    return r;
}

} // namespace dxvhost

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
    // A process's first build and first launch on a device pay for what the runtime sets up lazily (dxv_warmup, below): the first
    // context of a process on a device runs that pass once (DXV_WARMUP=0: not at all) and says in its stats what it cost.
    const char* wu = getenv("DXV_WARMUP");
    if (!(wu && wu[0] == '0')) (void)dxv_warmup(device, &c->stats.warmup_ms);
    return 0;
}

// The code object of the library, the first host-to-device copy's staging, every kernel's first dispatch: 11 ms for the first Init of a
// 1 M-triangle mesh against 3.3 for one that follows anything at all (LBVH 1.3 against 0.29 ms, lists 1.6 against 0.95, upload 7
// against 0.7).  This sends a four-triangle scene through every step once, on a context of its own that is gone when it returns.
// Once per process and device; errors of the pass are nobody's (the caller's own calls will report theirs), and the error state the
// caller sees is left as it was.
int dxv_warmup(int device, float* ms)
{
    static std::atomic<uint64_t> warmed{0};
    if (ms) *ms = 0.0f;
    if (device < 0) return 1;
    const uint64_t bit = 1ull << (device & 63);
    if (warmed.fetch_or(bit) & bit) return 0;
    const std::string keep = g_createError;
    const auto t0 = std::chrono::steady_clock::now();
    dxv_ctx* w = nullptr;
    if (dxv_create(&w, device) == 0) {                                 // (its own call of this function finds the device marked)
        static const float vb[4 * 6] = {0.9f, 0.9f, 0.9f, 0.58f, 0.58f, 0.58f,   -0.9f, -0.9f, 0.9f, -0.58f, -0.58f, 0.58f,
                                        -0.9f, 0.9f, -0.9f, -0.58f, 0.58f, -0.58f,   0.9f, -0.9f, -0.9f, 0.58f, -0.58f, -0.58f};
        static const uint32_t ib[4 * 3] = {0, 1, 2, 0, 3, 1, 0, 2, 3, 1, 3, 2};
        if (dxv_set_mesh(w, vb, 4, ib, 4) == 0 && dxv_build(w) == 0 && dxv_build_lists_for_grid(w, 32) == 0) {
            (void)dxv_voxelize(w, 32, DXV_MODE_REFERENCE, 0, 32);       // (through the prepared queue)
            (void)dxv_set_option(w, "prepared", 0);
            (void)dxv_voxelize(w, 32, DXV_MODE_REFERENCE, 0, 32);       // (through a queue of its own)
            (void)dxv_voxelize(w, 32, DXV_MODE_PARITY, 0, 32);
            std::vector<uint8_t> host(32 * 32 * 32);
            (void)dxv_grid_download(w, host.data(), host.size());
        }
        std::vector<uint8_t> mb(1u << 20);                       // (a copy of a size the staging path handles in pieces)
        void* d = nullptr;
        if (hipMalloc(&d, mb.size()) == hipSuccess) { (void)hipMemcpy(d, mb.data(), mb.size(), hipMemcpyHostToDevice); (void)hipFree(d); }
        dxv_destroy(w);
    }
    (void)hipGetLastError();
    g_createError = keep;
    if (ms) *ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
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
    drop_prepared(c, true);
    (void)hipFree(c->dFar32); (void)hipFree(c->dFarCells); (void)hipFree(c->dFarMip);
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
    // prepared queues of lists that are gone (their slots keep their memory for the next dxv_prepare_launch of the partition: 8 MB at
    // 512^3, half a gigabyte at 2048^3); the ones in use stay
    for (auto& q : c->prepared)
        if (q.epoch != c->listEpoch || c->listState != 1) {
            (void)hipFree(q.dMem); (void)hipFree(q.dLive);
            q.dMem = q.dLive = nullptr; q.words = q.liveWords = 0; q.epoch = 0; q.bricks = 0;
        }
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
    // One sweep over the caller's arrays, in up to eight threads: the largest index, and the AABB over every VB position
    // (XUSGObjLoader.cpp:386-416) -- branch-free, so the compiler vectorises the index half; the slow loops that name the
    // offending element run only when the sweep has found one.  Centre and half max extent: Content/Voxelizer.cpp:52-57.
    const MeshScan scan = scan_mesh(vb, V, ib, T);
    if (scan.maxIndex >= V)
        for (size_t i = 0; i < 3 * (size_t)T; ++i)
            if (ib[i] >= V) return fail(c, "dxv_set_mesh: index %u at position %zu out of range (V=%u)", ib[i], i, V);
    if (!scan.finite)
        for (uint32_t i = 0; i < V; ++i) {
            const float* p = vb + 6 * (size_t)i;
            if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2]))
                return fail(c, "dxv_set_mesh: vertex %u has a non-finite position (%g, %g, %g)", i, (double)p[0], (double)p[1], (double)p[2]);
        }
    const float* mn = scan.mn;
    const float* mx = scan.mx;
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    // (into locals: a mesh that is refused leaves the context's earlier mesh AND its bound alone -- a later dxv_refit / dxv_build of
    // that mesh normalises with the bound it was set with)
    const float eyz = ey > ez ? ey : ez;
    const float bound[4] = {(mx[0] + mn[0]) / 2.0f, (mx[1] + mn[1]) / 2.0f, (mx[2] + mn[2]) / 2.0f, (ex > eyz ? ex : eyz) / 2.0f};
    if (!(bound[3] > 0.0f) || !std::isfinite(bound[3]) || !std::isfinite(bound[0]) || !std::isfinite(bound[1]) || !std::isfinite(bound[2]))
        return fail(c, "dxv_set_mesh: degenerate or non-finite bound (half extent %g)", (double)bound[3]);

    DXV_HIP(c, hipSetDevice(c->device));
    if (sync_frames(c)) return 1;
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    c->vbCopyQueued = false;
    memcpy(c->bound, bound, sizeof(bound));
    drop_prepared(c);
    c->haveMesh = false; c->haveScene = false; c->haveHierarchy = false; c->listState = 0; c->specRes = 0; c->listResFloor = 0; c->listFloorTried = false; c->refitted = false; c->launchesOfScene = 0; c->plState = 0; c->parityLaunchesOfScene = 0; c->nodesStale = 0;
    // (a mesh of the size of the last one moves into its buffers: two frees and two allocations less on the way to the first launch)
    const size_t vbBytes = sizeof(float) * 6 * (size_t)V, ibBytes = sizeof(uint32_t) * 3 * (size_t)T;
    if (!c->dVb || c->vbCap < vbBytes || c->vbCap > 2 * vbBytes) {
        (void)hipFree(c->dVb); c->dVb = nullptr; c->vbCap = 0;
        DXV_HIP(c, hipMalloc(&c->dVb, vbBytes));
        c->vbCap = vbBytes;
    }
    if (!c->dIb || c->ibCap < ibBytes || c->ibCap > 2 * ibBytes) {
        (void)hipFree(c->dIb); c->dIb = nullptr; c->ibCap = 0;
        DXV_HIP(c, hipMalloc(&c->dIb, ibBytes));
        c->ibCap = ibBytes;
    }
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

} // extern "C"

namespace dxvhost {
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
    if (!c->dPyramid && c->optRefit == 1 && c->T > 1) {    // refit=2 keeps the level sweeps, refit=0 the atomic pass
        DXV_HIP(c, hipMalloc(&c->dPyramid, 28 * (size_t)pyramid_slots(c->T)));
        c->pyramidSlots = pyramid_slots(c->T);
    }
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
    ++c->sceneEpoch;
    c->stackNow = stack_round_up((int)(c->hdr.treeHeight + 3 < (uint32_t)c->optStack0 ? c->hdr.treeHeight + 3 : (uint32_t)c->optStack0));
    c->stats.num_nodes = c->hdr.numNodes;
    c->stats.tree_height = c->hdr.treeHeight;
    c->stats.tri_extent = c->hdr.triExtent;
    return 0;
}
} // namespace dxvhost

extern "C" {

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
    drop_prepared(c);                                                  // (queues prepared for the old surface; no launch that reads one is un-ordered: see above)
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
        DXV_HIP(c, dirmap_count(scene_tripos(c), c->hdr.numTris, spec, sa.rec, sa.counts, sa.pairs, sa.offsets, sa.total, c->stream));
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
    drop_prepared(c);
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

int dxv_get_stats(const dxv_ctx* c, dxv_stats* out)
{
    if (!c || !out) return 1;
    *out = c->stats;
    const Frame& f = c->frames[c->cur];
    out->voxelize_ms = f.voxelize_ms; out->grid_dim = f.grid_dim; out->z0 = f.z0; out->nz = f.nz;
    out->stack_entries = f.stack_entries; out->redo_rays = f.redo_rays; out->row_block = f.row_block;
    out->list_entries = f.list_entries; out->list_res = f.list_res; out->list_ms = f.lastMode == DXV_MODE_PARITY ? c->plMs : c->listMs;
    out->plan_bricks = f.plan_bricks; out->plan_waves = f.plan_waves; out->plan_ms = f.plan_ms;
    out->plan_prepared = f.lastPrepared >= 0 ? 1u : 0u;
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
    } else if (!strcmp(key, "prepared")) {
        if (value != 0 && value != 1) return fail(c, "option prepared: %lld not in {0,1}", (long long)value);
        c->optPrepared = (int)value;
    } else if (!strcmp(key, "listedwaves")) {
        if (value != 0 && (value < 8 || value > 32)) return fail(c, "option listedwaves: %lld not in {0,8..32}", (long long)value);
        c->optListedWaves = (int)value;
    } else if (!strcmp(key, "coop")) {
        if (value != 0 && value != 1) return fail(c, "option coop: %lld not in {0,1}", (long long)value);
        c->optCoop = (int)value;
    } else if (!strcmp(key, "farmap")) {
        if (value != 0 && value != 1) return fail(c, "option farmap: %lld not in {0,1}", (long long)value);
        c->optFarMap = (int)value;
    } else if (!strcmp(key, "prepclear")) {
        if (value < 0 || value > 3) return fail(c, "option prepclear: %lld not in {0,1,2,3}", (long long)value);
        c->optPrepClear = (int)value;
    } else if (!strcmp(key, "queuewaves")) {
        if (value < 0 || value > (1 << 20)) return fail(c, "option queuewaves: %lld not in [0, 2^20]", (long long)value);
        c->optQueueWaves = (int)value;
    } else if (!strcmp(key, "queuemin")) {
        if (value < 0 || value > 4096) return fail(c, "option queuemin: %lld not in [0, 4096]", (long long)value);
        c->optQueueMin = (int)value;
    } else if (!strcmp(key, "sortbits")) {
        if (value < 0 || value > 63 || ((value & 15) != 0 && ((value & 15) < 8 || (value & 15) > 11)))
            return fail(c, "option sortbits: %lld not 0 or 8..11 (+16 / +32)", (long long)value);
        radix_sort_set_plan((int)value);
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
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 6 && value != 8 && value != 16 && value != 18 && value != 32 && value != 64) return fail(c, "option ablate: %lld not in {0,1,2,4,6,8,16,18,32,64}", (long long)value);
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

} // extern "C"
