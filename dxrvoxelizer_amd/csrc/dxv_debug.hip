// dxv_debug.hip -- test hooks of the C-ABI: the exhaustive device checks of the lists' superset claim, of the triangle classes
// and of the work queue, and the download of internal arrays.  Not product paths.
#include "dxv_ctx.h"

using namespace dxv;
using namespace dxvhost;

#if defined(DXV_PHASE_TIMES)
namespace dxv { hipError_t phase_times_read(unsigned long long out[16], bool reset); }     // traverse.hip, diagnostic build only
#endif

extern "C" {

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

int dxv_debug_division_check(dxv_ctx* c, uint32_t n_first, uint32_t n_last, uint64_t out[8])
{
    if (!c || !out) return 1;
    if (n_first < 2 || (n_first & 1u) || n_last > 2048 || n_last < n_first) return fail(c, "dxv_debug_division_check: need even 2 <= n_first <= n_last <= 2048");
    DXV_HIP(c, hipSetDevice(c->device));
    unsigned long long* dOut = nullptr;
    DXV_HIP(c, hipMalloc(&dOut, 10 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(dOut, 0, 10 * sizeof(unsigned long long), c->stream);
    for (uint32_t N = n_first; N <= n_last && e == hipSuccess; N += 2u) e = launch_division_check(N, dOut, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dOut, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dOut);
    if (e != hipSuccess) return fail(c, "dxv_debug_division_check failed: %s", hipGetErrorString(e));
    return 0;
}

int dxv_debug_far_check(dxv_ctx* c, uint32_t N, uint32_t z0, uint32_t nz, int lists_mip, uint64_t out[12])
{
    if (!c || !out) return 1;
    if (!c->haveScene) return fail(c, "dxv_debug_far_check: no scene");
    if (N < 2 || (N & 1u) || N > 2048) return fail(c, "dxv_debug_far_check: grid_dim must be even and in [2, 2048], got %u", N);
    if (nz == 0 || z0 >= N || nz > N - z0) return fail(c, "dxv_debug_far_check: slab [%u, %u+%u) outside the grid (N=%u)", z0, z0, nz, N);
    if (c->hdr.treeHeight + 1 > 64) return fail(c, "dxv_debug_far_check: tree too deep for the checker's stack");
    DXV_HIP(c, hipSetDevice(c->device));
    if (sync_frames(c)) return 1;
    if (ensure_nodes(c, c->stream)) return 1;
    VoxelizeParams p{};
    p.scene.nodes = scene_nodes32(c); p.scene.triPos = scene_tripos(c); p.scene.triNrm = scene_trinrm(c);
    memcpy(p.scene.rootLo, c->hdr.rootLo, 12);
    memcpy(p.scene.rootHi, c->hdr.rootHi, 12);
    p.N = N; p.z0 = z0; p.nz = nz;
    if (lists_mip) {
        if (c->listState != 1 || !c->dMip) return fail(c, "dxv_debug_far_check: this scene has no lists");
        p.mip = c->dMip; p.mipR = c->listRes;
    } else {
        if (ensure_far_map(c, c->stream)) return 1;
        p.mip = c->dFarMip; p.mipR = c->farR;
    }
    unsigned long long* dOut = nullptr;
    DXV_HIP(c, hipMalloc(&dOut, 12 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(dOut, 0, 12 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = launch_far_check(p, dOut, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dOut, 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dOut);
    if (e != hipSuccess) return fail(c, "dxv_debug_far_check failed: %s", hipGetErrorString(e));
    return 0;
}

int dxv_debug_plan_check(dxv_ctx* c, uint64_t out[16])
{
    if (!c || !out) return 1;
    Frame& f = cur_frame(c);
    if (settle_lists(c)) return 1;
    // (the frame's last launch ran a queue of the context's: still the one for that partition?  A slot may have been reused since)
    bool prepared = f.lastPrepared >= 0 && c->prepared[f.lastPrepared].epoch == c->listEpoch;
    if (prepared) {
        const auto& q = c->prepared[f.lastPrepared];
        prepared = q.N == f.grid_dim && q.z0 == f.z0 && q.nz == f.nz && q.zBlock == f.lastZBlock && q.zPeriod == f.lastZPeriod && q.dMem;
    }
    if (!c->haveScene || c->listState != 1 || !(prepared || (f.lastQueued && f.dQueue)) || !f.grid_dim)
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
    if (prepared) {                                                     // (a queue of the context's, built by dxv_prepare_launch: same layout behind ONE header)
        const auto& q = c->prepared[f.lastPrepared];
        p.queue = q.dMem; p.queueSlots = q.dMem + kQueueHeaderWords; p.queueCap = q.cap;
    }
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
#if defined(DXV_PHASE_TIMES)
    if (what == 101 || what == 102) {                                   // the lists kernel's phase sums (102: and reset); diagnostic build only
        if (bytes != 16 * sizeof(unsigned long long)) return fail(c, "dxv_debug_download: phase times are 128 bytes");
        DXV_HIP(c, hipSetDevice(c->device));
        if (sync_frames(c)) return 1;
        DXV_HIP(c, hipDeviceSynchronize());
        DXV_HIP(c, phase_times_read(static_cast<unsigned long long*>(host), what == 102));
        return 0;
    }
#endif
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
