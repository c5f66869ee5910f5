// dxv_lists.hip -- the candidate lists of the two rules (direction-space lists of the reference rule, row lists of the parity
// rule): WHEN and on WHICH map they are built (the decisions are dxv_policy.h's pure functions; here they are carried out),
// their build orchestration, the deferred verdict of a build nobody waited for, and the C-ABI entry points that build them.
#include "dxv_ctx.h"

using namespace dxv;
using namespace dxvhost;

namespace dxvhost {

// Build the direction-space lists of the current scene (one-off per scene; synchronous).  Scenes whose
// lists would exceed 256 entries per triangle + 64 M (triangles through the grid centre cover whole
// faces) keep the tree walk: listState = -1.
// Texels per face side.  Measured optimum (tools/ab_lists.py): 5-10 entries per texel -- coarser maps
// have long lists, finer ones stop fitting the caches: 128 below 20 k triangles, 256 up to 3 M (512 when the
// 256 map holds more than 10 entries per texel and the scene is presumed static: build_lists), 512 beyond.
uint32_t list_resolution(const dxv_ctx* c) { return lists_base_map(c->hdr.numTris, c->optListRes); }

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
int build_lists(dxv_ctx* c, hipStream_t stream, uint64_t firstLaunchVoxels, bool defer)
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
    auto recount = [&](uint32_t res, uint32_t stride = 1u) -> int {
        R = res;
        if ((e = dirmap_count(scene_tripos(c), T, R, rec, counts, sa.pairs, offsets, dTotal, stream, stride)) != hipSuccess) return bail(e, "dirmap_count");
        (void)hipEventRecord(c->evList[1], stream);
        if ((e = hipMemcpyAsync(&c->pin->listTotal, dTotal, sizeof(total), hipMemcpyDeviceToHost, stream)) != hipSuccess) return bail(e, "hipMemcpyAsync");
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
        total = c->pin->listTotal;
        return 0;
    };
    const bool oneLaunch = firstLaunchVoxels != 0 || c->refitted;
    if (counted) total = c->pin->listTotal;
    else {
        (void)hipEventRecord(c->evList[0], stream);
        // A scene of millions of triangles is first counted on every 16th of them: a soup's count on the 512 map only says that
        // the scene is deep and the 256 map it is -- 4.7 ms of a 24 ms build for that -- and a sixteenth of the triangles says so too
        // (lists_sample_first; what the estimate decides is checked against the full count below like any other choice of map)
        if (lists_sample_first(T, c->optListRes)) {
            if (recount(R, kListsSampleStride)) return 1;
            const uint32_t next = lists_recount_on(R, total * kListsSampleStride, oneLaunch, c->optListRes, false);
            if (next && next < R) R = next;
        }
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
    // (the decisions are dxv_policy.h's: lists_pay_on_first_launch, lists_recount_on, lists_over_the_caps)
    if (firstLaunchVoxels && !lists_pay_on_first_launch(firstLaunchVoxels, total, R)) { release(); return 0; }
    // (round 4: with no entries for texels outside a triangle's outline the 512 map beats the 256 map at every grid size measured,
    // 128^3 to 1024^3, by 5 - 22 % -- profiles/r04/ab_texels_outside_the_outline.jsonl -- so every scene that is presumed static
    // takes it; a mesh that is being refitted, or a first launch that must pay for its build at once, keeps the base map; deep
    // scenes -- over 32 entries per texel: soups -- take the 256 map while it stays below 320 M entries, else the 128 map)
    for (int again = 0; again < 2; ++again) {
        const uint32_t next = lists_recount_on(R, total, oneLaunch, c->optListRes, again != 0);
        if (!next || next == R || (again && next > R)) break;
        if (recount(next)) return 1;
    }
    if (lists_over_the_caps(total, T) || (unsigned long long)T > (1ull << dm_key_layout(R).triBits) || T > kDmTriMask) {
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


} // namespace dxvhost

extern "C" {

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
    if (N && (N < 2 || (N & 1u) || N > 2048)) return fail(c, "dxv_build_lists_for_grid: grid_dim must be 0 or even and in [2, 2048], got %u", N);
    DXV_HIP(c, hipSetDevice(c->device));
    if (!c->optLists) return 0;                                        // (the caller asked for tree walks: no lists, no queue -- nothing a launch would read)
    // the map the launches of a static scene move to (launch_now: the 512 map, at every grid size)
    ListsState s{};
    s.optListRes = c->optListRes; s.numTris = c->hdr.numTris; s.refitted = c->refitted; s.floorTried = c->listFloorTried;
    int rc;
    if (lists_static_scene_takes_fine_map(s, c->listResFloor)) {
        if (sync_frames(c)) return 1;
        c->listResFloor = kListsFineMap; c->listFloorTried = true;
        rc = c->listState == 1 && c->listRes >= 512u ? 0 : build_lists(c, c->stream);
    } else rc = dxv_build_lists(c);
    if (rc || !N) return rc;
    // ... and the work queue of the whole grid the caller names: Init-time structure like the lists (include/dxv.h, dxv_prepare_launch)
    return dxv_prepare_launch(c, N, 0, N);
}

} // extern "C"
