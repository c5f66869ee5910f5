// dxv_blob.hip -- the built scene as one relocatable device blob (what rank 0 broadcasts to the other GPUs): layout, export,
// checksum, import with validation of everything the kernels will index with.
#include "dxv_ctx.h"

using namespace dxv;
using namespace dxvhost;

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

extern "C" {

size_t dxv_scene_bytes(const dxv_ctx* c)
{
    if (!c || !c->haveScene) return 0;
    return export_layout(c).total;
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
    drop_prepared(c);
    c->haveScene = false; c->listState = 0; c->specRes = 0; c->listResFloor = 0; c->listFloorTried = false; c->refitted = false; c->launchesOfScene = 0; c->plState = 0; c->parityLaunchesOfScene = 0; c->nodesStale = 0;
    // An imported scene carries no mesh and no build state: drop what an earlier dxv_set_mesh / dxv_build left on this
    // context, so that dxv_build, dxv_refit and dxv_update_vertices fail cleanly instead of running the imported
    // triangle count over the old, smaller buffers.
    DXV_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->dVb); (void)hipFree(c->dIb);
    c->dVb = nullptr; c->dIb = nullptr; c->vbCap = c->ibCap = 0; c->haveMesh = false; c->haveHierarchy = false;
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
    ++c->sceneEpoch;
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

} // extern "C"
