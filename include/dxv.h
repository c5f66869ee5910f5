/*
 * dxv.h -- C-ABI of the MI355X-native DXRVoxelizer hot path (libdxv.so).
 *
 * Plain C: opaque context, plain pointers and sizes, int return codes (0 = ok, message through
 * dxv_last_error).  No HIP, torch or C++ types cross this boundary.  Each entry point names the
 * reference interface it replaces; paths are relative to /root/reference/DXRVoxelizer/.
 *
 * One context drives one GPU (one process per GPU; see INTEGRATION.md for the multi-GPU slab
 * scheme and the cgo/ctypes/C++ bindings a maintainer of the reference would add).
 */
#ifndef DXV_H
#define DXV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DXV_API __attribute__((visibility("default")))

/* Bumped whenever an entry point changes its signature or a struct of this header its layout; dxv_api_version() returns the
 * value the loaded library was built with -- a binding compares the two before its first call.
 * 6: dxv_prepare_launch / dxv_prepare_launch_interleaved (the work queue of a static scene as Init-time structure), dxv_warmup,
 *    dxv_stats grows plan_prepared / prepare_ms / warmup_ms, options prepared / prepclear, dxv_build_lists_for_grid prepares the grid
 *    it is given;
 * 5: option plan defaults to 2 (every launch builds its queue and clears its grid: nothing carried from launch to launch; the kept
 *    queue is opt-in), options planregion / planheavy / fuse / queueheads;
 * 4: dxv_stats plan fields describe the work queue, options planorder / planregion gone, dxv_debug_plan_check, dxv_trim;
 * 3: dxv_debug_list_check takes a slab (z0, nz). */
#define DXV_API_VERSION 6
DXV_API int dxv_api_version(void);

typedef struct dxv_ctx dxv_ctx;

/* Occupancy rule. */
enum {
    /* The reference's rule: one radial ray from the voxel centre, closest hit,
     * dot(normalize(interpolated vertex normal), rayDir) > 0.12
     * (Content/Shaders/DXRVoxelizer.hlsl:44-53, :58-85, :132-140, :5). */
    DXV_MODE_REFERENCE = 0,
    /* north_star's second mode on the same traversal engine: +X axis ray, watertight hit count,
     * occupancy = count & 1 (no reference counterpart). */
    DXV_MODE_PARITY = 1
};

/* What dxv_debug_download copies (tests only; layouts in dxrvoxelizer_amd/csrc/dxv_types.h). */
enum {
    DXV_DBG_SORTED_KEYS = 0, /* T x uint64: (morton30 << 32) | triangle index, ascending   */
    DXV_DBG_NODES = 1,       /* max(T-1,1) x 64 B internal nodes                              */
    DXV_DBG_TRI_POS = 2,     /* T x 48 B: 3 x {x,y,z,w}; w of vertex 0 = triangle index bits  */
    DXV_DBG_TRI_NRM = 3,     /* T x 48 B: 3 x {nx,ny,nz,0}                                    */
    DXV_DBG_PARENTS = 4,     /* (T-1) internal + T leaf parent words: (parent << 1) | side    */
    DXV_DBG_NODES32 = 5,     /* max(T-1,1) x 32 B traversal nodes (half-float boxes)          */
    DXV_DBG_NODES64 = 6,     /* max(T-1,1) x 64 B wide traversal nodes (up to 4 boxes each)   */
    DXV_DBG_LIST_CELLS = 7,  /* 6 R R x 16 B: begin, end, far radius of every texel's list       */
    DXV_DBG_LIST_ENTRIES = 8,/* stats.list_entries x 16 B entries of the direction-space lists  */
    DXV_DBG_LIST_MIP = 9     /* max-mip of the texels' far radii: 16-bit words, levels R^2 .. 1 x 6 faces */
};

typedef struct dxv_stats {
    uint32_t num_tris, num_verts, num_nodes, tree_height;
    float bound[4];          /* centre.xyz, half extent (Content/Voxelizer.cpp:52-57)          */
    float upload_ms;         /* dxv_set_mesh H2D                                               */
    float prep_ms, sort_ms, hierarchy_ms, refit_ms, build_ms; /* last dxv_build, HIP events   */
    float voxelize_ms;       /* last dxv_voxelize kernel, HIP events on the ctx stream         */
    uint32_t grid_dim, z0, nz; /* last dxv_voxelize                                            */
    uint32_t stack_entries;  /* LDS traversal stack entries per thread of the last launch      */
    float render_ms;         /* last dxv_render kernel, HIP events                                  */
    uint32_t redo_rays;      /* rays of the last launch finished by the deep-stack redo pass       */
    uint32_t row_block;      /* parity rule: rows per side of a wave's block of rows (1, 2 or 4)   */
    float tri_extent;        /* mean triangle box extent along y/z, normalised units               */
    uint32_t list_entries;   /* reference rule: entries of the direction-space lists in use, 0 = tree walk */
    uint32_t list_res;       /* ... texels per cube-map face side                                   */
    float list_ms;           /* ... time of their build (first launch after a build / refit / import) */
    uint32_t plan_bricks;    /* reference rule through a work queue: 4^3-voxel bricks queued as possibly holding a live ray (0 = no queue) */
    uint32_t plan_waves;     /* ... persistent single-wave workgroups the launch ran (what the GPU holds at once)        */
    float plan_ms;           /* ... time of the queue's build on the device, in front of the kernel (launches that built one) */
    uint32_t plan_prepared;  /* 1: the last launch ran a queue PREPARED in Init / by dxv_prepare_launch (plan_bricks = its bricks, plan_waves = the
                                workgroups the hardware dealt out, plan_ms = 0: no queue was built inside the launch)                            */
    float prepare_ms;        /* device time of the context's last dxv_prepare_launch* that built a queue (queue build + its sixteen counts)       */
    float warmup_ms;         /* host time dxv_create spent in the process's one warm-up pass on this device (0: another context paid, or none)  */
} dxv_stats;

/* Create a context on HIP device `device` (Voxelizer::Voxelizer + the device objects that
 * Voxelizer::Init receives from its caller, Content/Voxelizer.cpp:19-42).  Fails when no HIP
 * device is present: there is no CPU fallback.
 * A process's first context on a device sends a four-triangle scene through every step once (about 10 ms): what the runtime sets up
 * lazily -- code object, staging of the first upload, the kernels' first dispatch -- is then paid here and not by the caller's first
 * Init, which it would cost 11 ms instead of 3.3 at 1 M triangles.  Environment DXV_WARMUP=0: no such pass. */
DXV_API int dxv_create(dxv_ctx** out, int device);
/* The warm-up pass on its own (idempotent per process and device; dxv_create calls it unless DXV_WARMUP=0): a host that wants its
 * first dxv_create to be cheap, or wants to time the pass, calls it first.  *ms (may be NULL): host milliseconds it took, 0 when
 * the device was warm already.  dxv_stats.warmup_ms of the context whose dxv_create ran the pass says the same. */
DXV_API int dxv_warmup(int device, float* ms);
DXV_API void dxv_destroy(dxv_ctx* ctx);

/* Last error text of this context ("" when none); with ctx == NULL the last dxv_create error.
 * Mirrors the reference's bool-return convention (XUSG/Core/XUSG.h:12-15) plus a message. */
DXV_API const char* dxv_last_error(const dxv_ctx* ctx);

/* Run all work of this context on an existing hipStream_t (e.g. torch's current stream).  NULL
 * restores the context's own stream.  Replaces the caller-owned command list every reference
 * entry point receives (Content/Voxelizer.h:16-22). */
DXV_API int dxv_set_stream(dxv_ctx* ctx, void* hip_stream);

/* Mesh ingest: replaces XUSG::ObjLoader::Import(file, needNorm=true, needAABB=true)
 * (XUSG/Optional/XUSGObjLoader.cpp:18-40, called at Content/Voxelizer.cpp:46-47).  Output layout
 * is the reference's: vb = numVerts x {float3 pos, float3 nrm} (stride 24), ib = numIndices
 * uint32 (already z-negated / reversed), aabb = {min.xyz, max.xyz}.  Free both with dxv_free. */
DXV_API int dxv_obj_load(const char* path, float** vb, uint32_t* num_verts, uint32_t** ib,
                         uint32_t* num_indices, float aabb[6]);
DXV_API void dxv_free(void* p);

/* Upload vertex/index buffers and derive the normalising bound: replaces createVB/createIB and
 * the bound extraction (Content/Voxelizer.cpp:48-57, :115-138).  The arrays are copied; the
 * caller keeps ownership.  vb: num_verts x 6 floats, ib: 3*num_tris uint32 < num_verts. */
DXV_API int dxv_set_mesh(dxv_ctx* ctx, const float* vb, uint32_t num_verts, const uint32_t* ib,
                         uint32_t num_tris);

/* Build the acceleration structure on the device: replaces Voxelizer::buildAccelerationStructures
 * (BLAS + TLAS with the mesh -> [-1,1]^3 instance transform, Content/Voxelizer.cpp:264-326) with
 * an LBVH: Morton keys -> radix sort -> Karras hierarchy -> bottom-up refit. */
DXV_API int dxv_build(dxv_ctx* ctx);

/* Dynamic meshes ("real-time voxelization", README.md:2): the reference API exposes
 * BuildFlag::ALLOW_UPDATE / PERFORM_UPDATE (XUSG/RayTracing/XUSGRayTracing.h:13-22) but the sample
 * never uses them.  dxv_update_vertices replaces the vertex buffer contents (same vertex count,
 * same index buffer; the normalising bound stays the one of dxv_set_mesh, as the reference's
 * m_bound stays the one of Init) and dxv_refit recomputes the boxes of the existing hierarchy. */
DXV_API int dxv_update_vertices(dxv_ctx* ctx, const float* vb, uint32_t num_verts);
/* The same from a DEVICE buffer (6 floats per vertex, on this context's GPU): the reference's vertex buffer is a GPU
 * resource (createVB, Content/Voxelizer.cpp:115-126), and a mesh animated on the GPU -- skinning, simulation -- never
 * passes through the host.  A device-to-device copy ENQUEUED on the context's stream (dxv_set_stream), nothing else:
 *  - ordering before the copy is the caller's: whatever wrote device_vb must be complete, or ordered before this stream (same
 *    stream, or an event the stream waits on), when the call is made -- a producer kernel still running on another stream
 *    would be read half-written;
 *  - the caller's buffer may be reused as soon as the call returns only if it is written on that same stream, else after
 *    dxv_sync_all;
 *  - positions are not inspected on the host (dxv_set_mesh refuses non-finite ones there): the following dxv_refit counts
 *    triangles with a NaN / Inf vertex while it gathers them and fails if there are any (the hierarchy stays: the next
 *    good update refits again).
 * Neither update waits for launches in flight (they read the scene's triangle records, never the vertex buffer).
 *
 * dxv_refit is the ONE host round trip of a refit-per-frame loop  { dxv_update_vertices_device; dxv_refit;
 * dxv_voxelize_async; }  whose grid is consumed on the GPU:
 *  - launches still in flight that have nothing left to report (reference rule through the lists, parity rule through row
 *    lists) are waited for on the device -- frame 0 shares the context's stream, the other frames' streams through an event --
 *    so the refit's kernels queue up behind a running launch; launches through the tree (whose stack can ask for a redo) are
 *    synchronised on the host first, as every launch was before;
 *  - when the scene had lists (or option lists = 2) their counting pass runs behind the refit's kernels, and the entry total
 *    comes back with the root box in the one synchronisation dxv_refit ends with;
 *  - the next launch builds the lists from that count and is queued behind the build without waiting for it; the one verdict
 *    only the host can act on (a texel with more than 65,535 entries: tree walk) is read when the frame is next
 *    synchronised -- dxv_sync, any dxv_grid_* call, the next dxv_refit -- and a frame launched with lists that fail it is
 *    launched again through the tree there.
 * (frames per second of this loop at 1 M triangles: README.md's table, from the round's evidence run) */
DXV_API int dxv_update_vertices_device(dxv_ctx* ctx, const void* device_vb, uint32_t num_verts);
DXV_API int dxv_refit(dxv_ctx* ctx);

/* Voxelize slices [z0, z0+nz) of a grid_dim^3 grid: replaces Voxelizer::voxelize =
 * DispatchRays(GRID_SIZE, GRID_SIZE*GRID_SIZE, 1) (Content/Voxelizer.cpp:351-369) with grid_dim
 * promoted from the GRID_SIZE macro (:8) to a parameter.  grid_dim must be even (an odd grid has
 * a NaN ray at its centre voxel, hlsl:52).  Output: uint8 {0,1}, id = ((iz-z0)*N + iy)*N + ix
 * (hlsl:64-67), i.e. the alpha channel the only consumer reads (Shaders/PSRayCast.hlsl:108).
 * Synchronous: returns after the kernel has finished and its status word was checked. */
DXV_API int dxv_voxelize(dxv_ctx* ctx, uint32_t grid_dim, int mode, uint32_t z0, uint32_t nz);

/* Same launch without the host synchronisation (for back-to-back timing); pair with dxv_sync,
 * which waits for the stream and reports any deferred kernel error. */
DXV_API int dxv_voxelize_async(dxv_ctx* ctx, uint32_t grid_dim, int mode, uint32_t z0, uint32_t nz);
DXV_API int dxv_sync(dxv_ctx* ctx);

/* Frames in flight: the reference's Voxelizer owns FrameCount = 3 grids and every per-frame call takes a
 * frameIndex (static const uint8_t FrameCount, Content/Voxelizer.h:24; m_grids[FrameCount], :110;
 * Render(pCommandList, frameIndex, ...), :21-22; voxelize(pCommandList, frameIndex), Content/Voxelizer.cpp:351-356),
 * so that the GPU works on one grid while the host still reads another.  dxv_set_frame selects the frame the
 * following dxv_voxelize* / dxv_sync / dxv_grid_* / dxv_texels_download / dxv_render / dxv_get_stats calls refer to
 * (default 0).  Each frame owns its grid, texel image, status words and -- frames 1 and 2 -- an internal stream,
 * so launches of different frames overlap on the GPU; scene, candidate lists and options are shared (an extra frame
 * costs its grid).  Calls that change what the frames read (dxv_set_mesh, dxv_build, dxv_scene_import, dxv_set_stream) first
 * wait for every frame -- dxv_refit too, on the device where it can (see there); dxv_sync_all does only that.
 * dxv_update_vertices does NOT wait: launches read
 * the scene's triangle records, not the vertex buffer, so the next frame's vertices upload (on a stream of the library's own)
 * while the current frame's launch still runs -- dxv_voxelize_async, dxv_update_vertices, dxv_refit (waits for the launch),
 * dxv_voxelize_async, ... is a loop whose PCIe time is hidden. */
#define DXV_FRAME_COUNT 3
DXV_API int dxv_set_frame(dxv_ctx* ctx, uint32_t frame_index);
DXV_API int dxv_sync_all(dxv_ctx* ctx);

/* The work queue of a STATIC scene as Init-time structure.  Which 4^3-voxel bricks of a (grid, partition) can hold a live ray is a
 * pure function of the scene's candidate lists, the grid size and the partition -- exactly like the lists are of the scene -- so it
 * can be built where the reference builds everything its frames trace through: once, in Init (Content/Voxelizer.cpp:73, :264-326),
 * leaving a frame ONE dispatch (:351-369).  dxv_prepare_launch builds that queue now (k_plan_bricks once, 0.02 ms at 512^3, plus one
 * host round trip for its sixteen counts) for slices [z0, z0 + nz) of a grid_dim^3 grid -- _interleaved: for rank's share of the
 * block-cyclic partition -- and keeps it with the context, for all its frames, until the scene or its lists change (dxv_set_mesh,
 * dxv_build, dxv_refit, dxv_scene_import, a rebuild of the lists: all drop it).  Every later dxv_voxelize* of that grid_dim and
 * partition in reference mode is then: the grid cleared (only the bricks nobody runs, by workgroups of the same dispatch) + one
 * workgroup per queued brick dealt out by the hardware.  Every voxel is still written in every launch and nothing a launch reads
 * was left behind by another LAUNCH; what is read was left by Init.  Launches of partitions that were not prepared, of scenes
 * without lists (over the caps: tree walk) and of refitted meshes build their queue themselves as before (option plan = 2).
 * Builds the lists first if the scene has none yet (as dxv_build_lists_for_grid does).  Up to 16 partitions per context (least
 * recently used goes).  Not an error when the scene cannot have lists: nothing is prepared then (dxv_stats.prepare_ms = 0). */
DXV_API int dxv_prepare_launch(dxv_ctx* ctx, uint32_t grid_dim, uint32_t z0, uint32_t nz);
DXV_API int dxv_prepare_launch_interleaved(dxv_ctx* ctx, uint32_t grid_dim, uint32_t rank, uint32_t world, uint32_t zblock);

/* Load-balanced multi-GPU partition: the grid's Z axis is cut into blocks of `zblock` slices dealt
 * round-robin to `world` ranks; this call voxelizes the grid_dim/world slices of `rank` (global
 * slice of local slice lz: (lz / zblock * world + rank) * zblock + lz % zblock, ascending) into a
 * compact grid_dim*grid_dim*(grid_dim/world)-byte grid.  Requires zblock to be a power of two and
 * grid_dim % (zblock*world) == 0.
 * Contiguous slabs (dxv_voxelize with z0/nz) starve the GPUs that own empty space; see DESIGN.md. */
DXV_API int dxv_voxelize_interleaved(dxv_ctx* ctx, uint32_t grid_dim, int mode, uint32_t rank, uint32_t world,
                                     uint32_t zblock);
DXV_API int dxv_voxelize_interleaved_async(dxv_ctx* ctx, uint32_t grid_dim, int mode, uint32_t rank,
                                           uint32_t world, uint32_t zblock);

/* Result access.  The grid stays resident on the device (the reference never reads it back,
 * it is consumed by the ray-cast pass on the GPU); download is for callers that want it.
 * dxv_grid_device_ptr: the selected frame's grid after dxv_sync; the caller may also write through it, now or later
 * (see dxv_grid_device_ptr_ro below for what that costs). */
DXV_API void* dxv_grid_device_ptr(dxv_ctx* ctx);
/* The same pointer for READING only (the consumer of the reference's grid SRV, Content/Voxelizer.cpp:371-399).
 * dxv_grid_device_ptr marks the frame as written to by the caller for as long as that pointer lives (until the grid is
 * reallocated by a larger launch): every later launch into the frame then clears the whole grid first instead of keeping
 * the zeros of its own last launch -- a caller who caches the pointer and writes through it later is safe.  This accessor
 * leaves the frame alone. */
DXV_API const void* dxv_grid_device_ptr_ro(const dxv_ctx* ctx);
DXV_API size_t dxv_grid_bytes(const dxv_ctx* ctx);
DXV_API int dxv_grid_download(dxv_ctx* ctx, uint8_t* host, size_t bytes);
/* The same grid as one BIT per voxel, packed on the device before it crosses PCIe (8x fewer
 * bytes): voxel 8j+i of the last launch's slab is bit i of byte j; bytes = dxv_grid_packed_bytes
 * = ceil(dxv_grid_bytes / 8).  What the reference's consumer reads is this one bit (alpha,
 * Shaders/PSRayCast.hlsl:108). */
DXV_API size_t dxv_grid_packed_bytes(const dxv_ctx* ctx);
DXV_API int dxv_grid_download_packed(dxv_ctx* ctx, uint8_t* host, size_t bytes);
/* Number of solid voxels of the last grid, reduced on the device. */
DXV_API int dxv_grid_count(dxv_ctx* ctx, uint64_t* solid);

/* The same grid as the reference's R10G10B10A2_UNORM texels (float4(Normal, 1), hlsl:83-84,
 * Content/Voxelizer.cpp:65): enable before dxv_voxelize to also fill a uint32 texel per voxel
 * (0 where the shader writes nothing).  Reference mode only. */
DXV_API int dxv_enable_texels(dxv_ctx* ctx, int enable);
DXV_API int dxv_texels_download(dxv_ctx* ctx, uint32_t* host, size_t bytes);

/* The grid's consumer, for visual A/B against the reference: Voxelizer::UpdateFrame + renderRayCast
 * (Content/Voxelizer.cpp:81-106, :371-399; Shaders/VSScreenQuad.hlsl + PSRayCast.hlsl: 128-step
 * march through the grid's alpha with a 32-step light march).  eye and view_proj (row-major, row
 * vectors: v' = v * M, as DirectXMath stores them) are what the app passes to UpdateFrame
 * (DXRVoxelizer.cpp:249-254); pos_scale = {x, y, z, scale} or NULL for the default {0,0,0,1}.
 * Renders the whole grid of the last dxv_voxelize into width*height R8G8B8A8 texels on the host. */
DXV_API int dxv_render(dxv_ctx* ctx, const float eye[3], const float view_proj[16], const float pos_scale[4],
                       uint32_t width, uint32_t height, uint8_t* rgba_host);

/* Multi-GPU: the built scene (nodes + triangle data) as one relocatable device blob, so that
 * rank 0 builds once and the host layer broadcasts it (RCCL over xGMI) to the other ranks.
 * export copies the blob into caller-provided DEVICE memory; import adopts a blob from DEVICE
 * memory as if dxv_set_mesh + dxv_build had run here. */
/* STATIC and DYNAMIC scenes -- the two ways through this header:
 *   static   dxv_set_mesh; dxv_build; dxv_build_lists_for_grid;  then dxv_voxelize* per frame.  The reference's case
 *            (Content/Voxelizer.cpp:73: Init builds everything the frames trace through) and what the host mirrors' Init does
 *            (include/dxv_voxelizer.hpp, dxrvoxelizer_amd/voxelizer.py): LBVH and candidate lists exist when Init returns, and
 *            every launch is the same launch -- it builds its work queue, clears its grid and runs (option plan = 2): a scene's
 *            first Voxelize costs what its hundredth costs.
 *   dynamic  dxv_set_mesh; dxv_build;  then per frame dxv_update_vertices(_device); dxv_refit; dxv_voxelize_async.  The lists of
 *            a mesh that is being refitted are built inside each frame's launch, on the coarser map (option lists, listres).
 * A caller who does neither (dxv_build, then dxv_voxelize) is treated as dynamic until the scene is launched a second time without
 * a refit in between: option lists says when the lists are built then.
 *
 * The candidate lists of the reference rule (direction-space lists): dxv_build_lists builds them now.  A
 * scene exported after that carries them as two more sections of the blob, and the importing contexts adopt them instead of
 * building their own. */
DXV_API int dxv_build_lists(dxv_ctx* ctx);
/* ... on the map the launches of a static scene move to (the 512 map for scenes of 20,000 triangles or more, at every grid size):
 * the exporting rank builds that map before dxv_scene_export, so that the importing ranks do not each rebuild it at their second
 * launch.  Lists that cannot be had on the finer map leave the ones there are.  grid_dim != 0 (even, <= 2048): the whole grid of
 * that size is prepared as well (dxv_prepare_launch(ctx, grid_dim, 0, grid_dim)) -- what the host mirrors' Init does with a grid
 * hint; 0: lists only.  Option lists = 0: nothing is built (the caller asked for tree walks). */
DXV_API int dxv_build_lists_for_grid(dxv_ctx* ctx, uint32_t grid_dim);
/* The same for the parity rule's row lists (option plists): built now instead of at the scene's second parity launch; a scene
 * exported after that carries them too (33 + 72 MB at 1 M triangles), and an importing context adopts them. */
DXV_API int dxv_build_parity_lists(dxv_ctx* ctx);
DXV_API size_t dxv_scene_bytes(const dxv_ctx* ctx);
DXV_API int dxv_scene_export(dxv_ctx* ctx, void* device_dst, size_t bytes);
DXV_API int dxv_scene_import(dxv_ctx* ctx, const void* device_src, size_t bytes);
/* Wrapping 64-bit sum of the 8-byte words of a blob in DEVICE memory of this context's GPU: a host that moves blobs between GPUs
 * compares the receivers' sums with the sender's before it imports anything (once per mesh; include/dxv_multi.hpp, slabs.py). */
DXV_API int dxv_scene_checksum(dxv_ctx* ctx, const void* device_blob, size_t bytes, uint64_t* sum);

DXV_API int dxv_get_stats(const dxv_ctx* ctx, dxv_stats* out);

/* Tuning knobs (kernel variant selection etc.); unknown keys fail.  None changes a result.
 *   brick  0..7   voxels per workgroup (default 4 = 4x4x4, one wavefront)
 *   stack  0|8..64  LDS column entries per thread; 0 (default) = adaptive from stack0
 *   stack0 8..64  starting depth of the adaptive column (default 20)
 *   queue  0|1    postponed-leaf walk (default 1)
 *   wide   0|1|2  reference rule over four-box nodes (1) or on wave-uniform visits only (2, default);
 *                 0 = binary nodes only and no four-box scene section
 *   rows   0|1    parity rule: one tree walk per grid row (default 1)
 *   rowblock 0|1|2|4  ... per row (1), per 2 x 2 or 4 x 4 rows; 0 (default) decides by triangle size
 *   refit  0|1|2  box merge of dxv_build and dxv_refit: min/max pyramid over the leaf order (1,
 *                 default), level sweeps (2), one atomic pass (0)
 *   deferboxes 0|1  dxv_refit while lists are wanted (lists != 0, refit = 1): stop at the pyramid -- triangle records and root
 *                 box are current, the node boxes are written when a tree walk, an export or a debug download first needs them
 *                 (1, default: a refit at 1 M triangles 0.14 -> 0.07 ms); 0 = every refit writes them
 *   lists  0|1|2  reference rule through direction-space lists (dxv_dirmap.h) or the tree walk (0).  The
 *                 lists are built from the scene's triangle records (0.59 ms at 1 M triangles): at the second
 *                 launch after a build / refit / import, or at the first when that launch is large enough for
 *                 the build to pay for itself at once -- 2^26 voxels or more and the estimate after the build's
 *                 counting pass says so (1, default: a mesh refitted every frame takes whichever is faster), or
 *                 always at the first (2); scenes whose lists would exceed 256 entries per triangle + 64 M or
 *                 65,535 entries in one texel keep the tree walk (stats.list_entries = 0)
 *   dispatch 0|1|2  a launch through a KEPT work queue (plan = 1, same lists / partition / buffers as the frame's last launch)
 *                 whose sixteen counts a dxv_sync has read since it was built: one workgroup per queued brick dealt out by the
 *                 hardware (1, default; 2: only for partitions of up to 2^25 voxels) instead of persistent waves (0).  A launch that
 *                 builds its queue (every launch under plan = 2) does not know its size and uses the persistent waves.
 *   listres 0|16..4096  texels per cube-map face side of the lists (power of two).  0 = automatic: 128 below 20,000 triangles,
 *                 256 up to 3 M, 512 beyond -- and the 512 map for every scene of 20,000 triangles or more that is presumed
 *                 STATIC: built by dxv_build_lists / lists = 2 on a scene that has not been refitted, or launched a second time
 *                 without a refit in between (one rebuild).  A mesh that is being refitted, and a first launch whose build
 *                 must pay for itself at once, keep the base map.  Deep scenes (over 32 entries per texel) take coarser maps.
 *   plists 0|1|2    parity rule through row lists of the (y, z) plane: 1 (default) from a scene's second parity launch,
 *                 2 from the first, 0 = always walk the tree; plistres 0|16..4096: texels per side of their grid
 *   plan   0|1|2  lists kernel through a work queue: only the 4^3-voxel bricks that can hold a live ray are run (decided per
 *                 brick on the device, in front of the kernel in the same stream: the brick's footprint in direction space and
 *                 its smallest start radius against a max-mip of the lists' far radii; no host round trip).  The kernel that builds
 *                 the queue clears the partition's grid as well (option fuse), deals the bricks to eight queues -- the bricks that
 *                 can look into a list that is long for the scene at the front (planheavy), every XCD running an equal share of
 *                 all eight -- and persistent waves take them from there.
 *                 2 (default): built and cleared on every launch -- NOTHING is carried from launch to launch: a scene's first,
 *                 second and hundredth launch cost the same and write every voxel;
 *                 1: queue and zeros are kept while the frame's next launch is the same one (same lists, partition, buffers) --
 *                 for a caller that voxelizes a static scene into the same frame again and again (the reference's own loop,
 *                 Content/Voxelizer.cpp:108-113): -13 % per launch at 512^3, -25 % on a rank's share at 8 ranks, and the launch
 *                 goes through the hardware's dispatcher (option dispatch);
 *                 0: no queue, brick box around the scene in Morton order
 *   coop   0|1    the lists kernel: when at most two lanes of a wave are still scanning and the first has 24 entries or more ahead, the whole
 *                 wave scans that ray's list, one entry per lane and round (1, default) -- a brick is as long as its longest list, and a
 *                 short launch (a rank's share) cannot end before its longest brick; 0: every lane scans alone
 *   listedwaves 0|8..32  the hardware-dispatched lists kernel (prepared and kept queues, no texel image) fits eight waves per SIMD: 32
 *                 single-wave workgroups per CU.  0 (default): all 32 when the grid side is at least 3/4 of the lists' map side (a brick's rays
 *                 fall on neighbouring texels: 512^3 -9 %, 1024^3 -12 % against seven waves), else 28 (256^3 on the 512 map: a brick is spread
 *                 over many texels and the eighth wave costs 7 %); 8 .. 32: held at so many workgroups per CU by LDS the launch does not use
 *   farmap 0|1    launches over the brick box (tree walks -- lists = 0, dynamic first launches, scenes over the lists' caps -- and plan = 0):
 *                 every workgroup makes the queue's brick test itself and a brick none of whose rays can reach a triangle is zeroed and
 *                 left (1, default); the test reads the lists' max-mip or, for a scene without lists, a far-radius map of the triangles'
 *                 own footprints made at the scene's SECOND such launch (0.13 ms at 1 M triangles: a mesh refitted every frame never pays
 *                 it); 0: every brick is walked
 *   prepared 0|1  launches of a partition that dxv_prepare_launch* prepared use its queue (1, default) or build their own (0)
 *   prepclear 0|1|2|3  how a launch through a prepared queue clears its grid: 0 = a clear kernel in front of the brick kernel; 1 / 2 / 3 =
 *                 only the bricks that are not queued, by workgroups in front of / behind / spread evenly between the bricks' in the
 *                 SAME dispatch
 *   planregion 0|6|7|8  log2 of the run of consecutive Morton bricks that goes to one queue (0 = by partition size)
 *   planheavy 0..65535  a brick that can look into a list of more entries than this goes to the front of its queue (0, default: one
 *                 and a half times the scene's mean at the level of a brick's patch of texels; 65535: no brick does)
 *   fuse   0|1    the queue build clears the grid (1, default) or memsets stand in front of it (0)
 *   queuewaves 0..2^20  persistent waves of a launch through the queue (0, default: what the device holds at once -- 7 per SIMD -- or
 *                 five / four sevenths of that for a mesh of 500,000 triangles or more on a grid of at most half / a quarter of its
 *                 lists' map: 256^3 on the 512 map, nothing carried, -13 % for 1 M-triangle meshes)
 *   queueheads 1|2|4|8  heads per queue the persistent waves draw from (default 8)
 *   queuemin 0..4096  persistent waves beyond one per this many bricks of an XCD's share leave before they touch the queue (0,
 *                 default: all stay; a caller's knob from before queuewaves picked its own default on coarse grids)
 *   sortbits 0|8..11 (+16, +32)  diagnostic, process-wide: widest digit of the builds' radix sort (0, default: 10 or 11 bits -- three
 *                 passes for the LBVH's keys, four for the lists'); +16 / +32: tiles of 4 / 16 waves whatever the size.  Same results.
 *   events 0|1    bracket every launch with two HIP events for stats.voxelize_ms (default 1); 0 for a caller that times its own
 *                 loop of back-to-back launches (the events cost ~8 us of stream time per launch)
 *   skipempty 0|1 dxv_render: skip the samples of empty 8^3 bricks (default 1; same image)
 *   morton 0|1, region 0..24, subbox 0|1   brick order, bricks per XCD region (log2), partial launch */
DXV_API int dxv_set_option(dxv_ctx* ctx, const char* key, int64_t value);

/* Test hook: the superset claim of the direction-space lists, checked for every voxel of slices [z0, z0 + nz) of a grid_dim^3
 * grid on the device:
 * every triangle the canonical triangle step accepts for a ray (found by an LBVH walk without distance culling) must be
 * selectable from that ray's texel list.  out[0] = accepted (ray, triangle) pairs, out[1] = violations (must be 0),
 * out[2 + 2k], out[3 + 2k] = voxel id and triangle slot of the first 16 violations.  Builds the lists if needed. */
DXV_API int dxv_debug_list_check(dxv_ctx* ctx, uint32_t grid_dim, uint32_t z0, uint32_t nz, uint64_t out[34]);

/* Test hook: the per-triangle class of the normal test (most hits of the reference rule are answered from two bits of the hit
 * triangle's record instead of its interpolated normal) against the predicate itself: for every voxel of
 * slices [z0, z0 + nz) of a grid_dim^3 grid the closest hit is found by the plain LBVH walk, and when its triangle carries a
 * class the canonical predicate (hlsl:137-138) is evaluated and compared.  out[0] = hits on classified triangles, out[1] =
 * disagreements (must be 0), out[2] = all hits, out[3 + 2k], out[4 + 2k] = voxel id and triangle slot of the first 15. */
DXV_API int dxv_debug_class_check(dxv_ctx* ctx, uint32_t grid_dim, uint32_t z0, uint32_t nz, uint64_t out[34]);

/* Test hook: the ray set-up divides with a scale-free sequence that shares the denominator's reciprocal (csrc/dxv_math.h: div_by) where the
 * canonical rules say `/`; for the operands a voxel origin produces the two are the same bits.  Checked here for EVERY voxel origin of
 * every even grid size n_first, n_first + 2, ... n_last (<= 2048): origin, cube-map point, direction, 1 / direction, shear constants, 14
 * words per voxel against IEEE quotients computed beside them.  out[0] = voxels checked, out[1] = voxels with a differing word (must be
 * 0), out[2 + k] = id of the first 6 (of the grid they occurred in).  All grids up to 2048^3: 2.2 x 10^12 voxels, about a minute. */
DXV_API int dxv_debug_division_check(dxv_ctx* ctx, uint32_t n_first, uint32_t n_last, uint64_t out[8]);

/* Test hook: the brick test of the launches over the brick box (tree walks, plan = 0; option farmap): for every 4^3-voxel brick of slices
 * [z0, z0 + nz) the test the kernel makes, and for every voxel of a brick it calls dead a plain LBVH walk.  lists_mip = 0: against
 * the far-radius map of the triangles' own footprints (what a scene without lists uses; made if need be), 1: against the max-mip of
 * the scene's lists.  out[0] = bricks, out[1] = bricks called dead, out[2] = their rays walked, out[3] = rays among them that hit
 * something (must be 0), out[4 + k] = voxel id of the first 8. */
DXV_API int dxv_debug_far_check(dxv_ctx* ctx, uint32_t grid_dim, uint32_t z0, uint32_t nz, int lists_mip, uint64_t out[12]);

/* Test hook: the work queue's claim -- no live ray in a brick that is not queued -- checked exhaustively on the device for the
 * partition of the current frame's last launch (which must have gone through a queue): every voxel makes exactly the first-step
 * decision of the kernel (origin beyond the root box / texel empty / start beyond the texel's far radius -> miss).
 * out[0] = live voxels, out[1] = bricks with a live voxel, out[2] = queued bricks, out[3] = live bricks that are NOT queued (must
 * be 0), out[4] = bricks queued more than once (must be 0), out[5 + k] = brick word (bx | by << 10 | bz << 20) of the first 11. */
DXV_API int dxv_debug_plan_check(dxv_ctx* ctx, uint64_t out[16]);

/* Give back what the context keeps only to make the next build faster: the list build's scratch (up to 16 GiB per buffer
 * after a 10 M-triangle scene), the LBVH build's scratch when no refit can follow (imported scenes), the memory of prepared queues
 * whose lists are gone.  Nothing a launch reads. */
DXV_API int dxv_trim(dxv_ctx* ctx);

/* Test hook: copy an internal device array to the host (enum above). */
DXV_API int dxv_debug_download(dxv_ctx* ctx, int what, void* host, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* DXV_H */
