// dxv_device.h -- declarations shared by the HIP translation units of libdxv.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dxv_types.h"
#include "dxv_trace.h"

namespace dxv {

// radix_sort.hip
// plan: the diagnostic override (option sortbits) as ONE caller snapshot of radix_sort_plan(), -1: read it now
hipError_t radix_sort_keys_bits(uint64_t* keys, uint64_t* tmp, uint32_t n, uint32_t* hist, int loBit, int numBits, uint64_t** result,
                                hipStream_t s, int plan = -1);
int radix_sort_passes(uint32_t n, int numBits, int plan = -1);     // how many times that sort swaps keys and tmp
int radix_sort_plan();                              // the process-wide override as it stands (a build asks once)
uint32_t radix_sort_hist_words(uint32_t n);
void radix_sort_set_plan(int v);                    // diagnostic (option sortbits)

// lbvh.hip -- device-side build of the scene blob.
struct BuildBuffers {
    const float* vb;        // V x 6
    const uint32_t* ib;     // 3T
    uint32_t T, V;
    float bound[4];
    uint64_t* keys;         // T
    uint64_t* keysTmp;      // T
    uint32_t* hist;         // radix_sort_hist_words(T)
    uint32_t* parents;      // (T-1) internal + T leaf words: (parent << 1) | side
    uint32_t* flags;        // T-1 arrival counters (atomic refit) / ready flags (sweep refit)
    uint32_t* flags2;       // T-1: far end of every node's run of leaves (k_hierarchy -> pyramid refit)
    void* pyramid;          // pyramid_slots(T) x 28 B: min/max pyramid over the leaf boxes + deepest leaf, or NULL (sweeps)
    uint32_t* rootInfo;     // 8 words: rootLo[3], rootHi[3] (float bits), height, done
    Node* nodes;            // max(T-1,1), exact boxes
    Node32* nodes32;        // max(T-1,1), traversal copy
    Node64* nodes64;        // max(T-1,1), wide traversal copy
    TriPos* triPos;         // T
    TriNrm* triNrm;         // T
    bool deferCopies;       // leave the four-box copy (nodes64) stale: lbvh_traversal_copies brings it up to date when a walk needs it (nodes32 always is)
    bool deferBoxes;        // lbvh_refit only (pyramid refit): stop after the min/max pyramid -- root box from its top, node boxes left stale
                            // until lbvh_refit_boxes; the lists are built from the triangle records alone
};
struct BuildTimes { float prep, sort, hierarchy, refit; };
// refitMode: 0 = one pass, bottom-up with per-node arrival counters; 1 = level-synchronous sweeps
hipError_t lbvh_build(const BuildBuffers& b, int refitMode, hipStream_t s, hipEvent_t ev[5]);
uint32_t pyramid_slots(uint32_t T);
hipError_t lbvh_refit(const BuildBuffers& b, int refitMode, uint32_t treeHeight, hipStream_t s, hipEvent_t ev[2]);
hipError_t lbvh_traversal_copies(const BuildBuffers& b, hipStream_t s);   // nodes32 -> nodes64: the four-box copy the wide tree walks read
hipError_t lbvh_refit_boxes(const BuildBuffers& b, hipStream_t s);        // what a refit with deferBoxes left undone: node boxes from the pyramid it made, then the copies

// dirmap.hip -- direction-space lists of the reference rule (dxv_dirmap.h)
struct DirEntry;
struct DirRecord;
struct DirCell;
hipError_t dirmap_count(const TriPos* triPos, uint32_t T, uint32_t R, DirRecord* rec, uint32_t* counts, uint32_t* pairs, uint32_t* wideList, unsigned long long* total, hipStream_t s, uint32_t stride = 1);
hipError_t dirmap_fill(uint32_t T, uint32_t R, const DirRecord* rec, const uint32_t* counts, const uint32_t* pairs, const unsigned long long* total,
                       uint32_t* offsets, uint32_t* sums, uint64_t* keys, uint64_t* keysTmp, uint32_t* hist, uint32_t n, DirCell* cells, DirEntry* entries, uint32_t* longest,
                       hipStream_t s);

// max-mips of the texels' far radii and entry counts (dxv_dirmap.h: dm_mip_max), dm_mip_buffer_words(R) 16-bit words: what the launch's work queue is probed against
hipError_t dirmap_mip(const DirCell* cells, uint32_t R, uint16_t* mip, hipStream_t s);

// the far-radius map of a scene WITHOUT lists (a tree walk's brick test): far32 = 6 R R words of scratch, cells = 6 R R, mip = dm_mip_buffer_words(R)
hipError_t dirmap_far(const TriPos* triPos, uint32_t T, uint32_t R, uint32_t* far32, DirCell* cells, uint16_t* mip, hipStream_t s);

// consistency of a list section that arrived in a blob (dxv_scene_import): out[0] = cells whose range leaves the entries,
// out[1] = entries whose triangle slot is >= T
hipError_t dirmap_validate(const DirCell* cells, uint32_t R, const DirEntry* entries, uint32_t n, uint32_t T, uint32_t* out, hipStream_t s);

hipError_t parity_lists_validate(const uint32_t* cells, uint32_t R, const uint32_t* entries, uint32_t n, uint32_t T, uint32_t* out, hipStream_t s);
hipError_t parity_lists_total(const TriPos* triPos, uint32_t T, uint32_t R, unsigned long long* total, hipStream_t s);
hipError_t parity_lists_fill(const TriPos* triPos, uint32_t T, uint32_t R, uint32_t* counts, uint32_t* offsets, uint32_t* sums, uint32_t* cells,
                             uint32_t* entries, hipStream_t s);

// traverse.hip
struct VoxelizeParams {
    SceneView scene;
    uint8_t* grid;          // N*N*nz bytes
    uint32_t* texels;       // optional N*N*nz words
    uint32_t* status;       // status[0] bit 0: a ray could not be finished (redo list full, or the redo pass itself ran out)
                            // status[1 + parity], status[2 - parity]: rays on the redo list of this / the next launch
    uint64_t* redo;         // voxel ids (local index into grid) whose LDS column was too small: finished by k_voxelize_redo
    uint32_t redoCap;       // entries of `redo`
    uint32_t redoParity;    // which of the two counters this launch appends to
    uint32_t N, z0, nz;     // nz = slices written by this launch (local index lz in [0, nz))
    uint32_t zBlock, zPeriod; // global slice of lz: z0 + (lz / zBlock) * zPeriod + lz % zBlock
    uint32_t zShift;          // log2(zBlock) when the partition is block-cyclic (zBlock is a power of two)
    uint32_t superX, superY;  // brick super-blocks per axis (filled by the launcher)
    uint32_t nbx, nby, nbz;   // bricks launched per axis and their offset: only the part of the grid
    uint32_t bx0, by0, bz0;   //   that the root early-out cannot clear (the rest is memset to 0)
    int mode;
    uint32_t morton;        // 1: Morton brick order (default), 0: linear x,y,z order
    uint32_t mortonBits;    // filled by the launcher
    uint32_t regionBits;    // log2 of the bricks per XCD region
    uint32_t queued;        // 1: postponed-leaf traversal (default), 0: leaves tested on the spot
    uint32_t subbox;        // 1: launch only bricks the root early-out cannot clear (default)
    uint32_t wide;          // 1: reference rule walks the wide nodes (default when the stack bound allows)
    uint32_t lists;         // 1: reference rule reads the direction-space lists of p.scene (no tree walk)
    uint64_t* clearSig;     // host word of the frame (or NULL): signature of the partial launch whose memset the grid still carries -- the same launch again skips the memset
    uint32_t ablate;        // timing-only builds of the lists kernel (wrong grids; tools/ablate.py), 0 = the real kernel
    uint32_t* queue;        // work queue of the lists kernel (traverse.hip): the header this launch uses (64 heads, 8 lengths, every word in a line of its own)
    uint32_t* queueSlots;   // ... its 8 x queueCap brick words
    uint32_t* queueZero;    // ... the frame's OTHER header, cleared by k_plan_bricks for the launch that builds the next queue (or NULL)
    uint32_t queueCap;
    uint32_t planRegionBits; // log2 of the run of consecutive Morton bricks that goes to one queue (6, 7 or 8)
    uint32_t planHeavy;     // a brick that can look into a list of more entries than this goes to the front of its queue; 0: the scene's own "long list" (k_dm_heavy_thresholds)
    uint32_t planClear;     // 1: k_plan_bricks also clears the grid (and the texel image): no memset in front of it
    uint32_t queueWaves;    // persistent waves to launch; 0 = queueSevenths / 7 of what the device holds at once
    uint32_t queueSevenths; // (dxv_policy.h: queue_waves_sevenths; 0 = 7)
    uint32_t queueHeads;    // heads per queue the persistent waves draw from: 1, 2, 4 or 8
    uint32_t queueMinBricks; // persistent waves beyond one per this many bricks of an XCD's share leave at once (0: all stay)
    const uint16_t* mip;    // max-mip of the lists' far radii (dxv_dirmap.h), what k_plan_bricks probes the bricks against
    uint32_t mipR;          // brick-box launches (k_voxelize, 4^3 bricks, reference rule): the map `mip` was made on -- every workgroup makes
                            // the queue's brick test itself and a brick that cannot hold a live ray is zeroed and left; 0: no test
    uint32_t listedWaves;   // k_voxelize_listed without the texel image fits eight waves per SIMD (64 VGPRs, 32 workgroups per CU): 8 .. 32 = held at so many
                            // workgroups per CU by LDS it does not use; 0: by grid and map (listed_lds_pad, traverse.hip; option listedwaves)
    uint32_t* liveMask;     // k_plan_bricks: one bit per brick of the partition, id (bz nbx + by) nbx + bx, set for every queued brick (or NULL):
                            // what the clear of a launch through a PREPARED queue reads (only the bricks nobody runs are zeroed)
};
hipError_t launch_voxelize(const VoxelizeParams& p, int brickShape, int stackEntries, hipStream_t s);
// header of a queue: 64 heads (eight per queue: head h of queue x hands out the slots k = h mod 8 of that queue; head
// number 8 x + h), then the eight lengths, every word in a 256-byte line of its own.  Queue memory of a frame: TWO headers, then
// the slots: a launch that builds a queue takes the header the last build did not use -- all zero, because that build's
// k_plan_bricks cleared it (and the allocation cleared both) -- so no memset stands between a launch and its queue build.
// A queue is filled from both ends: the bricks that can look into a long list (dm_box_max_count: the few that take several times
// the mean) from slot 0 upwards, all others from slot cap - 1 downwards; its items are numbered heavy first (queue_slot), so a
// launch starts with its long bricks and ends with ordinary ones.  Header words per queue: eight heads, the number of heavy and
// the number of light bricks.
constexpr uint32_t kQueueHeaderWords = 5632u;
constexpr uint32_t kQueueSlotsAt = 2u * kQueueHeaderWords;
DXV_HD constexpr uint32_t queue_head_word(uint32_t x, uint32_t h) { return 64u * (1u + 8u * x + h); }
DXV_HD constexpr uint32_t queue_len_word(uint32_t x) { return 64u * (65u + x); }      // light bricks of queue x
DXV_HD constexpr uint32_t queue_heavy_word(uint32_t x) { return 64u * (73u + x); }    // heavy bricks of queue x (the sixteen words lie behind one another: one copy for dxv_sync)
DXV_HD constexpr uint32_t queue_slot(uint32_t k, uint32_t heavy, uint32_t cap) { return k < heavy ? k : cap - 1u - (k - heavy); }   // item k of a queue
// work queue of the lists kernel with 4 x 4 x 4 bricks (traverse.hip): built on the device in front of the launch
uint32_t plan_layout(VoxelizeParams& p);           // fills the brick-order fields for the whole partition, returns its bricks
uint32_t plan_region_bits(uint32_t N, uint32_t nz); // the run length (log2 bricks) a partition of this size deals to its queues
size_t plan_queue_words(uint32_t N, uint32_t nz, uint32_t* capOut);     // 32-bit words of queue memory for a partition; *capOut = words per XCD queue
hipError_t plan_build(const VoxelizeParams& p, hipStream_t s);          // k_plan_bricks into the (zero) header p.queue; p.queueSlots, p.queueCap, p.mip set
// rebuild: grid cleared + queue built in front of the kernel; else only the queue heads are reset (same launch as before into the same buffers)
// (planEvents: two events recorded around the queue build of a rebuilding launch, or NULL)
// listedLens: the eight lengths of a kept queue and how many of each are heavy (16 words) as the host last read them (one workgroup per item, dealt out by the hardware), or NULL (persistent waves)
hipError_t launch_voxelize_queue(const VoxelizeParams& p, bool rebuild, uint32_t* wavesOut, hipEvent_t* planEvents, const uint32_t* listedLens, hipStream_t s);
// A launch through a queue that was PREPARED for (lists, grid, partition) -- built ONCE, in Init or by dxv_prepare_launch, like the lists
// it is a pure function of: the host knows the sixteen counts, so the hardware deals the bricks out (k_voxelize_listed, one workgroup
// per queued brick), and the grid is cleared inside the launch.  p.queueSlots / p.queueCap: the prepared queue's; lens: its sixteen
// counts (eight lengths, of which heavy); live: its bit per brick (NULL: clearMode 0).
//   clearMode 0: a kernel of its own clears the whole grid (16-byte non-temporal stores) in front of the brick kernel;
//             1 / 2 / 3: ONE dispatch -- workgroups in front of (1), behind (2) or spread evenly between (3) the bricks' zero exactly the
//             bricks that are not queued (every voxel is written once per launch, by the brick that owns it or by the clear; needs
//             N % 16 == 0).
hipError_t launch_voxelize_prepared(const VoxelizeParams& p, const uint32_t lens[16], const uint32_t* live, int clearMode, uint32_t* wavesOut, hipStream_t s);
size_t plan_live_words(uint32_t N, uint32_t nz);   // 32-bit words of a partition's brick mask
// test hook: every voxel's first-step decision against the queue; bits: one per brick of the partition, out: 16 words
hipError_t launch_plan_check(const VoxelizeParams& p, uint32_t* bits, unsigned long long* out, hipStream_t s);
hipError_t launch_voxelize_redo(const VoxelizeParams& p, hipStream_t s);   // finishes the rays on p.redo with a full-depth stack
int stack_round_up(int want);
int stack_for_brick(int brickShape, int want);   // the column depth compiled for this brick shape that is >= want
hipError_t launch_parity_rows(const VoxelizeParams& p, int rowBlock, hipStream_t s);   // parity mode: one walk per row run (1) or per 2 x 2 rows (2)
hipError_t launch_list_check(const VoxelizeParams& p, unsigned long long* out, hipStream_t s);   // test hook: superset claim of the lists (slices [p.z0, p.z0 + p.nz))
hipError_t launch_division_check(uint32_t N, unsigned long long* out, hipStream_t s);             // test hook: the ray set-up's divisions against `/` for every voxel of an N^3 grid (out: 10 words, zeroed by the caller)
hipError_t launch_far_check(const VoxelizeParams& p, unsigned long long* out, hipStream_t s);    // test hook: the brick test of the brick-box launches (p.mip, p.mipR; slices [p.z0, p.z0 + p.nz))
hipError_t launch_class_check(const VoxelizeParams& p, unsigned long long* out, hipStream_t s);  // test hook: per-triangle class of the normal test against the predicate
hipError_t launch_count(const uint8_t* grid, size_t n, unsigned long long* out, hipStream_t s);
hipError_t launch_checksum(const void* buf, size_t bytes, unsigned long long* out, hipStream_t s);   // wrapping sum of the buffer's 64-bit words
hipError_t launch_pack_bits(const uint8_t* grid, size_t n, uint8_t* packed, hipStream_t s);
int num_brick_shapes();

// raycast.hip
struct RayCastCB;
hipError_t launch_raycast(const RayCastCB& cb, const uint8_t* grid, uint32_t N, uint32_t width, uint32_t height,
                          uint32_t* rgba8, uint8_t* empty, hipStream_t s);
size_t empty_brick_bytes(uint32_t N);

} // namespace dxv
