// traverse.hip -- the voxelisation kernels: one thread per voxel, one ray per thread.
//
// Replaces DispatchRays(GRID_SIZE, GRID_SIZE*GRID_SIZE, 1) with raygenMain / closestHitMain /
// missMain (Content/Voxelizer.cpp:366-368, Content/Shaders/DXRVoxelizer.hlsl:58-85, :132-148).
//
// Launch shape: a workgroup owns a BX x BY x BZ brick of voxels (default 4x4x4 = one wavefront;
// lanes with neighbouring origins and near-parallel radial rays walk the same nodes).  Bricks are
// numbered along a Morton curve and dealt to the 8 XCDs in runs of 2^regionBits bricks, so each
// private L2 sees compact regions while the uneven per-region cost balances.  Only the bricks the
// exact root early-out cannot clear are launched when that removes a good part of the grid.
// The per-thread traversal stack (shared with the postponed-leaf queue) is an LDS column
// (stack[entry][thread]: consecutive lanes hit consecutive banks); a ray that runs out of it is
// listed and finished by k_voxelize_redo with a deep column, never ignored (dxv_api.hip).
// Parity mode normally runs k_parity_rows (one wave-uniform walk per block of grid rows, below).
#include "dxv_device.h"
#include "dxv_trace.h"
#include "dxv_dirmap.h"
#include <algorithm>
#include <vector>

namespace dxv {

template <int BX, int BY, int BZ>
struct Brick {
    static constexpr int x = BX, y = BY, z = BZ, threads = BX * BY * BZ;
    static_assert(threads == 64 || threads == 128 || threads == 256, "one voxel per thread, whole wavefronts");
};

__device__ __forceinline__ uint32_t compact1by2(uint32_t x)
{
    x &= 0x09249249u;
    x = (x ^ (x >> 2)) & 0x030c30c3u;
    x = (x ^ (x >> 4)) & 0x0300f00fu;
    x = (x ^ (x >> 8)) & 0xff0000ffu;
    x = (x ^ (x >> 16)) & 0x000003ffu;
    return x;
}

// Brick at position `lin` of the launch order: Morton inside 2^m-brick super-blocks (m = p.mortonBits, the largest power
// of two dividing all three brick counts), super-blocks linear; offset by the launch's brick box.
__device__ __forceinline__ void brick_of_lin(const VoxelizeParams& p, uint32_t lin, uint32_t& bx, uint32_t& by, uint32_t& bz)
{
    const uint32_t m = p.mortonBits;
    const uint32_t low = lin & ((1u << (3u * m)) - 1u), high = lin >> (3u * m);
    bx = compact1by2(low); by = compact1by2(low >> 1); bz = compact1by2(low >> 2);
    if (p.superX == 1u && p.superY == 1u) bz |= high << m;      // usual case (cubic power-of-two grid): no divisions
    else {
        bx |= (high % p.superX) << m;
        by |= ((high / p.superX) % p.superY) << m;
        bz |= (high / (p.superX * p.superY)) << m;
    }
    bx += p.bx0; by += p.by0; bz += p.bz0;
}

// WALK: 0 = leaves tested as met, 1 = postponed-leaf walk, 2 = the same over the wide nodes (MODE 0)
template <class B, int STACK, int MODE, bool TEXELS, int WALK, int ABL = 0>
__global__ __launch_bounds__(B::threads, WALK == 4 ? 6 : 8) void k_voxelize(VoxelizeParams p)   // walks: <= 64 VGPRs, 8 waves/SIMD; lists (WALK 4): 70 VGPRs, 7 waves
{
    __shared__ int32_t stack[STACK * B::threads];
    const uint32_t N = p.N;
    uint32_t bx, by, bz;
    {
    const uint32_t nb = p.nbx * p.nby * p.nbz;
    // XCD-aware remap: workgroups b and b + 8 share an XCD.  Bricks are numbered along a Morton
    // curve (below); runs of 2^regionBits consecutive bricks (compact regions) are dealt round-robin
    // to the 8 XCDs: each XCD's L2 sees compact regions, and the regions of all XCDs are fine
    // grained enough to balance the very uneven per-region cost.
    const uint32_t rb = p.regionBits;
    const uint32_t j = blockIdx.x >> 3;
    const uint32_t lin = ((((j >> rb) << 3) | (blockIdx.x & 7u)) << rb) | (j & ((1u << rb) - 1u));
    if (lin >= nb) return;
    // Consecutive workgroups of an XCD cover a compact region and reuse the same part of the tree in L1/L2.
    brick_of_lin(p, lin, bx, by, bz);
    }
    const uint32_t tid = threadIdx.x;
    const uint32_t ix = bx * B::x + tid % B::x;
    const uint32_t iy = by * B::y + (tid / B::x) % B::y;
    const uint32_t lz = bz * B::z + tid / (B::x * B::y);
    if (ix >= N || iy >= N || lz >= p.nz) return;
    const uint32_t iz = p.zBlock == p.nz ? p.z0 + lz                                    // contiguous slab
                      : p.z0 + (lz >> p.zShift) * p.zPeriod + (lz & (p.zBlock - 1u));    // block-cyclic, zBlock = 2^zShift
    const size_t id = ((size_t)lz * N + iy) * N + ix;
    if (MODE == 0 && B::x == 4 && B::y == 4 && B::z == 4 && p.mipR) {
        // The work queue's brick test without a queue (tree walks, plan = 0): can ANY ray of this brick reach a triangle?  p.mip is the
        // max-mip of the far radii of the scene's lists, or -- a scene without lists -- of the triangles' own footprints (dirmap_far).
        // Wave-uniform; a dead brick costs its workgroup a hundred instructions and four loads instead of 64 walks out of the tree.
        float x0, x1, y0, y1, z0, z1;
        dm_brick_hull(N, p.nz, p.z0, p.zBlock, p.zShift, p.zPeriod, bx, by, bz, x0, x1, y0, y1, z0, z1);
        if (!dm_box_may_be_live(x0, x1, y0, y1, z0, z1, p.scene.rootLo, p.scene.rootHi, p.mip, p.mipR)) {
            if (TEXELS) p.texels[id] = 0u;
            p.grid[id] = 0;
            return;
        }
    }

    const StridedStack stk{stack + tid, B::threads};
    bool overflow = false;
    uint8_t occ;
    if (MODE == 0) {
        uint32_t texel = 0;
        occ = voxel_reference<WALK, StridedStack, ABL>(p.scene, N, ix, iy, iz, stk, STACK, TEXELS ? &texel : nullptr, overflow);
        if (TEXELS) p.texels[id] = texel;
    } else {
        occ = voxel_parity<WALK != 0>(p.scene, N, ix, iy, iz, stk, STACK, overflow);
    }
    if (overflow) {
        // this ray needs a deeper column than the launch has: hand the voxel to k_voxelize_redo
        const uint32_t slot = atomicAdd(p.status + 1 + p.redoParity, 1u);
        if (slot < p.redoCap) p.redo[slot] = (uint64_t)id;
        else atomicOr(p.status, 1u);
    }
    p.grid[id] = occ;
}

// ---------------------------------------------------------------------------------------------
// Work queue of the lists kernel (4 x 4 x 4 bricks): WHICH bricks a launch runs, decided on the device inside the stream.
//  * which: a ray that starts beyond the last entry of its texel (or whose texel is empty, or whose origin has left the
//    root box) is a miss after one load -- on torus-1M four waves in ten of a launch over the brick box held no other
//    ray.  k_plan_bricks decides per BRICK, conservatively (dm_box_may_be_live, dxv_dirmap.h: the brick's footprint in
//    direction space and its smallest start radius against a max-mip of the texels' far radii; a false positive costs a
//    wave that finds nothing, a false negative cannot happen -- k_plan_check below is the exhaustive proof obligation);
//  * layout: regions of 256 consecutive bricks of the Morton order (8 x 8 x 4 bricks) are dealt round-robin to eight
//    queues, one per XCD (blocks b and b + 8 share one), so that an XCD's private L2 sees compact regions; a region's
//    workgroup appends its live bricks to its queue with one atomic add (small partitions: runs of 128 bricks, one add per
//    wave -- k_plan_bricks).  Queue memory (dxv_device.h): two headers -- eight heads per queue and the eight lengths, every
//    word in a 256-byte line of its own; a build takes the one the last build left cleared -- and 8 x cap brick words
//    (bx | by << 10 | bz << 20);
//  * how: k_voxelize_queue is launched with as many single-wave workgroups as the GPU holds at once.  Every wave takes its
//    bricks one at a time from a head of its XCD's queue with a returning atomic add, asked for one brick ahead.  Which
//    XCD a block really runs on is a matter of speed only: every head of every queue has its home waves by block number.
//    No host round trip: the launch's size does not depend on how many bricks are live.
//  * order: as built -- Morton order, regions dealt round-robin.  Measured and dropped (profiles/r04/ab_queue_*): dealing finer or to
//    the shortest queue; a second queue per XCD, run last, for the bricks near or across the outer end of their lists (three
//    definitions); and, for queues that are launched again, orders made on the device from MEASURED times -- the cheapest chunks of
//    64 slots last (-3 % of a rank's share, +1 % on a whole grid), all chunks by cost (-6 % / +4 %), the bricks that took over three
//    times the mean first and the shortest last (nothing): none earns a second copy of the queue.
// Bricks that are not queued are zero because k_plan_bricks clears the partition's grid while it builds the queue.
// ---------------------------------------------------------------------------------------------
[[maybe_unused]] constexpr uint32_t kQueueNoPrefetch = 1024u;
constexpr uint32_t kPlanRegionBits = 8u;                               // regions of 256 consecutive bricks = one workgroup of k_plan_bricks
// (header layout: queue_len_word / queue_head_word in dxv_device.h -- every queue's two words in a 256-byte line of its own:
// returning atomics on ONE line serialise at ~90 per us for all eight queues together, 2.7 ms of a launch when first tried)

// The launch's zeros travel with the queue build: workgroup b clears the b-th share of the grid (and of the texel image) with
// 16-byte stores while its threads wait for their four mip words -- one kernel in front of the brick kernel instead of a memset
// of the grid, a memset of the header and this one (three dependent dispatches: ~5 us each on top of their own time).
// Block 0 clears the frame's other header for the launch that builds the next queue.
__device__ __forceinline__ void plan_clear(uint8_t* base, size_t bytes, uint32_t nblocks)
{
    const size_t chunk = (((bytes + nblocks - 1u) / nblocks) + 15u) & ~(size_t)15u;
    const size_t lo = (size_t)blockIdx.x * chunk;
    if (lo >= bytes) return;
    const size_t hi = lo + chunk < bytes ? lo + chunk : bytes, full = lo + ((hi - lo) & ~(size_t)15u);
    // (non-temporal stores: 134 MB of zeros that nobody reads before the brick kernel has overwritten a fifth of them should not push
    // the lists out of the L2s and the memory-side cache on their way -- plain stores: the queue build 0.0375 instead of 0.0328 ms and
    // the brick kernel behind it 0.681 instead of 0.666, profiles/r05/ab_nontemporal_grid_stores.jsonl)
    typedef uint32_t Zero4 __attribute__((ext_vector_type(4)));
    const Zero4 z = {0u, 0u, 0u, 0u};
    for (size_t o = lo + 16u * threadIdx.x; o < full; o += 16u * 256u) __builtin_nontemporal_store(z, reinterpret_cast<Zero4*>(base + o));
    if (full + threadIdx.x < hi) base[full + threadIdx.x] = 0;           // (a grid whose bytes are no multiple of 16: the last block's tail)
}

__global__ __launch_bounds__(256) void k_plan_bricks(VoxelizeParams p, uint32_t nb)
{
    __shared__ uint32_t heavyCount[4], lightCount[4], heavyBase[4], lightBase[4];
    const uint32_t lin = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    bool live = false;
    uint32_t bx = 0, by = 0, bz = 0;
    if (lin < nb) {
        brick_of_lin(p, lin, bx, by, bz);
        float x0, x1, y0, y1, z0, z1;
        dm_brick_hull(p.N, p.nz, p.z0, p.zBlock, p.zShift, p.zPeriod, bx, by, bz, x0, x1, y0, y1, z0, z1);
        live = dm_box_may_be_live(x0, x1, y0, y1, z0, z1, p.scene.rootLo, p.scene.rootHi, p.mip, p.scene.dmR);
    }
    if (p.planClear) {
        plan_clear(p.grid, (size_t)p.N * p.N * p.nz, gridDim.x);
        if (p.texels) plan_clear(reinterpret_cast<uint8_t*>(p.texels), (size_t)p.N * p.N * p.nz * 4u, gridDim.x);
    }
    if (p.queueZero && blockIdx.x == 0u)
        for (uint32_t k = threadIdx.x; k < kQueueHeaderWords; k += 256u) p.queueZero[k] = 0u;
    // heavy: one of the brick's rays can look into a list that is long for this scene (one and a half times the mean of the count
    // mip at the level of a brick's patch of texels: k_dm_heavy_thresholds) -- 2 - 6 % of the queued bricks, among them 99 % of those
    // that take three times the mean and more (profiles/r05/brick_features.jsonl)
    bool heavy = false;
    if (live) {
        float x0, x1, y0, y1, z0, z1;
        dm_brick_hull(p.N, p.nz, p.z0, p.zBlock, p.zShift, p.zPeriod, bx, by, bz, x0, x1, y0, y1, z0, z1);
        const uint16_t* countMip = p.mip + dm_mip_words(p.scene.dmR);
        // (maps too small to have such a level -- R < 8 -- have no word: no brick is heavy there)
        const uint32_t longList = p.planHeavy ? p.planHeavy : dm_mip_levels(p.scene.dmR) > kDmHeavyLevelMin ? countMip[dm_mip_words(p.scene.dmR) + dm_heavy_level(p.scene.dmR, p.N)] : 0xffffu;
        heavy = dm_box_max_count(x0, x1, y0, y1, z0, z1, countMip, p.scene.dmR) > longList;
    }
    const unsigned long long mh = __ballot(live && heavy), ml = __ballot(live && !heavy);
    if (lane == 0u) { heavyCount[w] = (uint32_t)__builtin_popcountll(mh); lightCount[w] = (uint32_t)__builtin_popcountll(ml); }
    __syncthreads();
    // Runs of 2^planRegionBits consecutive Morton bricks go to one queue, the runs dealt round-robin: 256 (8 x 8 x 4 bricks, the whole
    // workgroup: an XCD's L2 sees compact pieces of the grid), 128 or 64 (one wave each).
    const uint32_t wavesPerRun = 1u << (p.planRegionBits - 6u), first = w & ~(wavesPerRun - 1u);
    const uint32_t x = (lin >> p.planRegionBits) & 7u;
    if (lane == 0u && w == first) {
        uint32_t nh = 0, nl = 0;
        for (uint32_t k = 0; k < wavesPerRun; ++k) { nh += heavyCount[first + k]; nl += lightCount[first + k]; }
        heavyBase[first] = nh ? atomicAdd(p.queue + queue_heavy_word(x), nh) : 0u;
        lightBase[first] = nl ? atomicAdd(p.queue + queue_len_word(x), nl) : 0u;
    }
    __syncthreads();
    if (!live) return;
    const unsigned long long before = (1ull << lane) - 1ull;
    uint32_t rank = (uint32_t)__builtin_popcountll((heavy ? mh : ml) & before);
    for (uint32_t k = first; k < w; ++k) rank += heavy ? heavyCount[k] : lightCount[k];
    // (heavy bricks from slot 0 upwards, the others from the far end downwards: queue_slot)
    const uint32_t slot = heavy ? heavyBase[first] + rank : p.queueCap - 1u - (lightBase[first] + rank);
    p.queueSlots[(size_t)x * p.queueCap + slot] = bx | (by << 10) | (bz << 20);
    if (p.liveMask) {                                                   // (a queue that is being prepared: the bit the launches' clear reads)
        const uint32_t nbx = (p.N + 3u) / 4u, id = (bz * nbx + by) * nbx + bx;
        atomicOr(p.liveMask + (id >> 5), 1u << (id & 31u));
    }
}
size_t plan_live_words(uint32_t N, uint32_t nz)
{
    const uint64_t nbx = (N + 3u) / 4u, nbz = (nz + 3u) / 4u;
    return (size_t)((nbx * nbx * nbz + 31u) / 32u) + 4u;
}

// the brick order of the whole partition (no brick box): what k_plan_bricks, the checker and the host agree on
uint32_t plan_layout(VoxelizeParams& p)
{
    const uint32_t nbx = (p.N + 3u) / 4u, nby = nbx, nbz = (p.nz + 3u) / 4u;
    p.nbx = nbx; p.nby = nby; p.nbz = nbz;
    p.bx0 = p.by0 = p.bz0 = 0;
    uint32_t m = 0;
    while (m < 10 && !((nbx >> m) & 1u) && !((nby >> m) & 1u) && !((nbz >> m) & 1u)) ++m;
    p.mortonBits = m;
    p.superX = nbx >> m;
    p.superY = nby >> m;
    return nbx * nby * nbz;
}
// Run length by partition size.  Large partitions: 256 bricks (an XCD's L2 sees compact pieces of the grid, and with thousands of
// runs per queue the eight queues end within 2 % of each other).  Small ones -- a 256^3 grid, a rank's share of 512^3 at 4 ranks
// or more: 2^19 bricks or fewer -- take shorter runs: a queue of a few hundred runs of very different cost ends 10 - 20 % away
// from its neighbours, and the launch ends with the longest.  (Runs of 64 until round 6; since every XCD runs an equal share of all
// eight queues -- queue_item -- their imbalance matters less than an XCD's locality: 128 is -3 % at 256^3 and -2 ... -3 % on a
// rank's share of the 1 M-triangle meshes at 512^3, +1.5 % on dragon x9's: profiles/r06/ab_planregion_at_eight_waves.jsonl.)
uint32_t plan_region_bits(uint32_t N, uint32_t nz)
{
    const uint64_t nb = (uint64_t)((N + 3u) / 4u) * ((N + 3u) / 4u) * ((nz + 3u) / 4u);
    return nb <= (1ull << 19) ? 7u : kPlanRegionBits;
}
// words of queue memory a partition needs (two headers + eight queues, each able to hold every run dealt to it in full, whatever
// the run length)
size_t plan_queue_words(uint32_t N, uint32_t nz, uint32_t* capOut)
{
    const uint64_t nb = (uint64_t)((N + 3u) / 4u) * ((N + 3u) / 4u) * ((nz + 3u) / 4u);
    uint64_t cap = 0;
    for (uint32_t rb = 6u; rb <= kPlanRegionBits; ++rb) {
        const uint64_t runs = (nb + (1u << rb) - 1u) >> rb, c = ((runs + 7u) / 8u) << rb;
        if (c > cap) cap = c;
    }
    if (capOut) *capOut = (uint32_t)cap;
    return kQueueSlotsAt + 8u * (size_t)cap;
}

// one workgroup per 256 bricks into the header p.queue, which the caller vouches is all zero; p.queueSlots / p.queueCap / p.mip set by the caller
hipError_t plan_build(const VoxelizeParams& pin, hipStream_t s)
{
    VoxelizeParams p = pin;
    const uint32_t nb = plan_layout(p), nr = (nb + (1u << kPlanRegionBits) - 1u) >> kPlanRegionBits;
    if (p.planRegionBits < 6u || p.planRegionBits > kPlanRegionBits) p.planRegionBits = kPlanRegionBits;
    k_plan_bricks<<<dim3(nr), dim3(256), 0, s>>>(p, nb);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Test hook (dxv_debug_plan_check): the queue's claim -- no live ray sits in a brick that is not queued -- checked
// exhaustively.  k_plan_mark sets one bit per queued brick (and counts bricks queued twice); k_plan_check makes, for every
// voxel of the partition, exactly the decision the kernel's first step makes (origin_leaves_root, dm_ray_start: the same
// functions) and requires the brick of every live voxel to be marked.
// out[0] live voxels, out[1] bricks with a live voxel, out[2] queued bricks, out[3] violations (must be 0), out[4] bricks
// queued more than once (must be 0), out[5 + k]: brick word of the first 11 violations.  Not a product path.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_plan_mark(VoxelizeParams p, uint32_t* __restrict__ bits, unsigned long long* __restrict__ out)
{
    const uint32_t nbx = (p.N + 3u) / 4u;
    for (uint32_t x = 0; x < 8u; ++x) {
        const uint32_t heavy = p.queue[queue_heavy_word(x)], len = heavy + p.queue[queue_len_word(x)];
        for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < len; k += gridDim.x * 256u) {
            const uint32_t w = p.queueSlots[(size_t)x * p.queueCap + queue_slot(k, heavy, p.queueCap)];
            const uint32_t id = ((w >> 20) * nbx + ((w >> 10) & 1023u)) * nbx + (w & 1023u);
            const uint32_t old = atomicOr(bits + (id >> 5), 1u << (id & 31u));
            if (old & (1u << (id & 31u))) atomicAdd(out + 4, 1ull);
            atomicAdd(out + 2, 1ull);
        }
    }
}
__global__ __launch_bounds__(64) void k_plan_check(VoxelizeParams p, uint32_t nb, const uint32_t* __restrict__ bits, unsigned long long* __restrict__ out)
{
    const uint32_t lin = blockIdx.x;
    if (lin >= nb) return;
    uint32_t bx, by, bz;
    brick_of_lin(p, lin, bx, by, bz);
    const uint32_t tid = threadIdx.x, N = p.N;
    const uint32_t ix = bx * 4u + (tid & 3u), iy = by * 4u + ((tid >> 2) & 3u), lz = bz * 4u + (tid >> 4);
    bool live = false;
    if (ix < N && iy < N && lz < p.nz) {
        const uint32_t iz = p.zBlock == p.nz ? p.z0 + lz : p.z0 + (lz >> p.zShift) * p.zPeriod + (lz & (p.zBlock - 1u));
        float ox, oy, oz;
        ray_origin(N, ix, iy, iz, ox, oy, oz);
        if (!origin_leaves_root(ox, oy, oz, p.scene.rootLo, p.scene.rootHi)) {
            const DirMapView dm{static_cast<const DirCell*>(p.scene.dmCells), static_cast<const DirEntry*>(p.scene.dmEntries), p.scene.dmR};
            live = dm_ray_start(ox, oy, oz, dm).live;
        }
    }
    const unsigned long long m = __ballot(live);
    if (tid != 0u || !m) return;
    atomicAdd(out, (unsigned long long)__builtin_popcountll(m));
    atomicAdd(out + 1, 1ull);
    const uint32_t nbx = (N + 3u) / 4u, id = (bz * nbx + by) * nbx + bx;
    if (!(bits[id >> 5] & (1u << (id & 31u)))) {
        const unsigned long long slot = atomicAdd(out + 3, 1ull);
        if (slot < 11ull) out[5 + slot] = bx | (by << 10) | (bz << 20);
    }
}
hipError_t launch_plan_check(const VoxelizeParams& pin, uint32_t* bits, unsigned long long* out, hipStream_t s)
{
    VoxelizeParams p = pin;
    const uint32_t nb = plan_layout(p);
    hipError_t e = hipMemsetAsync(bits, 0, sizeof(uint32_t) * (((size_t)nb + 31u) / 32u), s);
    if (e == hipSuccess) e = hipMemsetAsync(out, 0, 16 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    k_plan_mark<<<dim3(256), dim3(256), 0, s>>>(p, bits, out);
    k_plan_check<<<dim3(nb), dim3(64), 0, s>>>(p, nb, bits, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Eight queues of unequal length, eight XCDs of equal appetite.  Runs of bricks are dealt to the queues by their number, not by
// what they hold: on a rank's share of the grid the queues differ by up to 30 % in length (bunny x16 at 8 ranks: 6,400 against
// 9,100 bricks), and a launch ends with its longest queue while half of the GPU idles (profiles/r05/wg_times_before.jsonl).  So
// the launch is dealt out in EQUAL shares: XCD x runs T = ceil(total / 8) items -- its own queue's first min(len_x, T), and, when
// its queue is shorter than T, items from the far end of the queues that are longer (what they hold beyond their own first T), in
// queue order.  A pure function of the eight lengths, which every workgroup reads from the header: no second pass over the
// queues, nothing moved; 85 - 100 % of an XCD's bricks are still its own compact runs.
// ---------------------------------------------------------------------------------------------
struct QueueLens { uint32_t len[8], heavy[8]; };        // items per queue, of which heavy
#if defined(__HIP_DEVICE_COMPILE__)
// (the arithmetic runs on lanes 0 .. 7 of the wave -- one length each -- where a brick body that has not begun yet leaves every
// vector register free; held in scalar registers the eight lengths cost the persistent kernel a wave per SIMD)
__device__ __forceinline__ uint32_t dpp_row_shr(uint32_t v, int n)      // lane i <- lane i - n of its row of 16, 0 where there is none
{
    return n == 1 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true)
         : n == 2 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true)
                  : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t prefix8(uint32_t v)                 // inclusive prefix sums over lanes 0 .. 7 (v = 0 on the lanes behind them)
{
    v += dpp_row_shr(v, 1); v += dpp_row_shr(v, 2); v += dpp_row_shr(v, 4);
    return v;
}
__device__ __forceinline__ uint32_t queue_lens(const uint32_t* hdr, uint32_t& T, uint32_t& H)   // lane a < 8: items of queue a, of which H heavy; T = ceil(total / 8)
{
    uint32_t lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    H = lane < 8u ? hdr[queue_heavy_word(0) + 64u * lane] : 0u;
    const uint32_t L = lane < 8u ? hdr[queue_len_word(0) + 64u * lane] + H : 0u;
    T = ((uint32_t)__builtin_amdgcn_readlane((int)prefix8(L), 7) + 7u) >> 3;
    return L;
}
// item j (< T) of XCD x: queue and slot; false: none (the last few of the 8 T items when the total is no multiple of 8)
__device__ __forceinline__ bool queue_item(const uint32_t* hdr, uint32_t cap, uint32_t x, uint32_t j, uint32_t& y, uint32_t& slot)
{
    uint32_t T, H, lane, k;
    const uint32_t L = queue_lens(hdr, T, H), lenX = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)x);
    if (j < lenX) { y = x; k = j; }                             // (j < T: one of the queue's own first T)
    else {
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        const uint32_t spare = lane < 8u && L < T ? T - L : 0u, extra = L > T ? L - T : 0u;
        const uint32_t spareBefore = prefix8(spare) - spare, extraBefore = prefix8(extra) - extra;
        // the (j - len_x)-th slot this XCD has to spare, counted behind the spare slots of the XCDs 0 .. x - 1, is given the g-th
        // brick that some queue holds beyond its own first T
        const uint32_t g = j - lenX + (uint32_t)__builtin_amdgcn_readlane((int)spareBefore, (int)x);
        const uint64_t m = __builtin_amdgcn_ballot_w64(lane < 8u && g >= extraBefore && g - extraBefore < extra);
        if (!m) return false;
        y = (uint32_t)__builtin_ctzll(m);
        k = T + g - (uint32_t)__builtin_amdgcn_readlane((int)extraBefore, (int)y);
    }
    slot = queue_slot(k, (uint32_t)__builtin_amdgcn_readlane((int)H, (int)y), cap);
    return true;
}
// the same from eight lengths the host holds (a kept queue, k_voxelize_listed: kernel arguments, no load in front of the brick's own)
__device__ __forceinline__ bool queue_item(const QueueLens& q, uint32_t cap, uint32_t x, uint32_t j, uint32_t& y, uint32_t& slot)
{
    uint32_t total = 0, lenX = 0, k = 0;
#pragma unroll
    for (int a = 0; a < 8; ++a) { total += q.len[a]; lenX = x == (uint32_t)a ? q.len[a] : lenX; }
    const uint32_t T = (total + 7u) >> 3;
    bool found = j < lenX;
    y = x; k = j;
    if (!found) {
        uint32_t g = j - lenX;
#pragma unroll
        for (int a = 0; a < 8; ++a) g += ((uint32_t)a < x && q.len[a] < T) ? T - q.len[a] : 0u;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const uint32_t extra = q.len[a] > T ? q.len[a] - T : 0u;
            if (!found && g < extra) { y = (uint32_t)a; k = T + g; found = true; }
            g -= found ? 0u : extra;
        }
    }
    uint32_t heavy = 0;
#pragma unroll
    for (int a = 0; a < 8; ++a) heavy = y == (uint32_t)a ? q.heavy[a] : heavy;
    slot = queue_slot(k, heavy, cap);
    return found;
}
#endif

// ---------------------------------------------------------------------------------------------
// The lists kernel over the work queue: persistent single-wave workgroups (see above).  One brick = one pass of the body of
// k_voxelize<Brick<4,4,4>, 16, 0, TEXELS, 4>; the 64 result bytes of a brick leave as 16 dwords (one per 4-voxel row,
// assembled from the wave's ballot) instead of 64 bytes.
// ---------------------------------------------------------------------------------------------
template <bool TEXELS>
__global__ __launch_bounds__(64, 6) void k_voxelize_queue(VoxelizeParams p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ int32_t stack[16 * 64];
    // A queue is handed out through eight heads: head h counts the slots k = h (mod 8), so that the eight groups of an XCD's waves
    // (a wave's home head: its number among the XCD's waves mod 8) advance through the queue together, one brick per add -- the
    // bricks in flight on an XCD stay a compact window of its queue (what hardware dispatch of one workgroup per brick gave:
    // neighbouring bricks look into the same texels while they are in the caches; chunks of 8 consecutive bricks per wave
    // cost 7 %, of 16 15 %), and no head sees more than a few adds per microsecond (all bricks through ONE word: 2.7 ms).
#if defined(DXV_QUEUE_TIMES)
    const uint64_t tStart = __builtin_amdgcn_s_memrealtime();
    uint64_t tBrick = tStart, tLast = tStart, tMax = 0, nBricks = 0;
#endif
    // Every head has HOME waves that drain it to its last slot: wave w of XCD x (x = block % 8, w = block / 8) is home to the heads
    // h = w (mod 8) of queue x -- to h = w (mod W) when fewer than eight waves per XCD were launched, so that no head is without
    // one (which XCD a block really runs on is a matter of speed only).
    const uint32_t nh = p.queueHeads;                                   // heads per queue in use: 1, 2, 4 or 8 (head h hands out the items = h mod nh)
    const uint32_t x0 = blockIdx.x & 7u, wx = blockIdx.x >> 3;
    uint32_t perXcd = gridDim.x >> 3;
    // A short launch does not want every wave the GPU holds: with fewer than p.queueMinBricks bricks per wave the bricks of a wave
    // contend with seven times as many neighbours as they need to fill the launch's few rounds, and each wave holds one brick in
    // reserve at the end (256^3, 60 k bricks: 0.192 ms with 7,168 waves, 0.166 with 5,120: profiles/r05/ab_persistent_waves.jsonl).
    // The launch cannot know its size on the host; its waves can: the surplus ones leave before they touch the queue.
    if (p.queueMinBricks) {
        uint32_t share, heavy;
        (void)queue_lens(p.queue, share, heavy);
        uint32_t want = ((share + p.queueMinBricks - 1u) / p.queueMinBricks + 7u) & ~7u;      // (a multiple of 8: every head keeps its home waves)
        want = want < 64u ? 64u : want;
        if (want < perXcd) perXcd = want;
        if (wx >= perXcd) return;
    }
    const uint32_t homes = perXcd < nh ? perXcd : nh;
    uint64_t homeMask = 0;
    for (uint32_t h = wx % homes; h < nh; h += homes) homeMask |= 1ull << (8u * x0 + h);
    uint32_t cur = 8u * x0 + wx % homes;
    uint64_t tried = 0;
    for (;;) {
        tried |= 1ull << cur;
        const uint32_t x = cur >> 3, h = cur & 7u;
        uint32_t len, own, ownHeavy;                                    // items of XCD x: its equal share of the launch (queue_item)
        {
            uint32_t H;
            const uint32_t L = queue_lens(p.queue, len, H);
            own = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)x);  // ... of which its own queue's, and how many of those are heavy
            ownHeavy = (uint32_t)__builtin_amdgcn_readlane((int)H, (int)x);
        }
        uint32_t* head = p.queue + queue_head_word(x, h);
        if (len > h) {
        // One brick ahead: the add for the next brick is issued in front of the current one, and its answer is taken out of its
        // vector register as soon as the brick's first load (the rays' cells: all 64 lanes make that step together) has arrived --
        // by then it is there (memory operations return in order) -- so nothing of the queue lives in a vector register through
        // the scan and the triangle tests.
        // (... except near a queue's end: a wave that holds a second brick there keeps it from the waves that have run out of work)
        uint32_t jv = 0;
        if (threadIdx.x == 0u) jv = atomicAdd(head, 1u);
        uint32_t next = (uint32_t)__builtin_amdgcn_readlane((int)jv, 0);
        uint32_t wAhead = 0xffffffffu;                                  // the next brick's word when it was fetched during the current brick (no brick word has its top bits set)
        bool asked = false;                                             // an add is in flight (asked for behind the last brick's scan)
        for (;;) {
            const uint32_t k = nh * next + h;
            if (k >= len) break;
            const bool ahead = len - k > kQueueNoPrefetch;
            if (ahead && !asked && threadIdx.x == 0u) jv = atomicAdd(head, 1u);   // (a head's first brick; later ones: behind the scan of the brick before)
            // The launch's parameters are read from the kernel-argument segment again for every brick (scalar loads that
            // hit the scalar cache): kept across the loop they would hold fifty SGPRs through the whole brick body, and the
            // body (the one of k_voxelize: 70 VGPRs, 56 SGPRs) would lose a wave per SIMD to registers.
            typedef const __attribute__((address_space(4))) VoxelizeParams* KernArg;
            KernArg pp = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(pp));
            uint32_t w = wAhead;                                        // through the scalar cache: one word per wave
            if (w == 0xffffffffu) {
                const uint32_t* hdr = pp->queue;
                const uint32_t cap = pp->queueCap;
                uint32_t qy = x, qslot = queue_slot(k, ownHeavy, cap);
                bool any = true;
                if (k >= own) any = queue_item(hdr, cap, x, k, qy, qslot);      // (beyond the XCD's own queue: a longer queue's far end)
                if (!any) break;                                        // (only the very last items of the launch)
                const uint32_t* slot = pp->queueSlots + (uint32_t)__builtin_amdgcn_readfirstlane((int)(qy * cap + qslot));   // (8 cap <= 2^27 bricks)
                asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(slot) : "memory");
            }
            SceneView sc;                                               // (what the lists' path reads of it)
            sc.nodes = nullptr; sc.wide = nullptr; sc.plCells = nullptr; sc.plEntries = nullptr; sc.plR = 0;
            sc.triPos = pp->scene.triPos; sc.triNrm = pp->scene.triNrm;
            sc.dmCells = pp->scene.dmCells; sc.dmEntries = pp->scene.dmEntries; sc.dmR = pp->scene.dmR; sc.dmCoop = pp->scene.dmCoop;
#pragma unroll
            for (int a = 0; a < 3; ++a) { sc.rootLo[a] = pp->scene.rootLo[a]; sc.rootHi[a] = pp->scene.rootHi[a]; }
            const uint32_t N = pp->N, nz = pp->nz;
            const uint32_t bx = w & 1023u, by = (w >> 10) & 1023u, bz = w >> 20;
            // (the lane number anew for every brick, and once more behind the body: nothing of the loop lives in vector registers
            // through the body)
            uint32_t tid;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tid));
            uint32_t ix = bx * 4u + (tid & 3u), iy = by * 4u + ((tid >> 2) & 3u), lz = bz * 4u + (tid >> 4);
            ix = ix < N ? ix : N - 1u; iy = iy < N ? iy : N - 1u; lz = lz < nz ? lz : nz - 1u;   // (lanes that hang over the grid's end repeat its last voxels and store nothing)
            const uint32_t zBlock = pp->zBlock;
            const uint32_t iz = zBlock == nz ? pp->z0 + lz : pp->z0 + (lz >> pp->zShift) * pp->zPeriod + (lz & (zBlock - 1u));
            // raygenMain for the 64 voxels of the brick (voxel_reference<4>, dxv_trace.h, with its first step made by all lanes)
            Ray r;
            ray_origin(N, ix, iy, iz, r.ox, r.oy, r.oz);
            const DirMapView dm{static_cast<const DirCell*>(sc.dmCells), static_cast<const DirEntry*>(sc.dmEntries), sc.dmR, sc.dmCoop};
            DirRayStart start = dm_ray_start(r.ox, r.oy, r.oz, dm);
            wAhead = 0xffffffffu;
            if (ahead) {
                next = (uint32_t)__builtin_amdgcn_readlane((int)jv, 0);  // the next brick's number
                // ... and its word, asked for now: fetched at the top of the loop it is a scalar load that nothing hides -- 0.7 us of an
                // 11 us brick, the difference between these waves and a workgroup per brick dealt out by the hardware.  (One scalar
                // register through the body; items beyond the XCD's own queue -- queue_item -- are looked up when their turn comes.)
                const uint32_t kn = nh * next + h;
                if (kn < own) {
                    typedef const __attribute__((address_space(4))) uint32_t* ConstWords;
                    wAhead = ((ConstWords)pp->queueSlots)[x * pp->queueCap + queue_slot(kn, ownHeavy, pp->queueCap)];
                }
            }
            if (origin_leaves_root(r.ox, r.oy, r.oz, sc.rootLo, sc.rootHi)) start.live = false;   // provably missMain
            Hit best;
            float bestDet = 1.0f;
            const StridedStack stk{stack + tid, 64};
            trace_reference_dm_from<StridedStack, 0, TEXELS ? 0 : 2>(r, dm, start, sc.triPos, stk, 16, best, bestDet);
            // The add for the brick AFTER the next one, here: vector memory answers in order, so an add asked for right in front of a
            // brick's first load makes that load wait for the add's 1.1 - 1.3 us instead of its own 0.8 -- asked for behind the scan,
            // it has the predicate, the stores and the next brick's ray set-up (nine divisions) to come back in.
            asked = false;
            if (ahead) {
                const uint32_t kn = nh * next + h;
                if (kn < len && len - kn > kQueueNoPrefetch) {
                    if (threadIdx.x == 0u) jv = atomicAdd(head, 1u);
                    asked = true;
                }
            }
            uint32_t texel = 0;
            const uint8_t occ = TEXELS ? shade_reference<4, 0>(sc, r, best, bestDet, &texel) : shade_reference_again(sc, r, best.leaf);
            // the lane's voxel once more (nothing of it was kept through the body)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tid));
            uint8_t* grid = pp->grid;
            if (TEXELS || (N & 3u) != 0u) {
                const uint32_t vx = bx * 4u + (tid & 3u), vy = by * 4u + ((tid >> 2) & 3u), vz = bz * 4u + (tid >> 4);
                if (vx < N && vy < N && vz < nz) {
                    const size_t id = ((size_t)vz * N + vy) * N + vx;
                    if (TEXELS) pp->texels[id] = texel;
                    if ((N & 3u) != 0u) grid[id] = occ;
                }
            }
            if ((N & 3u) == 0u) {
                // rows of 4 voxels are aligned dwords: lane r < 16 stores row (y = r & 3, z = r >> 2) from the wave's ballot
                const uint64_t m = __builtin_amdgcn_ballot_w64(occ != 0);
                const uint32_t ry = by * 4u + (tid & 3u), rz = bz * 4u + ((tid >> 2) & 3u);
                if (tid < 16u && rz < nz) {
                    const uint32_t nib = (uint32_t)(m >> (4u * tid)) & 15u;
                    // (a plain store, like the hardware-dispatched kernel's below: four bytes that leave non-temporally reach the fabric as
                    // partial writes -- 253 MB written per launch for 29 MB of results, for 0.5 % that was inside the boxes' spread)
                    *reinterpret_cast<uint32_t*>(grid + ((size_t)rz * N + ry) * N + bx * 4u) = (nib * 0x00204081u) & 0x01010101u;   // bit i -> byte i
                }
            }
#if defined(DXV_QUEUE_TIMES)
            { const uint64_t now = __builtin_amdgcn_s_memrealtime(); tLast = tBrick; if (now - tBrick > tMax) tMax = now - tBrick; tBrick = now; ++nBricks; }
#endif
            if (!ahead) {
                if (threadIdx.x == 0u) jv = atomicAdd(head, 1u);
                next = (uint32_t)__builtin_amdgcn_readlane((int)jv, 0);
            }
        }
        }
        // this head is done: the wave's other home heads (fewer than eight waves per XCD), else the wave is done.  A wave does NOT go
        // looking for work on other heads or other XCDs' queues: at the end of a launch thousands of waves doing so at once are
        // thousands of adds and loads on single words (~90 per microsecond each) -- measured, in four variants: every wave spent
        // 30 - 60 us there and a rank's share of the grid took 0.195 instead of 0.147 ms (profiles/r04/queue_wave_times.jsonl,
        // ab_queue_helping.jsonl).  Round 5 tried the cheapest form once more -- one load of the wave's own queue's eight heads, then a
        // move to the head with most items left, never to another XCD's queue: a rank's share 0.142 -> 0.170 ms (the waves of a
        // drained head all pick the same head: profiles/r05/ab_steal_within_the_queue_rejected.jsonl, .patch).
        const uint64_t homeLeft = homeMask & ~tried;
        if (!homeLeft) break;
        cur = (uint32_t)__builtin_ctzll(homeLeft);
    }
#if defined(DXV_QUEUE_TIMES)
    // (diagnostic build only, tools/queue_times.py: start and end of every wave in 100 MHz ticks, in the frame's unused redo list)
    if (threadIdx.x == 0u && 4u * blockIdx.x + 3u < p.redoCap) {
        p.redo[4u * blockIdx.x] = tStart; p.redo[4u * blockIdx.x + 1u] = __builtin_amdgcn_s_memrealtime();
        p.redo[4u * blockIdx.x + 2u] = (nBricks << 32) | tMax; p.redo[4u * blockIdx.x + 3u] = tLast;      // bricks, longest brick, start of the last one
    }
#endif
#else
    (void)p;
#endif
}

// The same brick body with one workgroup per queued brick, dispatched by the hardware: for a queue that is launched AGAIN and whose
// eight lengths the host has read meanwhile (dxv_sync of an earlier launch of the same queue) -- the launch's size is then
// known without a round trip of its own.  Workgroup b takes item b / 8 of XCD b % 8's equal share (workgroups b and b + 8 share an XCD);
// no heads, no adds, parameters in scalar registers from the start.  What it is for: short launches (a 256^3 grid, a rank's
// share), whose few bricks per persistent wave leave the end of the launch ragged (option dispatch).
// The clear of a launch through a PREPARED queue (launch_voxelize_prepared), inside the brick kernel's own dispatch: the bricks that are
// queued write all 64 of their voxels themselves, so a launch only has to zero the bricks that are NOT queued -- and that has no order
// to keep with the brick workgroups (disjoint bytes), which is what lets both share one dispatch.  One thread per 16 voxels of a grid
// row (16 bytes = four bricks' rows) and step; the four bricks' bits sit in one nibble of the prepared queue's brick mask (ids run
// along x, N % 16 == 0).  Non-temporal stores, like every clear of this file: zeros nobody reads soon should not push the lists out
// of the caches.
struct ClearShare { const uint32_t* live; uint32_t blocks; uint32_t where; };   // blocks: workgroups that clear (0: none, a multiple of 8); where: 1 = the launch's first,
                                                                                // 2 = its last, 3 = spread evenly between the bricks' (rows of 8 workgroups, one per XCD)
__device__ __forceinline__ void clear_dead_bricks(const VoxelizeParams& p, const uint32_t* __restrict__ live, uint32_t block, uint32_t nblocks)
{
    typedef uint32_t Zero4 __attribute__((ext_vector_type(4)));
    const Zero4 z = {0u, 0u, 0u, 0u};
    const uint32_t N = p.N, px = N >> 4, nbx = N >> 2;
    const uint32_t pieces = px * N * p.nz, per = (pieces + nblocks - 1u) / nblocks;      // (<= 2^29 pieces: 32-bit arithmetic throughout)
    const uint32_t lo = block * per, hi = lo + per < pieces ? lo + per : pieces;
    // four pieces per thread and round: their mask words are asked for together (a chain of sixteen dependent loads per thread made
    // a clearing workgroup last 16 us -- longer than a brick)
    for (uint32_t base = lo; base < hi; base += 256u) {                 // (wave-uniform: the texel image's stores read other lanes' nibbles)
        const uint32_t q0 = base + threadIdx.x;
        uint32_t nib[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t q = q0 + 64u * u;
            nib[u] = 15u;                                              // (beyond the share: nothing to do)
            if (q < hi) {
                const uint32_t row = q / px, x16 = q - row * px, lz = row / N, y = row - lz * N;
                const uint32_t bit = ((lz >> 2) * nbx + (y >> 2)) * nbx + (x16 << 2);
                nib[u] = (live[bit >> 5] >> (bit & 31u)) & 15u;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            if (nib[u] == 15u) continue;
            const size_t q = q0 + 64u * u;
            uint8_t* g = p.grid + q * 16u;
            if (nib[u] == 0u) __builtin_nontemporal_store(z, reinterpret_cast<Zero4*>(g));
            else {
                // (a piece on the queued region's rim: plain stores -- four bytes that leave non-temporally reach the fabric as a partial write)
#pragma unroll
                for (uint32_t b = 0; b < 4u; ++b)
                    if (!((nib[u] >> b) & 1u)) *reinterpret_cast<uint32_t*>(g + 4u * b) = 0u;
            }
        }
        if (p.texels) {
            // the same bricks of the texel image: a piece is 64 bytes there.  Lane l of round k writes the (64 k + l)-th 16 bytes of the
            // wave's 4 KiB (consecutive lanes, consecutive bytes: a lane writing its own piece's four quarters would leave every store
            // instruction a quarter of each line) -- brick l & 3 of piece 16 k + (l >> 2), whose nibble lane 16 k + (l >> 2) holds
            const uint32_t lane = threadIdx.x;
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                uint32_t* t = p.texels + ((size_t)base + 64u * u) * 16u;
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    const uint32_t n = (uint32_t)__shfl((int)nib[u], (int)(16u * k + (lane >> 2)));
                    if (!((n >> (lane & 3u)) & 1u)) __builtin_nontemporal_store(z, reinterpret_cast<Zero4*>(t + (64u * k + lane) * 4u));
                }
            }
        }
    }
}
// ... and the clear as a kernel of its own (clearMode 0, and every grid whose side is no multiple of 16): the whole partition
__global__ __launch_bounds__(256) void k_clear_grid(VoxelizeParams p)
{
    plan_clear(p.grid, (size_t)p.N * p.N * p.nz, gridDim.x);
    if (p.texels) plan_clear(reinterpret_cast<uint8_t*>(p.texels), (size_t)p.N * p.N * p.nz * 4u, gridDim.x);
}

template <bool TEXELS>
__global__ __launch_bounds__(64, 6) void k_voxelize_listed(VoxelizeParams p, QueueLens lens, ClearShare clr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ int32_t stack[(TEXELS ? 20 : 16) * 64];     // per lane: 8 queued triangles of two words (TEXELS: then the closest hit's V, W, det, index)
    // The kernel's arguments, ALL asked for here: left to itself the compiler loads each word where it is first used, and the start of a
    // brick is then a chain of scalar-memory round trips each waited for before the next is asked for (the clear's share -> the sixteen
    // queue lengths -> the queue's address -> [the brick's word] -> the scene's words in two more batches).  Named here they are one batch of
    // loads behind one wait; only the brick's word itself is a second trip (-0.5 % on the full grid, -0.7 % on a rank's share).
    asm volatile("" :: "s"(lens.len[0]), "s"(lens.len[1]), "s"(lens.len[2]), "s"(lens.len[3]), "s"(lens.len[4]), "s"(lens.len[5]), "s"(lens.len[6]), "s"(lens.len[7]),
                 "s"(lens.heavy[0]), "s"(lens.heavy[1]), "s"(lens.heavy[2]), "s"(lens.heavy[3]), "s"(lens.heavy[4]), "s"(lens.heavy[5]), "s"(lens.heavy[6]),
                 "s"(lens.heavy[7]), "s"(clr.blocks), "s"(clr.where), "s"(clr.live), "s"(gridDim.x), "s"(p.queueSlots), "s"(p.queueCap));
    asm volatile("" :: "s"(p.N), "s"(p.z0), "s"(p.nz), "s"(p.zBlock), "s"(p.zPeriod), "s"(p.zShift), "s"(p.scene.dmCells), "s"(p.scene.dmEntries), "s"(p.scene.dmR),
                 "s"(p.scene.dmCoop), "s"(p.scene.triPos), "s"(p.grid), "s"(p.scene.rootLo[0]), "s"(p.scene.rootLo[1]), "s"(p.scene.rootLo[2]),
                 "s"(p.scene.rootHi[0]), "s"(p.scene.rootHi[1]), "s"(p.scene.rootHi[2]));
    uint32_t wg = blockIdx.x;
    if (clr.blocks) {
        // (clr.blocks is a multiple of 8: a brick workgroup's number keeps its residue mod 8 -- its XCD, its queue)
        const uint32_t bricks = gridDim.x - clr.blocks;
        if (clr.where == 3u) {
            // rows of 8 workgroups; of the launch's R rows C clear, spread evenly: row r clears iff floor((r + 1) C / R) > floor(r C / R),
            // and floor(r C / R) clearing rows lie in front of it -- the zeros leave as a trickle beside the bricks' loads, not as a burst
            const uint32_t r = wg >> 3, R = gridDim.x >> 3, C = clr.blocks >> 3;
            const uint32_t before = (uint32_t)(((uint64_t)r * C) / R), upto = (uint32_t)(((uint64_t)(r + 1u) * C) / R);
            if (upto != before) { clear_dead_bricks(p, clr.live, 8u * before + (wg & 7u), clr.blocks); return; }
            wg -= 8u * before;
        } else {
            const bool clears = clr.where == 1u ? wg < clr.blocks : wg >= bricks;
            if (clears) { clear_dead_bricks(p, clr.live, clr.where == 1u ? wg : wg - bricks, clr.blocks); return; }
            if (clr.where == 1u) wg -= clr.blocks;
        }
    }
    const uint32_t x = wg & 7u, k = wg >> 3;
#if defined(DXV_PHASE_TIMES)
    const unsigned long long tPhase0_ = __builtin_amdgcn_s_memrealtime();
#endif
#if defined(DXV_QUEUE_TIMES)
    const uint64_t tStart = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0u && 3u * wg + 2u < p.redoCap) { p.redo[3u * wg] = 0; p.redo[3u * wg + 1u] = 0; }
#endif
    // (x's equal share of the launch: its own queue's first bricks, then what longer queues hold beyond theirs -- queue_item)
    uint32_t qy, qslot;
    if (!queue_item(lens, p.queueCap, x, k, qy, qslot)) return;
    // the brick's word through the SCALAR cache (one word per wave; the queue was written long before this launch): as a vector load it was
    // a round trip through the busy vector-memory pipe (~1 us of a 10 us brick) in front of everything else the workgroup does
    uint32_t w;
    {
        const uint32_t* slot = p.queueSlots + (uint32_t)__builtin_amdgcn_readfirstlane((int)(qy * p.queueCap + qslot));     // (8 cap <= 2^27 bricks)
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(slot) : "memory");
    }
    const SceneView& sc = p.scene;
    const uint32_t N = p.N, nz = p.nz;
    const uint32_t bx = w & 1023u, by = (w >> 10) & 1023u, bz = w >> 20;
    const uint32_t tid = threadIdx.x;
    uint32_t ix = bx * 4u + (tid & 3u), iy = by * 4u + ((tid >> 2) & 3u), lz = bz * 4u + (tid >> 4);
    ix = ix < N ? ix : N - 1u; iy = iy < N ? iy : N - 1u; lz = lz < nz ? lz : nz - 1u;     // (lanes that hang over the grid's end trace a voxel of the grid and store nothing)
    const uint32_t iz = p.zBlock == nz ? p.z0 + lz : p.z0 + (lz >> p.zShift) * p.zPeriod + (lz & (p.zBlock - 1u));
    Ray r;
    ray_origin(N, ix, iy, iz, r.ox, r.oy, r.oz);
    const DirMapView dm{static_cast<const DirCell*>(sc.dmCells), static_cast<const DirEntry*>(sc.dmEntries), sc.dmR, sc.dmCoop};
    DirRayStart start = dm_ray_start(r.ox, r.oy, r.oz, dm);
    if (origin_leaves_root(r.ox, r.oy, r.oz, sc.rootLo, sc.rootHi)) start.live = false;
    Hit best;
    float bestDet = 1.0f;
    const StridedStack stk{stack + tid, 64};
#if defined(DXV_PHASE_TIMES)
    { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); if (threadIdx.x == 0u) { unsigned long long* slot_ = g_dxvPhase + (size_t)(blockIdx.x & (kPhaseSlots - 1u)) * 16u; atomicAdd(slot_, now_ - tPhase0_); atomicAdd(slot_ + 6, 1ull); } }
#endif
    // (with the texel image on, the closest hit's V, W, det and index wait in the LDS column: four registers that cost that variant its
    // seventh wave per SIMD; without it the allocator does better with them in registers: 68 against 74)
    trace_reference_dm_from<StridedStack, 0, TEXELS ? 1 : 2>(r, dm, start, sc.triPos, stk, 16, best, bestDet);
#if defined(DXV_PHASE_TIMES)
    const unsigned long long tPhase5_ = __builtin_amdgcn_s_memrealtime();
#endif
    uint32_t texel = 0;
    const uint8_t occ = TEXELS ? shade_reference_lds(sc, r, best.leaf, stk, 16, &texel) : shade_reference_again(sc, r, best.leaf);
    // the lane's voxel once more (nothing of it is kept through the body: with the texel image on, the lane's coordinates held across the
    // scan cost the kernel its seventh wave per SIMD -- k_voxelize_queue does the same)
    uint32_t lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    if (TEXELS || (N & 3u) != 0u) {
        const uint32_t vx = bx * 4u + (lane & 3u), vy = by * 4u + ((lane >> 2) & 3u), vz = bz * 4u + (lane >> 4);
        if (vx < N && vy < N && vz < nz) {
            const size_t id = ((size_t)vz * N + vy) * N + vx;
            if (TEXELS) p.texels[id] = texel;
            if ((N & 3u) != 0u) p.grid[id] = occ;
        }
    }
    if ((N & 3u) == 0u) {
        const uint64_t m = __builtin_amdgcn_ballot_w64(occ != 0);
        const uint32_t ry = by * 4u + (lane & 3u), rz = bz * 4u + ((lane >> 2) & 3u);
        if (lane < 16u && rz < nz) {
            const uint32_t nib = (uint32_t)(m >> (4u * lane)) & 15u;
            *reinterpret_cast<uint32_t*>(p.grid + ((size_t)rz * N + ry) * N + bx * 4u) = (nib * 0x00204081u) & 0x01010101u;
        }
    }
#if defined(DXV_PHASE_TIMES)
    { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); if (threadIdx.x == 0u) atomicAdd(g_dxvPhase + (size_t)(blockIdx.x & (kPhaseSlots - 1u)) * 16u + 5, now_ - tPhase5_); }
#endif
#if defined(DXV_QUEUE_TIMES)
    // (diagnostic build only, tools/wg_times.py: start and end of every workgroup in 100 MHz ticks, and the XCD it ran on)
    if (threadIdx.x == 0u && 3u * wg + 2u < p.redoCap) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.redo[3u * wg] = tStart; p.redo[3u * wg + 1u] = (__builtin_amdgcn_s_memrealtime() & 0x0fffffffffffffffull) | ((uint64_t)(xcc & 15u) << 60);
        p.redo[3u * wg + 2u] = w;
    }
#endif
#else
    (void)p;
#endif
}

// persistent waves the device holds at once (occupancy of the kernel x compute units), a multiple of 8
static uint32_t queue_waves(bool texels)
{
    static uint32_t cached[2] = {0, 0};
    uint32_t& c = cached[texels ? 1 : 0];
    if (c) return c;
    int dev = 0, cus = 0, perCu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const hipError_t e = texels ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_voxelize_queue<true>, 64, 0)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_voxelize_queue<false>, 64, 0);
    if (e != hipSuccess || perCu <= 0) { (void)hipGetLastError(); perCu = 24; }
    c = ((uint32_t)cus * (uint32_t)perCu + 7u) & ~7u;
    return c;
}

// Dynamic LDS a launch of k_voxelize_listed<false> asks for WITHOUT using it.  The kernel fits eight waves per SIMD (64 VGPRs, 4 KB of LDS per
// single-wave workgroup: 32 workgroups per CU); how many it should run depends on how a brick's rays fall on the lists' map.  Where they
// look into neighbouring texels (grid side >= 3/4 of the map's) the eighth wave is throughput: -9 % at 512^3, -12 % at 1024^3 against seven.
// Where a brick is spread over many texels (256^3 on the 512 map) it is more lines in flight per load and slower bricks: +7 %.  Such a
// launch is held at 28 workgroups per CU by LDS: the smallest pad that leaves so many, found once per value through the occupancy query
// (26 .. 30 measure the same: the hardware fills SIMDs evenly; profiles/r06/ab_listed_workgroups_per_cu.jsonl).  Option listedwaves overrides.
static uint32_t listed_lds_pad(const VoxelizeParams& p)
{
    if (p.texels) return 0u;                                           // (that variant holds 72 VGPRs: seven waves by itself)
    const uint32_t want = p.listedWaves ? p.listedWaves : (4u * p.N >= 3u * p.scene.dmR ? 32u : 28u);
    if (want >= 32u) return 0u;
    static int cached[33] = {0};                                        // 0: not asked yet; -1: no pad
    int& c = cached[want < 8u ? 8u : want];
    if (c == 0) {
        c = -1;
        int perCu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_voxelize_listed<false>, 64, 0) == hipSuccess && perCu > (int)want) {
            for (int pad = 64; pad <= 16384; pad += 64) {
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_voxelize_listed<false>, 64, (size_t)pad) != hipSuccess) break;
                if (perCu <= (int)want) { c = pad; break; }
            }
        }
        (void)hipGetLastError();
    }
    return c > 0 ? (uint32_t)c : 0u;
}

// rebuild: clear the grid and build the queue in front of the launch (a launch that may not rely on anything an earlier
// launch left behind); else the caller vouches that the frame's grid and queue are those of the same launch made before
// (same lists, partition and buffers: the kernel writes the same bricks every time) and only the queue heads are reset.
hipError_t launch_voxelize_queue(const VoxelizeParams& pin, bool rebuild, uint32_t* wavesOut, hipEvent_t* planEvents, const uint32_t* listedLens, hipStream_t s)
{
    QueueLens lens{};
    uint32_t listedLen = 0;
    if (listedLens) {
        uint32_t total = 0;
        for (int a = 0; a < 8; ++a) { lens.len[a] = listedLens[a]; lens.heavy[a] = listedLens[8 + a]; total += listedLens[a]; }
        listedLen = (total + 7u) / 8u;
    }
    const VoxelizeParams& p = pin;
    hipError_t e;
    if (!rebuild && listedLen) {
        // the queue as it stands, one workgroup per item of an XCD's equal share (listedLen = ceil(total / 8)) and per XCD
        if (wavesOut) *wavesOut = 8u * listedLen;
        if (p.texels) k_voxelize_listed<true><<<dim3(8u * listedLen), dim3(64), 0, s>>>(p, lens, ClearShare{nullptr, 0u, 0u});
        else k_voxelize_listed<false><<<dim3(8u * listedLen), dim3(64), listed_lds_pad(p), s>>>(p, lens, ClearShare{nullptr, 0u, 0u});
        return hipGetLastError();
    }
    if (rebuild) {
        if (!p.planClear) {
            if ((e = hipMemsetAsync(p.grid, 0, (size_t)p.N * p.N * p.nz, s)) != hipSuccess) return e;
            if (p.texels && (e = hipMemsetAsync(p.texels, 0, (size_t)p.N * p.N * p.nz * 4, s)) != hipSuccess) return e;
        }
        if (planEvents && (e = hipEventRecord(planEvents[0], s)) != hipSuccess) return e;
        if ((e = plan_build(p, s)) != hipSuccess) return e;
        if (planEvents && (e = hipEventRecord(planEvents[1], s)) != hipSuccess) return e;
    } else if ((e = hipMemsetAsync(p.queue + queue_head_word(0, 0), 0, sizeof(uint32_t) * (queue_len_word(0) - queue_head_word(0, 0)), s)) != hipSuccess) return e;   // the 64 heads
    const uint32_t held = queue_waves(p.texels != nullptr), sevenths = p.queueSevenths && p.queueSevenths < 7u ? p.queueSevenths : 7u;
    const uint32_t waves = p.queueWaves ? (p.queueWaves + 7u) & ~7u : (held * sevenths / 7u + 7u) & ~7u;     // (a multiple of 8, at least 8: every head has a home wave)
    if (wavesOut) *wavesOut = waves;
    if (p.texels) k_voxelize_queue<true><<<dim3(waves), dim3(64), 0, s>>>(p);
    else k_voxelize_queue<false><<<dim3(waves), dim3(64), 0, s>>>(p);
    return hipGetLastError();
}

#if defined(DXV_PHASE_TIMES)
// diagnostic build: the phase sums of the lists kernel (dxv_dirmap.h, DXV_PHASE) since the last reset
hipError_t phase_times_read(unsigned long long out[16], bool reset)
{
    std::vector<unsigned long long> all((size_t)kPhaseSlots * 16u);
    hipError_t e = hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(g_dxvPhase), all.size() * sizeof(unsigned long long));
    for (int k = 0; k < 16; ++k) out[k] = 0;
    for (size_t i = 0; i < all.size(); ++i) out[i & 15u] += all[i];
    if (e == hipSuccess && reset) {
        std::fill(all.begin(), all.end(), 0ull);
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_dxvPhase), all.data(), all.size() * sizeof(unsigned long long));
    }
    return e;
}
#endif

// A launch through a PREPARED queue (dxv_device.h): the queue is a pure function of (static scene's lists, grid, partition) and was
// built when those were fixed -- Init, dxv_prepare_launch -- like the lists themselves (the reference builds everything its frames
// trace through once, Content/Voxelizer.cpp:73, and a frame is one DispatchRays, :351-369).  The launch clears the grid and runs every
// queued brick: every voxel is written in every launch, nothing a launch reads was left behind by another LAUNCH.
hipError_t launch_voxelize_prepared(const VoxelizeParams& p, const uint32_t lens16[16], const uint32_t* live, int clearMode, uint32_t* wavesOut, hipStream_t s)
{
    QueueLens lens{};
    uint32_t total = 0;
    for (int a = 0; a < 8; ++a) { lens.len[a] = lens16[a]; lens.heavy[a] = lens16[8 + a]; total += lens16[a]; }
    const uint32_t listedLen = (total + 7u) / 8u;
    if (wavesOut) *wavesOut = 8u * listedLen;
    const size_t bytes = (size_t)p.N * p.N * p.nz;
    ClearShare clr{nullptr, 0u, 0u};
    if (clearMode != 0 && live && (p.N & 15u) == 0u && listedLen) {
        // 1,024 sixteen-byte pieces per clearing workgroup (16 per thread)
        const uint64_t pieces = bytes >> 4;
        uint64_t blocks = ((pieces + 1023u) >> 10);
        blocks = (blocks + 7u) & ~(uint64_t)7u;
        clr.live = live; clr.blocks = (uint32_t)blocks; clr.where = (uint32_t)clearMode;
    } else {
        // (about one workgroup of 256 threads per 64 KiB, at least 8 and at most 8,192)
        uint32_t nb = (uint32_t)((bytes + 65535u) >> 16);
        nb = nb < 8u ? 8u : nb > 8192u ? 8192u : nb;
        k_clear_grid<<<dim3(nb), dim3(256), 0, s>>>(p);
        if (!listedLen) return hipGetLastError();
    }
    const uint32_t wgs = 8u * listedLen + clr.blocks;
    if (p.texels) k_voxelize_listed<true><<<dim3(wgs), dim3(64), 0, s>>>(p, lens, clr);
    else k_voxelize_listed<false><<<dim3(wgs), dim3(64), listed_lds_pad(p), s>>>(p, lens, clr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Test hook (dxv_debug_list_check): the lists' superset claim, checked exhaustively on the device.  For every voxel the
// LBVH is walked WITHOUT distance culling; every triangle the canonical step accepts for the ray (own padded box passed,
// watertight hit at 0 < t < TMax, box entry <= t) must be found in the ray's texel list and pass that entry's integer
// test (box, edge, radial range) even with the radial cut already drawn at its own t, and lie in front of the point where a
// scan holding a hit at that t stops -- then no order of scanning, no
// cut by an earlier hit and no early stop can keep the closest hit out of the queue (dxv_dirmap.h).
// out[0] accepted (ray, triangle) pairs, out[1] violations, out[2 + 2 k], out[3 + 2 k]: voxel id and triangle slot of the
// first 16 violations.  Not a product path.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_list_check(VoxelizeParams p, unsigned long long* out)
{
    __shared__ int32_t stack[64 * 64];
    const uint32_t N = p.N, nbx = (N + 3u) / 4u;
    const uint32_t b = blockIdx.x, bx = b % nbx, by = (b / nbx) % nbx, bz = b / (nbx * nbx);
    const uint32_t lane = threadIdx.x;
    const uint32_t ix = bx * 4u + (lane & 3u), iy = by * 4u + ((lane >> 2) & 3u), lz = bz * 4u + (lane >> 4), iz = p.z0 + lz;
    if (ix >= N || iy >= N || lz >= p.nz) return;
    const SceneView& sc = p.scene;
    Ray r;
    ray_origin(N, ix, iy, iz, r.ox, r.oy, r.oz);
    if (origin_leaves_root(r.ox, r.oy, r.oz, sc.rootLo, sc.rootHi)) return;
    finish_ray_reference(r);
    ray_shear(r);
    const DirMapView dm{static_cast<const DirCell*>(sc.dmCells), static_cast<const DirEntry*>(sc.dmEntries), sc.dmR};
    const DirRayStart start = dm_ray_start(r.ox, r.oy, r.oz, dm);
    const DirCell cell = start.cell;
    const DirRayLocal loc = dm_ray_local(start.cx, start.cy);
    const float rho = start.rho, near = start.near;
    const size_t id = ((size_t)lz * N + iy) * N + ix;
    auto leaf = [&](int32_t l) {
        const TriPos tp = load_tri(sc.triPos, l);
        float lo[3], hi[3], tn, t, b1, b2;
        tri_box(tp.v0, tp.v1, tp.v2, lo, hi);
        if (!slab(r, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], tn)) return;
        if (!tri_test<false>(r, tp.v0, tp.v1, tp.v2, t, b1, b2) || tn > t) return;
        atomicAdd(out, 1ull);
        const uint32_t rc = dm_radial_word(near, (rho + t) * 1.001f + 1e-4f);
        bool found = false;
        const float step = dm_stop_step(half_bits_to_float(cell.thick)), bound = (rho + t) * 1.001f + 1e-4f;
        if (start.live)
            for (uint32_t k = cell.begin; k < cell.begin + cell.count && !found; ++k) {
                const DirEntry e = dm.entries[k];
                if (dm_stop_radius(e, step) > bound) break;             // a scan with this hit in hand would stop here: the entry must come before
                found = dm_entry_tri(e) == (uint32_t)l && dm_local_pass(e, loc, rc);
            }
        if (!found) {
            const unsigned long long slot = atomicAdd(out + 1, 1ull);
            if (slot < 16ull) { out[2 + 2 * slot] = (unsigned long long)id; out[3 + 2 * slot] = (unsigned long long)(uint32_t)l; }
        }
    };
    int32_t* stk = stack + lane;
    int sp = 0;
    int32_t node = 0;
    for (;;) {
        const NodePlanes n = load_node(sc.nodes, node);
        float tn0, tn1;
        const bool h0 = slab(r, n.b[0], n.b[1], n.b[2], n.b[3], n.b[4], n.b[5], tn0);
        const bool h1 = slab(r, n.b[6], n.b[7], n.b[8], n.b[9], n.b[10], n.b[11], tn1);
        if (h0 && n.c0 < 0) leaf(~n.c0);
        if (h1 && n.c1 < 0) leaf(~n.c1);
        const bool i0 = h0 && n.c0 >= 0, i1 = h1 && n.c1 >= 0;
        if (i0 && i1) { if (sp < 64) stk[64 * sp++] = n.c1; node = n.c0; }
        else if (i0) node = n.c0;
        else if (i1) node = n.c1;
        else {
            if (sp == 0) break;
            node = stk[64 * --sp];
        }
    }
}

hipError_t launch_list_check(const VoxelizeParams& p, unsigned long long* out, hipStream_t s)
{
    const uint32_t nb = (p.N + 3u) / 4u, nbz = (p.nz + 3u) / 4u;
    k_list_check<<<dim3(nb * nb * nbz), dim3(64), 0, s>>>(p, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Test hook (dxv_debug_class_check): the per-triangle class of the normal test (normal_class, dxv_math.h: "every ray of the
// rule that can hit this triangle gets the same answer from the predicate"), checked against the predicate itself for
// every closest hit of a grid.  The closest hit comes from the plain LBVH walk (no lists, no shortcut); when its triangle
// carries a class, the canonical predicate (hlsl:137-138: interpolated normal, normalize, dot > 0.12) is evaluated as for
// an unclassified triangle and must agree.  out[0] hits on classified triangles, out[1] disagreements, out[2] all hits,
// out[3 + 2 k], out[4 + 2 k]: voxel id and triangle slot of the first 15 disagreements.  Not a product path.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_class_check(VoxelizeParams p, unsigned long long* out)
{
    __shared__ int32_t stack[64 * 64];
    const uint32_t N = p.N, nbx = (N + 3u) / 4u;
    const uint32_t b = blockIdx.x, bx = b % nbx, by = (b / nbx) % nbx, bz = b / (nbx * nbx);
    const uint32_t lane = threadIdx.x;
    const uint32_t ix = bx * 4u + (lane & 3u), iy = by * 4u + ((lane >> 2) & 3u), lz = bz * 4u + (lane >> 4), iz = p.z0 + lz;
    if (ix >= N || iy >= N || lz >= p.nz) return;
    const SceneView& sc = p.scene;
    Ray r;
    ray_origin(N, ix, iy, iz, r.ox, r.oy, r.oz);
    if (origin_leaves_root(r.ox, r.oy, r.oz, sc.rootLo, sc.rootHi)) return;
    finish_ray_reference(r);
    const StridedStack stk{stack + lane, 64};
    Hit best;
    if (!trace_reference(r, sc.nodes, sc.triPos, stk, 64, best)) { atomicAdd(out + 1, 1ull); return; }     // (cannot happen: 64 >= any tree height)
    if (best.k == 0xffffffffu) return;
    atomicAdd(out + 2, 1ull);
    const TriPos tp = load_tri(sc.triPos, best.leaf);
    const uint32_t cls = __builtin_bit_cast(uint32_t, tp.v1.w) >> kClassShift;
    if (cls == 0u) return;
    atomicAdd(out, 1ull);
    const TriNrm tn = sc.triNrm[best.leaf];
    float nx, ny, nz;
    const bool in = predicate(r, tn.n0, tn.n1, tn.n2, best.b1, best.b2, nx, ny, nz);
    if (in != (cls == kClassIn)) {
        const unsigned long long slot = atomicAdd(out + 1, 1ull);
        if (slot < 15ull) { out[3 + 2 * slot] = ((unsigned long long)lz * N + iy) * N + ix; out[4 + 2 * slot] = (unsigned long long)(uint32_t)best.leaf; }
    }
}

hipError_t launch_class_check(const VoxelizeParams& p, unsigned long long* out, hipStream_t s)
{
    const uint32_t nb = (p.N + 3u) / 4u, nbz = (p.nz + 3u) / 4u;
    k_class_check<<<dim3(nb * nb * nbz), dim3(64), 0, s>>>(p, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Test hook (dxv_debug_far_check): the brick test of the launches over the brick box -- "no ray of a brick the test calls dead hits
// anything" -- checked exhaustively: for every brick of slices [p.z0, p.z0 + p.nz) the test k_voxelize makes (dm_box_may_be_live
// against p.mip), and for every voxel of a brick it calls dead the plain LBVH walk without any shortcut but the provable root
// early-out.  out[0] bricks, out[1] bricks called dead, out[2] their rays walked, out[3] rays among them with a hit (must be 0),
// out[4 + k]: voxel id of the first 8.  Not a product path.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_far_check(VoxelizeParams p, unsigned long long* out)
{
    __shared__ int32_t stack[64 * 64];
    const uint32_t N = p.N, nbx = (N + 3u) / 4u;
    const uint32_t b = blockIdx.x, bx = b % nbx, by = (b / nbx) % nbx, bz = b / (nbx * nbx);
    const uint32_t lane = threadIdx.x;
    float x0, x1, y0, y1, z0, z1;
    dm_brick_hull(N, p.nz, p.z0, p.nz, 0u, p.nz, bx, by, bz, x0, x1, y0, y1, z0, z1);
    const bool live = dm_box_may_be_live(x0, x1, y0, y1, z0, z1, p.scene.rootLo, p.scene.rootHi, p.mip, p.mipR);
    if (lane == 0u) { atomicAdd(out, 1ull); if (!live) atomicAdd(out + 1, 1ull); }
    if (live) return;
    const uint32_t ix = bx * 4u + (lane & 3u), iy = by * 4u + ((lane >> 2) & 3u), lz = bz * 4u + (lane >> 4), iz = p.z0 + lz;
    if (ix >= N || iy >= N || lz >= p.nz) return;
    const SceneView& sc = p.scene;
    Ray r;
    ray_origin(N, ix, iy, iz, r.ox, r.oy, r.oz);
    if (origin_leaves_root(r.ox, r.oy, r.oz, sc.rootLo, sc.rootHi)) return;
    finish_ray_reference(r);
    atomicAdd(out + 2, 1ull);
    const StridedStack stk{stack + lane, 64};
    Hit best;
    const bool done = trace_reference(r, sc.nodes, sc.triPos, stk, 64, best);
    if (!done || best.k != 0xffffffffu) {
        const unsigned long long slot = atomicAdd(out + 3, 1ull);
        if (slot < 8ull) out[4 + slot] = ((unsigned long long)lz * N + iy) * N + ix;
    }
}
hipError_t launch_far_check(const VoxelizeParams& p, unsigned long long* out, hipStream_t s)
{
    const uint32_t nb = (p.N + 3u) / 4u, nbz = (p.nz + 3u) / 4u;
    k_far_check<<<dim3(nb * nb * nbz), dim3(64), 0, s>>>(p, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Test hook (dxv_debug_division_check): the ray set-up's scale-free divisions (dxv_math.h: rcp_refined / div_by) against the IEEE
// quotient `/` the host computes, for EVERY voxel origin of an N^3 grid: origin (grids whose side is no power of two divide by N), the
// cube-map point (u, v) and start radius, direction, 1 / direction, the three shear constants -- 15 words per voxel, compared bit for bit.
// out[0] voxels, out[1] voxels with a differing word (must be 0), out[2 + k]: id of the first 6.  Not a product path.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_division_check(uint32_t N, unsigned long long* out)
{
    const uint32_t ix = blockIdx.x * 256u + threadIdx.x, iy = blockIdx.y, iz = blockIdx.z;      // (one grid row per (y, z): no 64-bit index arithmetic)
    if (ix >= N) return;
    const uint64_t id = ((uint64_t)iz * N + iy) * N + ix;
    // the product's own code
    Ray r;
    ray_origin(N, ix, iy, iz, r.ox, r.oy, r.oz);
    uint32_t face;
    float u, v, rho;
    dm_ray_point(r.ox, r.oy, r.oz, face, u, v, rho);
    finish_ray_reference(r, rho);
    ray_shear_finished(r);
    // the same with IEEE quotients
    const float fn = (float)N;
    float ox, oy, oz;
    if ((N & (N - 1u)) == 0u) { ox = r.ox; oy = r.oy; oz = r.oz; }      // (a power of two multiplies by an exact reciprocal: no division there)
    else {
        ox = ((float)ix + 0.5f) / fn * 2.0f - 1.0f;
        oy = -(((float)iy + 0.5f) / fn * 2.0f - 1.0f);
        oz = ((float)iz + 0.5f) / fn * 2.0f - 1.0f;
    }
    const float ax = __builtin_fabsf(ox), ay = __builtin_fabsf(oy), az = __builtin_fabsf(oz);
    float wu, wv;
    if (ax >= ay && ax >= az) { wu = oy / ax; wv = oz / ax; }
    else if (ay >= az) { wu = oz / ay; wv = ox / ay; }
    else { wu = ox / az; wv = oy / az; }
    const float len = __builtin_sqrtf((ox * ox + oy * oy) + oz * oz);
    const float dx = ox / len, dy = oy / len, dz = oz / len;
    const float ivx = 1.0f / dx, ivy = 1.0f / dy, ivz = 1.0f / dz;
    int kz = 0;
    float m = abs_(dx);
    if (abs_(dy) > m) { kz = 1; m = abs_(dy); }
    if (abs_(dz) > m) { kz = 2; }
    int kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1;
    const float dkz = sel3(dx, dy, dz, kz);
    if (dkz < 0.0f) { const int t = kx; kx = ky; ky = t; }
    const float Sx = sel3(dx, dy, dz, kx) / dkz, Sy = sel3(dx, dy, dz, ky) / dkz, Sz = 1.0f / dkz;
    auto ne = [](float a, float b) { return __builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, b); };
    const bool bad = ne(ox, r.ox) || ne(oy, r.oy) || ne(oz, r.oz) || ne(wu, u) || ne(wv, v) || ne(len, rho) || ne(dx, r.dx) || ne(dy, r.dy) || ne(dz, r.dz) ||
                     ne(ivx, r.ivx) || ne(ivy, r.ivy) || ne(ivz, r.ivz) || ne(Sx, r.Sx) || ne(Sy, r.Sy) || ne(Sz, r.Sz) || kz != r.kz;
    // (the count of voxels: one add per grid SLICE -- an add per wave on one word was 90 % of this kernel's time)
    if (blockIdx.x == 0u && blockIdx.y == 0u && threadIdx.x == 0u) atomicAdd(out, (unsigned long long)N * N);
    const unsigned long long mb = __ballot(bad);
    if (mb && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(mb)) atomicAdd(out + 1, (unsigned long long)__builtin_popcountll(mb));
    if (bad) {
        const unsigned long long slot = atomicAdd(out + 8, 1ull);
        if (slot < 6ull) out[2 + slot] = id;
    }
}
hipError_t launch_division_check(uint32_t N, unsigned long long* out, hipStream_t s)
{
    k_division_check<<<dim3((N + 255u) / 256u, N, N), dim3(256), 0, s>>>(N, out);
    return hipGetLastError();
}

// The rays whose LDS column was too small in k_voxelize (a few per million: DESIGN.md), one per
// lane with a column of kRedoStack entries -- enough for any tree the builder makes (height <= 62).
// Plain binary walk, leaves tested where they are met; same voxel as every other walk.
constexpr int kRedoStack = 64;
template <int MODE, bool TEXELS>
__global__ __launch_bounds__(64) void k_voxelize_redo(VoxelizeParams p)
{
    __shared__ int32_t stack[kRedoStack * 64];
    const uint32_t mine = 1u + p.redoParity, other = 2u - p.redoParity;
    uint32_t count = p.status[mine];
    if (count > p.redoCap) count = p.redoCap;
    if (blockIdx.x == 0 && threadIdx.x == 0) p.status[other] = 0;      // the next launch appends there
    const StridedStack stk{stack + threadIdx.x, 64};
    const uint64_t plane = (uint64_t)p.N * p.N;
    for (uint32_t i = blockIdx.x * 64u + threadIdx.x; i < count; i += gridDim.x * 64u) {
        const uint64_t id = p.redo[i];
        const uint32_t lz = (uint32_t)(id / plane), rem = (uint32_t)(id % plane), iy = rem / p.N, ix = rem % p.N;
        const uint32_t iz = p.zBlock == p.nz ? p.z0 + lz : p.z0 + (lz >> p.zShift) * p.zPeriod + (lz & (p.zBlock - 1u));
        bool overflow = false;
        uint8_t occ;
        if (MODE == 0) {
            uint32_t texel = 0;
            occ = voxel_reference<0>(p.scene, p.N, ix, iy, iz, stk, kRedoStack, TEXELS ? &texel : nullptr, overflow);
            if (TEXELS) p.texels[id] = texel;
        } else occ = voxel_parity<false>(p.scene, p.N, ix, iy, iz, stk, kRedoStack, overflow);
        if (overflow) atomicOr(p.status, 1u);
        p.grid[id] = occ;
    }
}

hipError_t launch_voxelize_redo(const VoxelizeParams& p, hipStream_t s)
{
    const dim3 g(128), b(64);
    if (p.mode == 0) {
        if (p.texels) k_voxelize_redo<0, true><<<g, b, 0, s>>>(p);
        else k_voxelize_redo<0, false><<<g, b, 0, s>>>(p);
    } else k_voxelize_redo<1, false><<<g, b, 0, s>>>(p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Parity mode, row kernel: one wavefront per run of 64*CH voxels of one grid row.  The tree walk
// is wave-uniform (it depends on the row and the run's left end only): node and triangle records
// arrive through the scalar cache into SGPRs, the stack is one LDS column per wave, branches are
// scalar.  Lanes only diverge in data: lane l owns voxels x0 + 64 c + l (c < CH) and evaluates
// parity_row_voxel for them.  Same per-voxel results as k_voxelize<..., MODE 1> (tests), an
// order of magnitude fewer node visits.
// ---------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ TriPos load_tri_scalar(const TriPos* tris, int32_t uniformLeaf)
{
    const char* p = reinterpret_cast<const char*>(tris) + (uint64_t)(uint32_t)uniformLeaf * 48u;
    uint64_t w0, w1, w2, w3, w4, w5;
    asm volatile("s_load_dwordx2 %0, %6, 0x0\n\ts_load_dwordx2 %1, %6, 0x8\n\ts_load_dwordx2 %2, %6, 0x10\n\t"
                 "s_load_dwordx2 %3, %6, 0x18\n\ts_load_dwordx2 %4, %6, 0x20\n\ts_load_dwordx2 %5, %6, 0x28\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(w0), "=&s"(w1), "=&s"(w2), "=&s"(w3), "=&s"(w4), "=&s"(w5) : "s"(p) : "memory");
    auto lo = [](uint64_t v) { return __builtin_bit_cast(float, (uint32_t)v); };
    auto hi = [](uint64_t v) { return __builtin_bit_cast(float, (uint32_t)(v >> 32)); };
    TriPos t;
    t.v0 = F4{lo(w0), hi(w0), lo(w1), hi(w1)};
    t.v1 = F4{lo(w2), hi(w2), lo(w3), hi(w3)};
    t.v2 = F4{lo(w4), hi(w4), lo(w5), hi(w5)};
    return t;
}

struct WaveStack {
    int32_t* base;   // LDS, one column per wave
    __device__ __forceinline__ void push(int& sp, int32_t v) { base[sp++] = v; }
    __device__ __forceinline__ int32_t pop(int& sp) { return __builtin_amdgcn_readfirstlane(base[--sp]); }
};
#endif

// RB = rows per side of the block of grid rows a wave owns: 1 (one row), 2 or 4.  The RB x RB rows
// share one walk over the union of their y/z: up to 2.7x faster where triangles span several
// voxels, slower where they are voxel sized (every visited triangle is set up once per row it
// might cross) -- the launcher decides by the mean triangle extent.
// WIDE: the walk takes the four-box nodes (Node64) -- half as many dependent scalar fetches, which is
// what the walk waits on (triangle arithmetic is 6 % of the kernel).
// LISTS (RB = 1): the candidates of a row come from the row lists of the parity rule (dirmap.hip) -- one cell, then the
// triangles of its list four at a time -- instead of from a walk of the tree.
template <int CH, int RB, bool WIDE, bool LISTS = false>
__global__ __launch_bounds__(64, RB == 1 ? 8 : 6) void k_parity_rows(VoxelizeParams p)   // <= 64 / 80 VGPRs
{
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass only needs the stub: the body uses SGPR inline asm)
    static_assert(RB == 1 || RB == 2 || RB == 4, "1, 2 x 2 or 4 x 4 rows");
    constexpr int ROWS = RB * RB, WORDS = (ROWS * CH + 31) / 32;
    static_assert(32 % CH == 0, "a row's parity bits do not straddle registers");
    __shared__ int32_t stack[64];
    const uint32_t N = p.N;
    const uint32_t segLen = 64u * CH, nseg = (N + segLen - 1) / segLen;
    // blocks of RB x RB rows (y, z); rows past the end of the grid or slab repeat the last one
    // (same values written twice)
    const uint32_t by = (N + RB - 1u) / RB, bz = (p.nz + RB - 1u) / RB;
    const uint32_t nblocks = by * bz, nwaves = nblocks * nseg;
    const uint32_t rb = p.regionBits;
    const uint32_t j = blockIdx.x >> 3;
    const uint32_t lin = ((((j >> rb) << 3) | (blockIdx.x & 7u)) << rb) | (j & ((1u << rb) - 1u));
    if (lin >= nwaves) return;
    const uint32_t seg = lin % nseg;
    uint32_t blk = lin / nseg, biy, blz;
    constexpr uint32_t TS = RB == 4 ? 4u : 8u / RB, TB = RB == 1 ? 3u : 2u;   // 8 x 8 (16 x 16) rows per tile: neighbours share tree paths
    if (!(by & (TS - 1u)) && !(bz & (TS - 1u))) {
        const uint32_t tile = blk >> (2u * TB), in = blk & (TS * TS - 1u), tx = by >> TB;
        biy = (tile % tx) * TS + (in & (TS - 1u));
        blz = (tile / tx) * TS + (in >> TB);
    } else { biy = blk % by; blz = blk / by; }
    const uint32_t lane = threadIdx.x, x0 = seg * segLen;

    uint32_t iy[RB], lz[RB];
    float oy[RB], oz[RB], oxMin = 0.0f, t0, t1;
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        iy[k] = biy * RB + k < N ? biy * RB + k : N - 1u;
        lz[k] = blz * RB + k < p.nz ? blz * RB + k : p.nz - 1u;
        const uint32_t iz = p.zBlock == p.nz ? p.z0 + lz[k] : p.z0 + (lz[k] >> p.zShift) * p.zPeriod + (lz[k] & (p.zBlock - 1u));
        ray_origin(N, x0, iy[k], iz, oxMin, oy[k], t0);
        ray_origin(N, x0, iy[0], iz, t0, t1, oz[k]);
    }
    float ox[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) ray_origin(N, x0 + 64u * c + lane, iy[0], p.z0, ox[c], t0, t1);
    // lane r < ROWS carries the origin of row r = ry + RB * rz (the other lanes repeat rows; RB > 1 only)
    float oyLane = oy[0], ozLane = oz[0];
#pragma unroll
    for (int k = 1; k < RB; ++k) {
        if ((lane % ROWS) % RB == (uint32_t)k) oyLane = oy[k];
        if ((lane % ROWS) / RB == (uint32_t)k) ozLane = oz[k];
    }
    uint32_t bits[WORDS];                              // parity of voxel (row r = ry + RB * rz, run c) in bit r * CH + c
#pragma unroll
    for (int w = 0; w < WORDS; ++w) bits[w] = 0;
    float ylo = oy[0], yhi = oy[0], zlo = oz[0], zhi = oz[0];
#pragma unroll
    for (int k = 1; k < RB; ++k) { ylo = min_(ylo, oy[k]); yhi = max_(yhi, oy[k]); zlo = min_(zlo, oz[k]); zhi = max_(zhi, oz[k]); }
    const SceneView& sc = p.scene;
    if (sc.rootLo[1] <= yhi && ylo <= sc.rootHi[1] && sc.rootLo[2] <= zhi && zlo <= sc.rootHi[2] && sc.rootHi[0] >= oxMin) {
        WaveStack stk{stack};
        // Node tests in the half domain: a stored plane a is a half, so a <= y holds exactly when
        // a <= half_down(y), and y <= a exactly when half_up(y) <= a.  The five bounds are rounded
        // once per wave.  A word of the node holds one plane of BOTH children, so the five
        // differences "how far outside" are five packed half subtractions, their maximum four packed
        // max, and a child is met when its half of the result is <= 0 (the difference of two halves is
        // a multiple of 2^-24, so rounding never turns a non-zero difference into zero or flips its
        // sign).  9 vector + 7 scalar instructions per node; written as ten float comparisons the
        // test was a convert, a compare, a select and a readfirstlane each.
        typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
        auto H2 = [](uint32_t w) { return __builtin_bit_cast(half2_t, w); };
        auto both = [](uint32_t h) { return (h & 0xffffu) | (h << 16); };
        const half2_t ydn = H2(both(half_down(yhi))), yup = H2(both(half_up(ylo))), zdn = H2(both(half_down(zhi)));
        const half2_t zup = H2(both(half_up(zlo))), xup = H2(both(half_up(oxMin)));
        auto triangle = [&](const TriPos& tp) {
                if (RB == 1) {
                    const ParityRowTri s = parity_row_setup(oy[0], oz[0], tp.v0, tp.v1, tp.v2);
                    if (s.hit) {
                        uint32_t hits = 0;
#pragma unroll
                        for (int c = 0; c < CH; ++c) hits |= (parity_row_voxel(s, ox[c]) ? 1u : 0u) << c;
                        bits[0] ^= hits;
                    }
                } else {
                    // The per-row set-up is the same arithmetic for every row of the block: lane r does it
                    // for row r (all at once, instead of once per row on wave-uniform values), the rows
                    // that the triangle can cross are then taken one by one, their eight set-up values
                    // broadcast from their lane.
                    const ParityRowTri mine = parity_row_setup(oyLane, ozLane, tp.v0, tp.v1, tp.v2);
                    uint64_t rows = __builtin_amdgcn_ballot_w64(mine.hit) & ((1ull << ROWS) - 1ull);
                    while (rows) {
                        const int r = __builtin_ctzll(rows);
                        rows &= rows - 1ull;
                        auto bc = [r](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r)); };
                        ParityRowTri s;
                        s.U = bc(mine.U); s.V = bc(mine.V); s.W = bc(mine.W); s.det = bc(mine.det);
                        s.v0x = tp.v0.x; s.v1x = tp.v1.x; s.v2x = tp.v2.x; s.hix = bc(mine.hix); s.hit = true;
                        uint32_t hits = 0;
#pragma unroll
                        for (int c = 0; c < CH; ++c) hits |= (parity_row_voxel(s, ox[c]) ? 1u : 0u) << c;
                        const uint32_t at = (uint32_t)r * CH, word = at >> 5, contrib = hits << (at & 31u);
#pragma unroll
                        for (int w = 0; w < WORDS; ++w) bits[w] ^= word == (uint32_t)w ? contrib : 0u;
                    }
                }
        };
        auto triAt = [&](int32_t leaf) { return load_tri_scalar(sc.triPos, leaf); };
        auto outside = [&](uint32_t xh, uint32_t yl, uint32_t yh, uint32_t zl, uint32_t zh) {   // two children per word; > 0: outside
            half2_t m = __builtin_elementwise_max(__builtin_elementwise_max(H2(yl) - ydn, yup - H2(yh)),
                                                  __builtin_elementwise_max(H2(zl) - zdn, zup - H2(zh)));
            m = __builtin_elementwise_max(m, xup - H2(xh));
            return (uint32_t)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, m));
        };
        if (LISTS) {
            static_assert(!LISTS || RB == 1, "row lists: one row per wave");
            const uint32_t R = sc.plR;
            const uint32_t cell = (uint32_t)__builtin_amdgcn_readfirstlane((int)(dm_texel(oz[0], R) * R + dm_texel(oy[0], R)));
            const uint32_t begin = sc.plCells[2u * cell], count = sc.plCells[2u * cell + 1u];
            const uint32_t* list = sc.plEntries + begin;
            for (uint32_t k = 0; k < count; k += 4u) {
                // four triangle records in flight (the words behind the end of a list are the next list's or the buffer's
                // spare ones: valid slots either way, fetched and not used)
                const uint32_t s0 = list[k], s1 = list[k + 1u], s2 = list[k + 2u], s3 = list[k + 3u];
                const TriPos t0 = load_tri(sc.triPos, (int32_t)s0), t1 = load_tri(sc.triPos, (int32_t)s1);
                const TriPos t2 = load_tri(sc.triPos, (int32_t)s2), t3 = load_tri(sc.triPos, (int32_t)s3);
                triangle(t0);
                if (k + 1u < count) triangle(t1);
                if (k + 2u < count) triangle(t2);
                if (k + 3u < count) triangle(t3);
            }
        } else if (WIDE) {
            walk_parity_rows_wide(
                [&](int32_t i) {
                    const WideSgpr n = load_wide_scalar(sc.wide, i);   // words: x lo, x hi, y lo, y hi, z lo, z hi (children 0,1 | 2,3), links
                    const uint32_t o01 = outside((uint32_t)n.w[1], (uint32_t)n.w[2], (uint32_t)n.w[3], (uint32_t)n.w[4], (uint32_t)n.w[5]);
                    const uint32_t o23 = outside((uint32_t)(n.w[1] >> 32), (uint32_t)(n.w[2] >> 32), (uint32_t)(n.w[3] >> 32),
                                                 (uint32_t)(n.w[4] >> 32), (uint32_t)(n.w[5] >> 32));
                    WideHits r;
                    r.h[0] = (o01 & 0x8000u) != 0u || (o01 & 0x7fffu) == 0u;
                    r.h[1] = (o01 & 0x80000000u) != 0u || (o01 & 0x7fff0000u) == 0u;
                    r.h[2] = (o23 & 0x8000u) != 0u || (o23 & 0x7fffu) == 0u;
                    r.h[3] = (o23 & 0x80000000u) != 0u || (o23 & 0x7fff0000u) == 0u;
                    r.c[0] = (int32_t)(uint32_t)n.w[6]; r.c[1] = (int32_t)(uint32_t)(n.w[6] >> 32);
                    r.c[2] = (int32_t)(uint32_t)n.w[7]; r.c[3] = (int32_t)(uint32_t)(n.w[7] >> 32);
                    return r;
                },
                triAt, stk, triangle);
        } else {
            walk_parity_rows(
                [&](int32_t i) {
                    const NodeSgpr n = load_node_scalar(sc.nodes, i);  // words: x lo, x hi | y lo, y hi | z lo, z hi | links
                    const uint32_t out = outside((uint32_t)(n.w[0] >> 32), (uint32_t)n.w[1], (uint32_t)(n.w[1] >> 32), (uint32_t)n.w[2],
                                                 (uint32_t)(n.w[2] >> 32));
                    NodeHits r;
                    r.h0 = (out & 0x8000u) != 0u || (out & 0x7fffu) == 0u;
                    r.h1 = (out & 0x80000000u) != 0u || (out & 0x7fff0000u) == 0u;
                    r.c0 = (int32_t)(uint32_t)n.w[3];
                    r.c1 = (int32_t)(uint32_t)(n.w[3] >> 32);
                    return r;
                },
                triAt, stk, triangle);
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const size_t rowBase = ((size_t)lz[r / RB] * N + iy[r % RB]) * N;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint32_t ix = x0 + 64u * c + lane;
            if (ix < N) p.grid[rowBase + ix] = (uint8_t)((bits[(r * CH) / 32] >> ((r * CH) % 32 + c)) & 1u);
        }
    }
#else
    (void)p;
#endif
}

template <int CH, int RB, bool WIDE, bool LISTS = false>
static hipError_t launch_parity_rows_ch(const VoxelizeParams& pin, hipStream_t s)
{
    VoxelizeParams p = pin;
    const uint32_t segLen = 64u * CH, nseg = (p.N + segLen - 1) / segLen;
    const uint64_t nwaves = (uint64_t)((p.N + RB - 1u) / RB) * ((p.nz + RB - 1u) / RB) * nseg;
    uint32_t rb = p.regionBits;
    while (rb > 0 && (8ull << rb) > nwaves) --rb;
    p.regionBits = rb;
    const uint64_t span = 8ull << rb;
    const uint64_t grid = (nwaves + span - 1) / span * span;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    k_parity_rows<CH, RB, WIDE, LISTS><<<dim3((uint32_t)grid), dim3(64), 0, s>>>(p);
    return hipGetLastError();
}

template <int RB, bool WIDE>
static hipError_t launch_parity_rows_rb(const VoxelizeParams& p, hipStream_t s)
{
    if (p.N <= 64) return launch_parity_rows_ch<1, RB, WIDE>(p, s);
    if (p.N <= 128) return launch_parity_rows_ch<2, RB, WIDE>(p, s);
    if (p.N <= 256) return launch_parity_rows_ch<4, RB, WIDE>(p, s);
    return launch_parity_rows_ch<8, RB, WIDE>(p, s);    // 512 voxels per wave; longer rows take several waves
}

// rowBlock: rows per side of a wave's block of rows (1, 2 or 4); the walk takes the four-box nodes when the scene has them
hipError_t launch_parity_rows(const VoxelizeParams& p, int rowBlock, hipStream_t s)
{
    if (p.scene.plCells) {                                             // row lists: one row per wave, no walk
        if (p.N <= 64) return launch_parity_rows_ch<1, 1, false, true>(p, s);
        if (p.N <= 128) return launch_parity_rows_ch<2, 1, false, true>(p, s);
        if (p.N <= 256) return launch_parity_rows_ch<4, 1, false, true>(p, s);
        return launch_parity_rows_ch<8, 1, false, true>(p, s);
    }
    if (p.scene.wide) {
        if (rowBlock == 4) return launch_parity_rows_rb<4, true>(p, s);
        if (rowBlock == 2) return launch_parity_rows_rb<2, true>(p, s);
        return launch_parity_rows_rb<1, true>(p, s);
    }
    if (rowBlock == 4) return launch_parity_rows_rb<4, false>(p, s);
    if (rowBlock == 2) return launch_parity_rows_rb<2, false>(p, s);
    return launch_parity_rows_rb<1, false>(p, s);
}

// brick shapes: (x, y, z) voxels per workgroup; a wavefront owns 64 consecutive threads of it
using Brick0 = Brick<64, 4, 1>;    // 256 threads, wave = 64x1x1 row
using Brick1 = Brick<8, 8, 4>;     // 256 threads, wave = 8x8x1 tile
using Brick2 = Brick<4, 4, 16>;    // 256 threads, wave = 4x4x4 cube
using Brick3 = Brick<16, 4, 4>;    // 256 threads, wave = 16x4x1
using Brick4 = Brick<4, 4, 4>;     // 64 threads,  one wave per workgroup
using Brick5 = Brick<8, 8, 1>;     // 64 threads
using Brick6 = Brick<4, 4, 8>;     // 128 threads
using Brick7 = Brick<8, 4, 2>;     // 64 threads

int num_brick_shapes() { return 8; }

// Local slice index -> global slice (contiguous slab or block-cyclic partition).
static uint32_t global_slice(const VoxelizeParams& p, uint32_t lz)
{
    return p.zBlock == p.nz ? p.z0 + lz : p.z0 + (lz >> p.zShift) * p.zPeriod + (lz & (p.zBlock - 1u));
}

// Voxel index ranges [lo, hi] per axis outside of which origin_leaves_root() is certain (evaluated
// with the very same float formulas on the host).  Returns false when no voxel can be non-zero.
static bool live_ranges(const VoxelizeParams& p, uint32_t lo[3], uint32_t hi[3])
{
    for (int a = 0; a < 3; ++a) {
        const uint32_t n = a == 2 ? p.nz : p.N;
        bool any = false;
        for (uint32_t i = 0; i < n; ++i) {
            float o[3];
            const uint32_t g = a == 2 ? global_slice(p, i) : i;
            ray_origin(p.N, a == 0 ? g : 0, a == 1 ? g : 0, a == 2 ? g : 0, o[0], o[1], o[2]);
            if (p.mode == 0 ? axis_leaves_root(o[a], p.scene.rootLo[a], p.scene.rootHi[a])
                            : (a == 0 ? !(p.scene.rootHi[0] >= o[0]) : !(p.scene.rootLo[a] <= o[a] && o[a] <= p.scene.rootHi[a])))
                continue;
            if (!any) lo[a] = i;
            hi[a] = i;
            any = true;
        }
        if (!any) return false;
    }
    return true;
}

template <class B, int STACK>
static hipError_t launch_shape(const VoxelizeParams& pin, hipStream_t s)
{
    VoxelizeParams p = pin;
    const uint32_t tbx = (p.N + B::x - 1) / B::x, tby = (p.N + B::y - 1) / B::y, tbz = (p.nz + B::z - 1) / B::z;
    // launch only the bricks the root early-out cannot clear; everything else is zero by memset
    uint32_t lo[3], hi[3];
    const bool live = live_ranges(p, lo, hi);
    uint32_t b0[3] = {0, 0, 0}, b1[3] = {tbx, tby, tbz};
    if (live && p.subbox) {
        const uint32_t bs[3] = {(uint32_t)B::x, (uint32_t)B::y, (uint32_t)B::z}, tb[3] = {tbx, tby, tbz};
        for (int a = 0; a < 3; ++a) {
            b0[a] = (lo[a] / bs[a]) & ~7u;                       // 8-brick alignment keeps Morton locality
            b1[a] = (hi[a] / bs[a] + 1u + 7u) & ~7u;
            if (b1[a] > tb[a]) b1[a] = tb[a];
        }
    }
    // worth it only when a good part of the grid goes away: the cleared bricks are cheap (their waves
    // fill idle slots) while the memset is serial (measured: -5 % on a thin mesh, +3 % on a full one)
    if (live && (uint64_t)(b1[0] - b0[0]) * (b1[1] - b0[1]) * (b1[2] - b0[2]) * 10u > (uint64_t)tbx * tby * tbz * 6u) {
        b0[0] = b0[1] = b0[2] = 0;
        b1[0] = tbx; b1[1] = tby; b1[2] = tbz;
    }
    const bool partial = !live || b0[0] || b0[1] || b0[2] || b1[0] != tbx || b1[1] != tby || b1[2] != tbz;
    // The memset of a partial launch (the whole grid: 134 MB, 35 us at 512^3) is still good when the frame's last writer was
    // the same partial launch -- same grid, slab, partition, brick box, rule and buffers: the kernel only ever writes inside
    // the box.  The frame's signature word says so; every other writer of the grid resets it (full launches here, the row
    // kernel and reallocations in dxv_api.hip).
    uint64_t sig = 0;
    if (partial && live) {
        auto mix = [&](uint64_t v) { sig = (sig ^ v) * 0x9E3779B97F4A7C15ull; sig ^= sig >> 29; };
        mix(p.N); mix(p.nz); mix(p.z0); mix(p.zBlock); mix(p.zPeriod); mix((uint64_t)p.mode);
        for (int a = 0; a < 3; ++a) { mix(b0[a]); mix(b1[a]); }
        mix((uint64_t)B::x | ((uint64_t)B::y << 16) | ((uint64_t)B::z << 32));
        mix(reinterpret_cast<uint64_t>(p.grid)); mix(reinterpret_cast<uint64_t>(p.texels));
        sig |= 1ull;                                                    // never 0 (= no valid memset)
    }
    if (partial && !(p.clearSig && sig && *p.clearSig == sig)) {
        hipError_t e = hipMemsetAsync(p.grid, 0, (size_t)p.N * p.N * p.nz, s);
        if (e != hipSuccess) return e;
        if (p.texels && (e = hipMemsetAsync(p.texels, 0, (size_t)p.N * p.N * p.nz * 4, s)) != hipSuccess) return e;
    }
    if (p.clearSig) *p.clearSig = sig;                                  // (0 after a full launch or an all-empty grid)
    if (!live) return hipSuccess;
    const uint32_t nbx = b1[0] - b0[0], nby = b1[1] - b0[1], nbz = b1[2] - b0[2];
    p.nbx = nbx; p.nby = nby; p.nbz = nbz;
    p.bx0 = b0[0]; p.by0 = b0[1]; p.bz0 = b0[2];
    uint32_t m = 0;
    while (m < 10 && p.morton && !((nbx >> m) & 1u) && !((nby >> m) & 1u) && !((nbz >> m) & 1u)) ++m;
    p.mortonBits = m;
    p.superX = nbx >> m;
    p.superY = nby >> m;
    const uint64_t nb = (uint64_t)nbx * nby * nbz;
    uint32_t rb = p.regionBits;
    while (rb > 0 && (8ull << rb) > nb) --rb;          // small grids: keep all XCDs busy
    p.regionBits = rb;
    const uint64_t span = 8ull << rb;                  // bricks per round of 8 regions
    const uint64_t grid = (nb + span - 1) / span * span;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 g((uint32_t)grid), b(B::threads);
    if (p.mode == 0 && p.lists) {
        // direction-space lists: no stack, the column is the queue of selected triangles (8 entries; 16 for deep scenes,
        // where a ray meets many candidates before its first flush)
        if constexpr (STACK == 8 || STACK == 16) {
#if defined(DXV_ABLATE)                                                  // (only in the library tools/ablate.py builds for itself: libdxv_ablate.so)
            if constexpr (B::threads == 64 && B::x == 4 && STACK == 16) {    // timing-only ablations of the default shape (tools/ablate.py)
                switch (p.ablate) {
                case 0: break;
                case 1: k_voxelize<B, 16, 0, false, 4, 1><<<g, b, 0, s>>>(p); return hipGetLastError();
                case 2: k_voxelize<B, 16, 0, false, 4, 2><<<g, b, 0, s>>>(p); return hipGetLastError();
                case 4: k_voxelize<B, 16, 0, false, 4, 4><<<g, b, 0, s>>>(p); return hipGetLastError();
                case 6: k_voxelize<B, 16, 0, false, 4, 6><<<g, b, 0, s>>>(p); return hipGetLastError();
                case 8: k_voxelize<B, 16, 0, false, 4, 8><<<g, b, 0, s>>>(p); return hipGetLastError();
                case 16: k_voxelize<B, 16, 0, false, 4, 16><<<g, b, 0, s>>>(p); return hipGetLastError();
                case 18: k_voxelize<B, 16, 0, false, 4, 18><<<g, b, 0, s>>>(p); return hipGetLastError();
                case 32: k_voxelize<B, 16, 0, false, 4, 32><<<g, b, 0, s>>>(p); return hipGetLastError();
                case 64: k_voxelize<B, 16, 0, false, 4, 64><<<g, b, 0, s>>>(p); return hipGetLastError();
                default: return hipErrorInvalidValue;
                }
            }
#endif
            if (p.texels) k_voxelize<B, STACK, 0, true, 4><<<g, b, 0, s>>>(p);
            else k_voxelize<B, STACK, 0, false, 4><<<g, b, 0, s>>>(p);
        } else return hipErrorInvalidValue;
    } else if (p.mode == 0) {
        if (p.texels) {
            if (p.wide) k_voxelize<B, STACK, 0, true, 2><<<g, b, 0, s>>>(p);
            else k_voxelize<B, STACK, 0, true, 1><<<g, b, 0, s>>>(p);
        } else if (p.wide == 2) k_voxelize<B, STACK, 0, false, 3><<<g, b, 0, s>>>(p);
        else if (p.wide) k_voxelize<B, STACK, 0, false, 2><<<g, b, 0, s>>>(p);
        else if (p.queued) k_voxelize<B, STACK, 0, false, 1><<<g, b, 0, s>>>(p);
        else k_voxelize<B, STACK, 0, false, 0><<<g, b, 0, s>>>(p);
    } else {
        if (p.queued) k_voxelize<B, STACK, 1, false, 1><<<g, b, 0, s>>>(p);
        else k_voxelize<B, STACK, 1, false, 0><<<g, b, 0, s>>>(p);
    }
    return hipGetLastError();
}

int stack_round_up(int want);
// Column depths a brick shape is compiled for: all eight for the shipped shape (4 x 4 x 4: the adaptive column and its
// tuning), three for the shapes that exist for sweeps and cross-checks (16: the lists' queue, 20: the walks' default,
// 64: always sufficient).  A depth in between takes the next one up -- a deeper column than asked is never wrong.
int stack_for_brick(int brickShape, int want)
{
    want = stack_round_up(want);
    if (brickShape == 4) return want;
    return want <= 16 ? 16 : want <= 20 ? 20 : 64;
}

template <class B>
static hipError_t launch_stack(const VoxelizeParams& p, int stackEntries, hipStream_t s)
{
    if constexpr (B::x == 4 && B::y == 4 && B::z == 4) {
        switch (stackEntries) {
        case 8: return launch_shape<B, 8>(p, s);
        case 12: return launch_shape<B, 12>(p, s);
        case 16: return launch_shape<B, 16>(p, s);
        case 20: return launch_shape<B, 20>(p, s);
        case 24: return launch_shape<B, 24>(p, s);
        case 32: return launch_shape<B, 32>(p, s);
        case 48: return launch_shape<B, 48>(p, s);
        default: return launch_shape<B, 64>(p, s);
        }
    } else {
        switch (stackEntries) {
        case 16: return launch_shape<B, 16>(p, s);
        case 20: return launch_shape<B, 20>(p, s);
        case 64: return launch_shape<B, 64>(p, s);
        default: return hipErrorInvalidValue;              // (stack_for_brick maps every depth to one of the three)
        }
    }
}

// The stack holds internal nodes only, one entry per level at most: treeHeight entries always
// suffice.  Smallest instantiated depth >= want (LDS = depth * 4 B per thread).
int stack_round_up(int want)
{
    const int sizes[] = {8, 12, 16, 20, 24, 32, 48, 64};
    for (int v : sizes) if (want <= v) return v;
    return 64;
}

hipError_t launch_voxelize(const VoxelizeParams& p, int brickShape, int stackEntries, hipStream_t s)
{
    if (stack_round_up(stackEntries) != stackEntries || stack_for_brick(brickShape, stackEntries) != stackEntries) return hipErrorInvalidValue;
    switch (brickShape) {
    case 0: return launch_stack<Brick0>(p, stackEntries, s);
    case 1: return launch_stack<Brick1>(p, stackEntries, s);
    case 2: return launch_stack<Brick2>(p, stackEntries, s);
    case 3: return launch_stack<Brick3>(p, stackEntries, s);
    case 4: return launch_stack<Brick4>(p, stackEntries, s);
    case 5: return launch_stack<Brick5>(p, stackEntries, s);
    case 6: return launch_stack<Brick6>(p, stackEntries, s);
    case 7: return launch_stack<Brick7>(p, stackEntries, s);
    default: return hipErrorInvalidValue;
    }
}

// Wrapping 64-bit sum of the 8-byte words of a device buffer (dxv_scene_checksum: what arrived after a broadcast is what was sent)
__global__ __launch_bounds__(256) void k_checksum(const unsigned long long* __restrict__ words, size_t n, unsigned long long* out)
{
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c += words[i];
    for (int off = 32; off; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, c);
}
hipError_t launch_checksum(const void* buf, size_t bytes, unsigned long long* out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    const size_t n = bytes / 8;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks) k_checksum<<<(uint32_t)blocks, 256, 0, s>>>(static_cast<const unsigned long long*>(buf), n, out);
    return hipGetLastError();
}

// Solid-voxel count: 16 B per lane streaming reduction, one atomic per workgroup.
__global__ __launch_bounds__(256) void k_count(const uint8_t* __restrict__ grid, size_t n, unsigned long long* out)
{
    __shared__ unsigned long long part[4];
    const size_t n16 = n / 16;
    const uint4* g16 = reinterpret_cast<const uint4*>(grid);
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = g16[i];
        c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); // bytes are 0 or 1
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 15)) c += grid[n16 * 16 + threadIdx.x];
    for (int off = 32; off; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

hipError_t launch_count(const uint8_t* grid, size_t n, unsigned long long* out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    size_t blocks = (n / 16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    k_count<<<(uint32_t)blocks, 256, 0, s>>>(grid, n, out);
    return hipGetLastError();
}

// Bit-packed copy of the occupancy bytes for the host: output byte j holds voxels 8j .. 8j+7,
// voxel 8j+i in bit i.  One lane reads 16 grid bytes and writes 2; HBM bound (9/8 B per voxel).
__global__ __launch_bounds__(256) void k_pack_bits(const uint8_t* __restrict__ grid, size_t n, uint8_t* __restrict__ packed)
{
    constexpr unsigned long long kGather = 0x0102040810204080ull;   // byte i (0 or 1) -> bit 56 + i of the product
    const size_t n16 = n / 16;
    const uint4* g16 = reinterpret_cast<const uint4*>(grid);
    uint16_t* p16 = reinterpret_cast<uint16_t*>(packed);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = g16[i];
        const unsigned long long lo = ((unsigned long long)v.y << 32) | v.x, hi = ((unsigned long long)v.w << 32) | v.z;
        p16[i] = (uint16_t)(((lo * kGather) >> 56) | (((hi * kGather) >> 56) << 8));
    }
    if (blockIdx.x == 0 && threadIdx.x < 2) {                       // the last n % 16 voxels: at most two bytes
        const size_t first = n16 * 16 + (size_t)threadIdx.x * 8;
        if (first < n) {
            uint32_t b = 0;
            for (size_t k = 0; k < 8 && first + k < n; ++k) b |= (uint32_t)(grid[first + k] & 1u) << k;
            packed[first / 8] = (uint8_t)b;
        }
    }
}

hipError_t launch_pack_bits(const uint8_t* grid, size_t n, uint8_t* packed, hipStream_t s)
{
    size_t blocks = (n / 16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    k_pack_bits<<<(uint32_t)blocks, 256, 0, s>>>(grid, n, packed);
    return hipGetLastError();
}

} // namespace dxv
