// traverse.hip -- the voxelisation kernels: one thread per voxel, one ray per thread.
//
// Replaces DispatchRays(GRID_SIZE, GRID_SIZE*GRID_SIZE, 1) with raygenMain / closestHitMain /
// missMain (Content/Voxelizer.cpp:366-368, Content/Shaders/DXRVoxelizer.hlsl:58-85, :132-148).
//
// Launch shape: a 256-thread workgroup owns a BX x BY x BZ brick of voxels, a wavefront a
// 64-voxel sub-brick of it (lanes with neighbouring origins and near-parallel radial rays walk
// the same nodes).  Workgroup ids are remapped so that each of the 8 XCDs works on one
// contiguous Z range of bricks and its private L2 keeps the matching part of the tree.
// The per-thread traversal stack is an LDS column (stack[entry][thread]: consecutive lanes hit
// consecutive banks); its depth is chosen from the tree height recorded by the build, and an
// overflow is reported through the status word, never ignored.
#include "dxv_device.h"
#include "dxv_trace.h"

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

template <class B, int STACK, int MODE, bool TEXELS, bool QUEUED>
__global__ __launch_bounds__(B::threads, 8) void k_voxelize(VoxelizeParams p)   // 8 waves/SIMD: <= 64 VGPRs
{
    __shared__ int32_t stack[STACK * B::threads];
    const uint32_t N = p.N;
    const uint32_t nbx = (N + B::x - 1) / B::x, nby = (N + B::y - 1) / B::y, nbz = (p.nz + B::z - 1) / B::z;
    const uint32_t nb = nbx * nby * nbz;
    // XCD-aware remap: workgroups b and b + 8 share an XCD.  Bricks are numbered along a Morton
    // curve (below); runs of 2^regionBits consecutive bricks (compact regions) are dealt round-robin
    // to the 8 XCDs: each XCD's L2 sees compact regions, and the regions of all XCDs are fine
    // grained enough to balance the very uneven per-region cost.
    const uint32_t rb = p.regionBits;
    const uint32_t j = blockIdx.x >> 3;
    const uint32_t lin = ((((j >> rb) << 3) | (blockIdx.x & 7u)) << rb) | (j & ((1u << rb) - 1u));
    if (lin >= nb) return;
    // brick order: Morton inside 2^m-brick super-blocks (m = p.mortonBits, the largest power of two
    // dividing all three brick counts), super-blocks linear.  Consecutive workgroups of an XCD then
    // cover a compact region and reuse the same part of the tree in L1/L2.
    const uint32_t m = p.mortonBits;
    const uint32_t low = lin & ((1u << (3u * m)) - 1u), high = lin >> (3u * m);
    const uint32_t sx = nbx >> m, sy = nby >> m;
    const uint32_t bx = ((high % sx) << m) | compact1by2(low);
    const uint32_t by = (((high / sx) % sy) << m) | compact1by2(low >> 1);
    const uint32_t bz = ((high / (sx * sy)) << m) | compact1by2(low >> 2);
    const uint32_t tid = threadIdx.x;
    const uint32_t ix = bx * B::x + tid % B::x;
    const uint32_t iy = by * B::y + (tid / B::x) % B::y;
    const uint32_t lz = bz * B::z + tid / (B::x * B::y);
    if (ix >= N || iy >= N || lz >= p.nz) return;
    const uint32_t iz = p.z0 + (lz / p.zBlock) * p.zPeriod + lz % p.zBlock;
    const size_t id = ((size_t)lz * N + iy) * N + ix;

    const StridedStack stk{stack + tid, B::threads};
    bool overflow = false;
    uint8_t occ;
    if (MODE == 0) {
        uint32_t texel = 0;
        occ = voxel_reference<QUEUED>(p.scene, N, ix, iy, iz, stk, STACK, TEXELS ? &texel : nullptr, overflow);
        if (TEXELS) p.texels[id] = texel;
    } else {
        occ = voxel_parity<QUEUED>(p.scene, N, ix, iy, iz, stk, STACK, overflow);
    }
    if (overflow) atomicOr(p.status, 1u);
    p.grid[id] = occ;
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

template <class B, int STACK>
static hipError_t launch_shape(const VoxelizeParams& pin, hipStream_t s)
{
    VoxelizeParams p = pin;
    const uint32_t nbx = (p.N + B::x - 1) / B::x, nby = (p.N + B::y - 1) / B::y, nbz = (p.nz + B::z - 1) / B::z;
    uint32_t m = 0;
    while (m < 10 && p.morton && !((nbx >> m) & 1u) && !((nby >> m) & 1u) && !((nbz >> m) & 1u)) ++m;
    p.mortonBits = m;
    const uint64_t nb = (uint64_t)nbx * nby * nbz;
    uint32_t rb = p.regionBits;
    while (rb > 0 && (8ull << rb) > nb) --rb;          // small grids: keep all XCDs busy
    p.regionBits = rb;
    const uint64_t span = 8ull << rb;                  // bricks per round of 8 regions
    const uint64_t grid = (nb + span - 1) / span * span;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 g((uint32_t)grid), b(B::threads);
    if (p.mode == 0) {
        if (p.texels) k_voxelize<B, STACK, 0, true, true><<<g, b, 0, s>>>(p);
        else if (p.queued) k_voxelize<B, STACK, 0, false, true><<<g, b, 0, s>>>(p);
        else k_voxelize<B, STACK, 0, false, false><<<g, b, 0, s>>>(p);
    } else {
        if (p.queued) k_voxelize<B, STACK, 1, false, true><<<g, b, 0, s>>>(p);
        else k_voxelize<B, STACK, 1, false, false><<<g, b, 0, s>>>(p);
    }
    return hipGetLastError();
}

template <class B>
static hipError_t launch_stack(const VoxelizeParams& p, int stackEntries, hipStream_t s)
{
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
    if (stack_round_up(stackEntries) != stackEntries) return hipErrorInvalidValue;
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

} // namespace dxv
