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
    static constexpr int x = BX, y = BY, z = BZ;
    static_assert(BX * BY * BZ == 256, "one voxel per thread of a 256-thread workgroup");
};

template <class B, int STACK, int MODE, bool TEXELS>
__global__ __launch_bounds__(256) void k_voxelize(VoxelizeParams p)
{
    __shared__ int32_t stack[STACK * 256];
    const uint32_t N = p.N;
    const uint32_t nbx = (N + B::x - 1) / B::x, nby = (N + B::y - 1) / B::y, nbz = (p.nz + B::z - 1) / B::z;
    const uint32_t nb = nbx * nby * nbz;
    // XCD-aware remap: workgroups b and b + 8 share an XCD; give each XCD a contiguous chunk
    const uint32_t chunk = (nb + 7u) / 8u;
    const uint32_t lin = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
    if (lin >= nb) return;
    const uint32_t bx = lin % nbx, by = (lin / nbx) % nby, bz = lin / (nbx * nby);
    const uint32_t tid = threadIdx.x;
    const uint32_t ix = bx * B::x + tid % B::x;
    const uint32_t iy = by * B::y + (tid / B::x) % B::y;
    const uint32_t lz = bz * B::z + tid / (B::x * B::y);
    if (ix >= N || iy >= N || lz >= p.nz) return;
    const uint32_t iz = p.z0 + lz;
    const size_t id = ((size_t)lz * N + iy) * N + ix;

    StridedStack stk{stack + tid, 256};
    uint8_t occ = 0;
    if (MODE == 0) {
        const Ray r = make_ray_reference(N, ix, iy, iz);
        Hit best;
        if (!trace_reference(r, p.nodes, p.triPos, stk, STACK, best)) atomicOr(p.status, 1u);
        uint32_t texel = 0;
        if (best.k != 0xffffffffu) {                     // else missMain: nothing written
            const TriNrm tn = p.triNrm[best.leaf];
            float nx, ny, nz;
            occ = predicate(r, tn.n0, tn.n1, tn.n2, best.b1, best.b2, nx, ny, nz) ? 1 : 0;
            if (TEXELS && occ) texel = pack_texel(nx, ny, nz);
        }
        if (TEXELS) p.texels[id] = texel;
    } else {
        const Ray r = make_ray_parity(N, ix, iy, iz);
        uint32_t count;
        if (!trace_parity(r, p.nodes, p.triPos, stk, STACK, count)) atomicOr(p.status, 1u);
        occ = (uint8_t)(count & 1u);
    }
    p.grid[id] = occ;
}

using BrickRow = Brick<64, 4, 1>;    // wave = 64x1x1 row
using BrickTile = Brick<8, 8, 4>;    // wave = 8x8x1 tile
using BrickCube = Brick<4, 4, 16>;   // wave = 4x4x4 cube
using BrickSlab = Brick<16, 4, 4>;   // wave = 16x4x1

int num_brick_shapes() { return 4; }

template <class B, int STACK>
static hipError_t launch_shape(const VoxelizeParams& p, hipStream_t s)
{
    const uint32_t nbx = (p.N + B::x - 1) / B::x, nby = (p.N + B::y - 1) / B::y, nbz = (p.nz + B::z - 1) / B::z;
    const uint64_t nb = (uint64_t)nbx * nby * nbz;
    const uint64_t chunk = (nb + 7) / 8;
    const uint64_t grid = chunk * 8;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    const dim3 g((uint32_t)grid), b(256);
    if (p.mode == 0) {
        if (p.texels) k_voxelize<B, STACK, 0, true><<<g, b, 0, s>>>(p);
        else k_voxelize<B, STACK, 0, false><<<g, b, 0, s>>>(p);
    } else {
        k_voxelize<B, STACK, 1, false><<<g, b, 0, s>>>(p);
    }
    return hipGetLastError();
}

template <class B>
static hipError_t launch_stack(const VoxelizeParams& p, int stackEntries, hipStream_t s)
{
    switch (stackEntries) {
    case 16: return launch_shape<B, 16>(p, s);
    case 24: return launch_shape<B, 24>(p, s);
    case 32: return launch_shape<B, 32>(p, s);
    case 48: return launch_shape<B, 48>(p, s);
    default: return launch_shape<B, 64>(p, s);
    }
}

// The stack holds internal nodes only and one entry per level at most, so treeHeight - 1 entries
// always suffice; pick the smallest instantiated depth that covers it (LDS = depth * 1 KiB).
static int stack_for_height(uint32_t h)
{
    const int need = (int)h;
    if (need <= 16) return 16;
    if (need <= 24) return 24;
    if (need <= 32) return 32;
    if (need <= 48) return 48;
    return 64;
}

hipError_t launch_voxelize(const VoxelizeParams& p, int brickShape, int forceStack, hipStream_t s, uint32_t* stackUsed)
{
    const int st = forceStack > 0 ? forceStack : stack_for_height(p.treeHeight);
    if (st != 16 && st != 24 && st != 32 && st != 48 && st != 64) return hipErrorInvalidValue;
    if (stackUsed) *stackUsed = (uint32_t)st;
    switch (brickShape) {
    case 0: return launch_stack<BrickRow>(p, st, s);
    case 1: return launch_stack<BrickTile>(p, st, s);
    case 2: return launch_stack<BrickCube>(p, st, s);
    case 3: return launch_stack<BrickSlab>(p, st, s);
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
