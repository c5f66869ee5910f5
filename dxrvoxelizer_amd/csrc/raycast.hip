// raycast.hip -- the display pass that consumes the grid: Voxelizer::renderRayCast
// (Content/Voxelizer.cpp:371-399: Draw(3) of VSScreenQuad + PSRayCast) as one HIP kernel, one
// thread per pixel, 16x16-pixel tiles so neighbouring rays share grid cache lines.
#include "dxv_device.h"
#include "dxv_raycast.h"

namespace dxv {

// Empty-brick flags for the display pass (dxv_raycast.h sample_alpha), in two steps so that every voxel
// is read once and coalesced.  A sample whose low corner lies in brick b touches voxels [8b, 8b+8] per
// axis: brick b itself plus the first plane / edge / corner of its +x, +y, +z neighbours.
//   k_brick_summary: one byte per brick -- bit0 any voxel, bit1 any on the x=0 face, bit2 z=0 face,
//     bit3 x=0,z=0 edge, bit4 y=0 face, bit5 x=0,y=0 edge, bit6 y=0,z=0 edge, bit7 the corner voxel.
//     One wave per run of 8 bricks along x: lane = xb + 8*y reads 8 voxels as one 64-bit word, so 8
//     lanes cover one 64-byte line; the wave loops over the 8 z planes and OR-reduces over y.
//   k_brick_empty: empty[b] = none of the 8 summaries that reach into [8b, 8b+8]^3 has its bit set.
__device__ inline uint64_t load_row8(const uint8_t* __restrict__ grid, uint32_t N, uint32_t x, uint32_t y, uint32_t z)
{
    if (y >= N || z >= N || x >= N) return 0;
    const uint8_t* row = grid + ((size_t)z * N + y) * N + x;
    if ((N & 7u) == 0u) return *reinterpret_cast<const uint64_t*>(row);
    uint64_t w = 0;
    for (uint32_t i = 0; i < 8u && x + i < N; ++i) w |= (uint64_t)row[i] << (8u * i);
    return w;
}

__global__ __launch_bounds__(256) void k_brick_summary(const uint8_t* __restrict__ grid, uint32_t N, uint32_t M,
                                                       uint8_t* __restrict__ summary)
{
    const uint32_t lane = threadIdx.x & 63u, xb = lane & 7u, y = lane >> 3;
    const uint32_t runs = (M + 7u) / 8u;                                   // runs of 8 bricks along x
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t run = wave % runs, by = (wave / runs) % M, bz = wave / (runs * M);
    if (bz >= M) return;
    const uint32_t bx = run * 8u + xb;
    uint32_t bits = 0;
    for (uint32_t z = 0; z < 8u; ++z) {
        const uint64_t w = load_row8(grid, N, bx * 8u, by * 8u + y, bz * 8u + z);
        const uint32_t any = w != 0ull, face = (w & 0xFFull) != 0ull;
        bits |= any | (face << 1);
        if (z == 0u) bits |= (any << 2) | (face << 3);
    }
    if (y == 0u) bits |= bits << 4;
    bits |= (uint32_t)__shfl_xor((int)bits, 8);
    bits |= (uint32_t)__shfl_xor((int)bits, 16);
    bits |= (uint32_t)__shfl_xor((int)bits, 32);
    if (y == 0u && bx < M) summary[((size_t)bz * M + by) * M + bx] = (uint8_t)bits;
}

__global__ __launch_bounds__(256) void k_brick_empty(const uint8_t* __restrict__ summary, uint32_t M, uint8_t* __restrict__ empty)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= M * M * M) return;
    const uint32_t x = i % M, y = (i / M) % M, z = i / (M * M);
    auto at = [&](uint32_t dx, uint32_t dy, uint32_t dz) -> uint32_t {
        return (x + dx < M && y + dy < M && z + dz < M) ? summary[((size_t)(z + dz) * M + (y + dy)) * M + (x + dx)] : 0u;
    };
    const uint32_t touched = (at(0, 0, 0) & 0x01u) | (at(1, 0, 0) & 0x02u) | (at(0, 0, 1) & 0x04u) | (at(1, 0, 1) & 0x08u) |
                             (at(0, 1, 0) & 0x10u) | (at(1, 1, 0) & 0x20u) | (at(0, 1, 1) & 0x40u) | (at(1, 1, 1) & 0x80u);
    empty[i] = touched ? 0 : 1;
}

__global__ __launch_bounds__(256) void k_raycast(RayCastCB cb, const uint8_t* __restrict__ grid, uint32_t N,
                                                 uint32_t width, uint32_t height, uint32_t* __restrict__ rgba8,
                                                 const uint8_t* __restrict__ empty)
{
    const uint32_t px = blockIdx.x * 16u + (threadIdx.x & 15u), py = blockIdx.y * 16u + (threadIdx.x >> 4);
    if (px >= width || py >= height) return;
    float c[4];
    raycast_pixel(cb, grid, N, (float)px + 0.5f, (float)py + 0.5f, c, empty);
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = c[k];
        if (!(v > 0.0f)) v = 0.0f;
        if (v > 1.0f) v = 1.0f;
        out |= (uint32_t)(v * 255.0f + 0.5f) << (8 * k);          // R8G8B8A8_UNORM
    }
    rgba8[(size_t)py * width + px] = out;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// scratch for the flags and, behind them, the summaries they are made from
size_t empty_brick_bytes(uint32_t N)
{
    const size_t M = (N + kEmptyBrick - 1) / kEmptyBrick;
    return 2 * align_up(M * M * M, 256);
}

// empty: scratch of empty_brick_bytes(N) bytes for the empty-brick flags, or NULL to march without them
hipError_t launch_raycast(const RayCastCB& cb, const uint8_t* grid, uint32_t N, uint32_t width, uint32_t height,
                          uint32_t* rgba8, uint8_t* empty, hipStream_t s)
{
    if (empty) {
        const uint32_t M = (N + kEmptyBrick - 1) / kEmptyBrick, waves = ((M + 7u) / 8u) * M * M;
        uint8_t* summary = empty + align_up(empty_brick_bytes(N) / 2, 256);      // second half of the scratch
        k_brick_summary<<<(waves + 3u) / 4u, 256, 0, s>>>(grid, N, M, summary);
        k_brick_empty<<<(M * M * M + 255u) / 256u, 256, 0, s>>>(summary, M, empty);
    }
    const dim3 g((width + 15) / 16, (height + 15) / 16), b(256);
    k_raycast<<<g, b, 0, s>>>(cb, grid, N, width, height, rgba8, empty);
    return hipGetLastError();
}

}  // namespace dxv
