// raycast.hip -- the display pass that consumes the grid: Voxelizer::renderRayCast
// (Content/Voxelizer.cpp:371-399: Draw(3) of VSScreenQuad + PSRayCast) as one HIP kernel, one
// thread per pixel, 16x16-pixel tiles so neighbouring rays share grid cache lines.
#include "dxv_device.h"
#include "dxv_raycast.h"

namespace dxv {

__global__ __launch_bounds__(256) void k_raycast(RayCastCB cb, const uint8_t* __restrict__ grid, uint32_t N,
                                                 uint32_t width, uint32_t height, uint32_t* __restrict__ rgba8)
{
    const uint32_t px = blockIdx.x * 16u + (threadIdx.x & 15u), py = blockIdx.y * 16u + (threadIdx.x >> 4);
    if (px >= width || py >= height) return;
    float c[4];
    raycast_pixel(cb, grid, N, (float)px + 0.5f, (float)py + 0.5f, c);
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

hipError_t launch_raycast(const RayCastCB& cb, const uint8_t* grid, uint32_t N, uint32_t width, uint32_t height,
                          uint32_t* rgba8, hipStream_t s)
{
    const dim3 g((width + 15) / 16, (height + 15) / 16), b(256);
    k_raycast<<<g, b, 0, s>>>(cb, grid, N, width, height, rgba8);
    return hipGetLastError();
}

} // namespace dxv
