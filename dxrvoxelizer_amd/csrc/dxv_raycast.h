// dxv_raycast.h -- the grid's consumer: the volume ray-cast display pass of the reference
// (Content/Shaders/PSRayCast.hlsl:1-187, constants built by Voxelizer::UpdateFrame,
// Content/Voxelizer.cpp:81-106, full-screen triangle Shaders/VSScreenQuad.hlsl).  SURVEY 8(f) N3.
//
// Float32 with a fixed operation order (min16float is a precision hint in HLSL; the canonical form
// is float), trilinear CLAMP sampling of the alpha channel done in float (hardware samplers use
// fixed-point weights, so the reference image itself is only reproducible to about 1/256).
// __host__ __device__ like the rest of the arithmetic so tests can run it on the CPU.
#pragma once
#include <stddef.h>
#include "dxv_math.h"

namespace dxv {

struct RayCastCB {
    float lightPt[3];         // g_localSpaceLightPt
    float eyePt[3];           // g_localSpaceEyePt
    float screenToLocal[16];  // row-major, row-vector convention: (x, y, z, 1) * M   (PSRayCast.hlsl:60-65)
};

constexpr int kNumSamples = 128;        // PSRayCast.hlsl:7
constexpr int kNumLightSamples = 32;    // :8
constexpr float kAbsorption = 1.0f;     // :9
constexpr float kZeroThreshold = 0.01f; // :10

// Texture3D.SampleLevel(LINEAR_CLAMP, tex, 0).w of the R10G10B10A2 grid: alpha is 1 where the voxel
// is occupied, 0 elsewhere (Content/Shaders/DXRVoxelizer.hlsl:84).  Texel (ix,iy,iz) centre sits at
// (i + 0.5) / N.
// `empty` (optional): one byte per 8 x 8 x 8 brick of the grid, 1 where the voxels [8b, 8b+8] per axis
// -- the brick and the first plane of its +x/+y/+z neighbours -- are all 0 (raycast.hip k_brick_empty).
// The eight texels of a sample whose low corner lies in the brick are among those voxels, so a flagged
// brick means the interpolation below would return exactly 0.0f: returning it at once changes no bit
// of the image and spares the eight loads for the samples that march through empty space (most of them).
constexpr uint32_t kEmptyBrick = 8;

DXV_HD float sample_alpha(const uint8_t* grid, uint32_t N, float tx, float ty, float tz, const uint8_t* empty = nullptr)
{
    const float fn = (float)N;
    const float ux = tx * fn - 0.5f, uy = ty * fn - 0.5f, uz = tz * fn - 0.5f;
    const float fx0 = __builtin_floorf(ux), fy0 = __builtin_floorf(uy), fz0 = __builtin_floorf(uz);
    const float wx = ux - fx0, wy = uy - fy0, wz = uz - fz0;
    const int32_t n1 = (int32_t)N - 1;
    auto clampi = [n1](float f) { int32_t i = (int32_t)f; return i < 0 ? 0 : (i > n1 ? n1 : i); };
    const int32_t x0 = clampi(fx0), x1 = clampi(fx0 + 1.0f), y0 = clampi(fy0), y1 = clampi(fy0 + 1.0f);
    const int32_t z0 = clampi(fz0), z1 = clampi(fz0 + 1.0f);
    if (empty) {
        const uint32_t M = (N + kEmptyBrick - 1) / kEmptyBrick;
        if (empty[(((uint32_t)z0 / kEmptyBrick) * M + (uint32_t)y0 / kEmptyBrick) * M + (uint32_t)x0 / kEmptyBrick]) return 0.0f;
    }
    auto at = [grid, N](int32_t x, int32_t y, int32_t z) { return grid[((size_t)z * N + y) * N + x] ? 1.0f : 0.0f; };
    const float c00 = at(x0, y0, z0) + wx * (at(x1, y0, z0) - at(x0, y0, z0));
    const float c10 = at(x0, y1, z0) + wx * (at(x1, y1, z0) - at(x0, y1, z0));
    const float c01 = at(x0, y0, z1) + wx * (at(x1, y0, z1) - at(x0, y0, z1));
    const float c11 = at(x0, y1, z1) + wx * (at(x1, y1, z1) - at(x0, y1, z1));
    const float c0 = c00 + wy * (c10 - c00), c1 = c01 + wy * (c11 - c01);
    return c0 + wz * (c1 - c0);
}

DXV_HD float get_sample(const uint8_t* grid, uint32_t N, float px, float py, float pz, const uint8_t* empty = nullptr)
{
    // tex = float3(0.5, -0.5, 0.5) * pos + 0.5 (PSRayCast.hlsl:137), GetSample :104-113
    const float d = sample_alpha(grid, N, 0.5f * px + 0.5f, -0.5f * py + 0.5f, 0.5f * pz + 0.5f, empty);
    return min_(d * 8.0f, 16.0f);
}

DXV_HD bool outside_unit(float x, float y, float z) { return abs_(x) > 1.0f || abs_(y) > 1.0f || abs_(z) > 1.0f; }

// ComputeStartPoint, PSRayCast.hlsl:70-99
DXV_HD bool compute_start_point(float pos[3], const float dir[3])
{
    if (abs_(pos[0]) <= 1.0f && abs_(pos[1]) <= 1.0f && abs_(pos[2]) <= 1.0f) return true;
    float U = 3.402823466e+38f;
    bool isHit = false;
    for (int i = 0; i < 3; ++i) {
        const float sgn = dir[i] > 0.0f ? 1.0f : (dir[i] < 0.0f ? -1.0f : 0.0f);
        const float u = (-sgn - pos[i]) / dir[i];
        if (u < 0.0f) continue;
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        if (abs_(dir[j] * u + pos[j]) > 1.0f) continue;
        if (abs_(dir[k] * u + pos[k]) > 1.0f) continue;
        if (u < U) { U = u; isHit = true; }
    }
    for (int i = 0; i < 3; ++i) {
        const float v = dir[i] * U + pos[i];
        pos[i] = min_(max_(v, -1.0f), 1.0f);
    }
    return isHit;
}

// One pixel, sspos = (px + 0.5, py + 0.5) (SV_POSITION).  rgba in [0,1] (PSRayCast.hlsl:118-187).
DXV_HD void raycast_pixel(const RayCastCB& cb, const uint8_t* grid, uint32_t N, float sx, float sy, float rgba[4],
                          const uint8_t* empty = nullptr)
{
    const float clear[3] = {0.0f, 0.2f, 0.4f};                                  // SharedConst.h:8
    const float maxDist = 2.0f * __builtin_sqrtf(3.0f);
    const float stepScale = maxDist / (float)kNumSamples, lightStepScale = maxDist / (float)kNumLightSamples;
    const float* m = cb.screenToLocal;
    // ScreenToLocal(float3(sspos.xy, 0)): mul(float4(p, 1), M) then / w
    const float hx = (sx * m[0] + sy * m[4]) + m[12], hy = (sx * m[1] + sy * m[5]) + m[13];
    const float hz = (sx * m[2] + sy * m[6]) + m[14], hw = (sx * m[3] + sy * m[7]) + m[15];
    float pos[3] = {hx / hw, hy / hw, hz / hw};
    float dir[3] = {pos[0] - cb.eyePt[0], pos[1] - cb.eyePt[1], pos[2] - cb.eyePt[2]};
    const float dl = __builtin_sqrtf((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
    dir[0] /= dl; dir[1] /= dl; dir[2] /= dl;
    if (!compute_start_point(pos, dir)) { rgba[0] = clear[0]; rgba[1] = clear[1]; rgba[2] = clear[2]; rgba[3] = 0.0f; return; }
    const float step[3] = {dir[0] * stepScale, dir[1] * stepScale, dir[2] * stepScale};
    const float ll = __builtin_sqrtf((cb.lightPt[0] * cb.lightPt[0] + cb.lightPt[1] * cb.lightPt[1]) + cb.lightPt[2] * cb.lightPt[2]);
    const float lstep[3] = {cb.lightPt[0] / ll * lightStepScale, cb.lightPt[1] / ll * lightStepScale, cb.lightPt[2] / ll * lightStepScale};
    float transmit = 1.0f, scatter = 0.0f;
    for (int i = 0; i < kNumSamples; ++i) {
        if (outside_unit(pos[0], pos[1], pos[2])) break;
        const float density = get_sample(grid, N, pos[0], pos[1], pos[2], empty);
        if (density > kZeroThreshold) {
            const float scaledDens = density * stepScale;
            transmit *= min_(max_(1.0f - scaledDens * kAbsorption, 0.0f), 1.0f);
            if (transmit < kZeroThreshold) break;
            float lightTrans = 1.0f;
            float lp[3] = {pos[0] + lstep[0], pos[1] + lstep[1], pos[2] + lstep[2]};
            for (int j = 0; j < kNumLightSamples; ++j) {
                if (outside_unit(lp[0], lp[1], lp[2])) break;
                const float lightDens = get_sample(grid, N, lp[0], lp[1], lp[2], empty);
                lightTrans *= min_(max_(1.0f - kAbsorption * lightStepScale * lightDens, 0.0f), 1.0f);
                if (lightTrans < kZeroThreshold) break;
                lp[0] += lstep[0]; lp[1] += lstep[1]; lp[2] += lstep[2];
            }
            scatter += lightTrans * transmit * scaledDens;
        }
        pos[0] += step[0]; pos[1] += step[1]; pos[2] += step[2];
    }
    for (int c = 0; c < 3; ++c) {
        float r = scatter * 0.8f + 0.2f;
        r = r + transmit * (clear[c] * clear[c] - r);          // lerp(result, clear^2, transmit)
        rgba[c] = __builtin_sqrtf(r);
    }
    rgba[3] = 1.0f;
}

// ---- Voxelizer::UpdateFrame (Content/Voxelizer.cpp:81-106): row-major 4x4, row-vector convention ----
DXV_HD void mat_mul(const float a[16], const float b[16], float out[16])
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            out[4 * r + c] = ((a[4 * r] * b[c] + a[4 * r + 1] * b[4 + c]) + a[4 * r + 2] * b[8 + c]) + a[4 * r + 3] * b[12 + c];
}

inline bool mat_inverse(const float m[16], float out[16])
{
    double a[16], inv[16];
    for (int i = 0; i < 16; ++i) a[i] = m[i];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    if (det == 0.0) return false;
    for (int i = 0; i < 16; ++i) out[i] = (float)(inv[i] / det);
    return true;
}

inline void transform_coord(const float p[3], const float m[16], float out[3])
{
    const float x = ((p[0] * m[0] + p[1] * m[4]) + p[2] * m[8]) + m[12], y = ((p[0] * m[1] + p[1] * m[5]) + p[2] * m[9]) + m[13];
    const float z = ((p[0] * m[2] + p[1] * m[6]) + p[2] * m[10]) + m[14], w = ((p[0] * m[3] + p[1] * m[7]) + p[2] * m[11]) + m[15];
    out[0] = x / w; out[1] = y / w; out[2] = z / w;
}

// bound = {c.xyz, w}; posScale = {x, y, z, s}; eye[3]; viewProj row-major.  Returns false for a singular chain.
inline bool update_frame(const float bound[4], const float posScale[4], const float eye[3], const float viewProj[16],
                         float width, float height, RayCastCB& cb)
{
    auto scaling = [](float s, float m[16]) { for (int i = 0; i < 16; ++i) m[i] = 0; m[0] = m[5] = m[10] = s; m[15] = 1; };
    auto translation = [](float x, float y, float z, float m[16]) { for (int i = 0; i < 16; ++i) m[i] = 0; m[0] = m[5] = m[10] = m[15] = 1; m[12] = x; m[13] = y; m[14] = z; };
    float s1[16], t1[16], s2[16], t2[16], a[16], b[16], world[16], worldI[16], wvp[16], toScreen[16], l2s[16];
    scaling(bound[3], s1); translation(bound[0], bound[1], bound[2], t1);
    scaling(posScale[3], s2); translation(posScale[0], posScale[1], posScale[2], t2);
    mat_mul(s1, t1, a); mat_mul(a, s2, b); mat_mul(b, t2, world);                       // Voxelizer.cpp:84-87
    if (!mat_inverse(world, worldI)) return false;                                       // :88
    mat_mul(world, viewProj, wvp);                                                       // :89
    const float light[3] = {-10.0f, 45.0f, -75.0f};                                      // :93
    transform_coord(light, worldI, cb.lightPt);
    transform_coord(eye, worldI, cb.eyePt);                                              // :94
    for (int i = 0; i < 16; ++i) toScreen[i] = 0;                                        // :96-102
    toScreen[0] = 0.5f * width; toScreen[5] = -0.5f * height; toScreen[10] = 1.0f;
    toScreen[12] = 0.5f * width; toScreen[13] = 0.5f * height; toScreen[15] = 1.0f;
    mat_mul(wvp, toScreen, l2s);                                                         // :103
    return mat_inverse(l2s, cb.screenToLocal);                                           // :104 (stored transposed for HLSL)
}

} // namespace dxv
