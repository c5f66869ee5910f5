// dxv_math.h -- the arithmetic of the hot path: ray generation, slab test, watertight
// ray/triangle test, occupancy predicate, Morton keys and the Karras hierarchy rule.
//
// Everything is float32 with a FIXED operation order and no compiler contraction (the only
// fused operations are the explicit fmaf calls); the translation units that include this file are
// built with -ffp-contract=off.  Results therefore do not depend on BVH topology or traversal
// order and are reproducible bit for bit on any IEEE-754 machine.
//
// Functions are __host__ __device__ so that tests can drive the very same code on the CPU
// (tests/hostcheck); the shipped library only instantiates them in device kernels.
//
// Reference citations are relative to /root/reference/DXRVoxelizer/.
#pragma once
#include "dxv_types.h"

#pragma clang fp contract(off)

namespace dxv {

DXV_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DXV_HD float min_(float a, float b) { return __builtin_fminf(a, b); }
DXV_HD float max_(float a, float b) { return __builtin_fmaxf(a, b); }
DXV_HD float abs_(float a) { return __builtin_fabsf(a); }
DXV_HD float sel3(float x, float y, float z, int k) { return k == 0 ? x : (k == 1 ? y : z); }

// ------------------------------------------------------------------------------------------
// a / b, correctly rounded, for operands the RAY SET-UP divides: voxel-centre coordinates, their length, direction components
// (|a|, |b| in [2^-13, 2^12], |a / b| in [2^-13, 2^12]).  The compiler's quotient is the IEEE sequence  div_scale x 2, rcp, two
// Newton steps on the reciprocal, the product, two residual corrections, div_fmas, div_fixup  -- eleven instructions, of which
// div_scale and div_fixup only act on operands near the ends of the exponent range or on specials, and div_fmas is a plain fma when
// nothing was scaled.  In that range the quotient is therefore the eight instructions below, bit for bit, and the first three depend on the
// DENOMINATOR alone: the three components of o / |o|, the two texel coordinates, the two shear constants share them.  The host (oracle-side
// checks, tests/hostcheck) divides with `/`: equality of the two on EVERY voxel origin of every even grid up to 2048^3 is checked on the
// device (dxv_debug_division_check, tests/test_gpu_parity.py).  Divisions whose operands depend on the mesh (t, barycentrics, the normal)
// stay `/`.
// ------------------------------------------------------------------------------------------
struct RcpRefined { float b, r; };                            // a denominator and its reciprocal after two Newton steps
DXV_HD RcpRefined rcp_refined(float b)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DXV_IEEE_SETUP_DIVISIONS)
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r0, 1.0f);
    return RcpRefined{b, __builtin_fmaf(e0, r0, r0)};
#else
    return RcpRefined{b, 0.0f};
#endif
}
DXV_HD float div_by(float a, const RcpRefined& d)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DXV_IEEE_SETUP_DIVISIONS)
    const float q0 = a * d.r;
    const float e1 = __builtin_fmaf(-d.b, q0, a);
    const float q1 = __builtin_fmaf(e1, d.r, q0);
    const float e2 = __builtin_fmaf(-d.b, q1, a);
    return __builtin_fmaf(e2, d.r, q1);
#else
    return a / d.b;
#endif
}

// sqrtf(x), correctly rounded, for x = |o|^2 of a voxel origin (in [2^-21, 4]): the compiler's sequence scales tiny operands up and back
// and passes 0 / inf through -- sixteen instructions; in range it is v_sqrt_f32 (1 ulp) and the choice between it and its two neighbours
// by the sign of their residuals, nine instructions, the same bits (dxv_debug_division_check compares this word too).
DXV_HD float sqrt_in_range(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DXV_IEEE_SETUP_DIVISIONS)
    float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) - 1u), su = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) + 1u);
    const float ed = __builtin_fmaf(-sd, s, x), eu = __builtin_fmaf(-su, s, x);
    s = ed <= 0.0f ? sd : s;
    s = eu > 0.0f ? su : s;
    return s;
#else
    return __builtin_sqrtf(x);
#endif
}

// ------------------------------------------------------------------------------------------
// Ray generation: Content/Shaders/DXRVoxelizer.hlsl:44-53 (generateRay), :64-67 (un-flatten).
// ------------------------------------------------------------------------------------------
struct Ray {
    float ox, oy, oz;     // origin = voxel centre in [-1,1]^3, y flipped (hlsl:46,49)
    float dx, dy, dz;     // direction
    float ivx, ivy, ivz;  // 1/d
    float nox, noy, noz;  // -(o * 1/d)
    float Sx, Sy, Sz;     // watertight shear
    int kx, ky, kz;       // watertight axis permutation
};

DXV_HD void ray_origin(uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz, float& ox, float& oy, float& oz)
{
    const float fn = (float)N;
    if ((N & (N - 1u)) == 0u) {
        // power-of-two grid: x / N == x * (1/N) exactly (both are exact scalings), no divisions
        const float rn = div_by(1.0f, rcp_refined(fn));
        ox = ((float)ix + 0.5f) * rn * 2.0f - 1.0f;
        oy = -(((float)iy + 0.5f) * rn * 2.0f - 1.0f);
        oz = ((float)iz + 0.5f) * rn * 2.0f - 1.0f;
        return;
    }
    const RcpRefined byN = rcp_refined(fn);
    ox = div_by((float)ix + 0.5f, byN) * 2.0f - 1.0f;          // hlsl:46
    oy = -(div_by((float)iy + 0.5f, byN) * 2.0f - 1.0f);       // hlsl:49
    oz = div_by((float)iz + 0.5f, byN) * 2.0f - 1.0f;
}

// A radial ray whose origin lies beyond the scene's root box on the side it is moving to can not
// enter any box inside the root box: on that axis sign(d) == sign(o), so the exit distance
// fma(hi, 1/d, -(o/d)) is negative for every box (margin 1e-5 >> the 2^-24 relative rounding of
// o/d) and slab() fails for both children of the root.  Exact shortcut, not an approximation.
DXV_HD bool axis_leaves_root(float o, float lo, float hi)
{
    const float m = 1e-5f;
    return (o > hi + m && o > 0.0f) || (o < lo - m && o < 0.0f);
}

DXV_HD bool origin_leaves_root(float ox, float oy, float oz, const float* rootLo, const float* rootHi)
{
    return axis_leaves_root(ox, rootLo[0], rootHi[0]) || axis_leaves_root(oy, rootLo[1], rootHi[1]) ||
           axis_leaves_root(oz, rootLo[2], rootHi[2]);
}

DXV_HD void ray_shear(Ray& r)
{
    int kz = 0;
    float m = abs_(r.dx);
    if (abs_(r.dy) > m) { kz = 1; m = abs_(r.dy); }
    if (abs_(r.dz) > m) { kz = 2; }
    int kx = kz == 2 ? 0 : kz + 1;
    int ky = kx == 2 ? 0 : kx + 1;
    const float dkz = sel3(r.dx, r.dy, r.dz, kz);
    if (dkz < 0.0f) { const int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    const RcpRefined byDkz = rcp_refined(dkz);
    r.Sx = div_by(sel3(r.dx, r.dy, r.dz, kx), byDkz);
    r.Sy = div_by(sel3(r.dx, r.dy, r.dz, ky), byDkz);
    r.Sz = div_by(1.0f, byDkz);
}
// the same for a ray whose 1 / d is in place (finish_ray_reference): 1 / d[kz] is one of its three words -- the same operation on the
// same operand, bit for bit -- so the third division is a select
DXV_HD void ray_shear_finished(Ray& r)
{
    int kz = 0;
    float m = abs_(r.dx);
    if (abs_(r.dy) > m) { kz = 1; m = abs_(r.dy); }
    if (abs_(r.dz) > m) { kz = 2; }
    int kx = kz == 2 ? 0 : kz + 1;
    int ky = kx == 2 ? 0 : kx + 1;
    const float dkz = sel3(r.dx, r.dy, r.dz, kz);
    if (dkz < 0.0f) { const int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    const RcpRefined byDkz = rcp_refined(dkz);
    r.Sx = div_by(sel3(r.dx, r.dy, r.dz, kx), byDkz);
    r.Sy = div_by(sel3(r.dx, r.dy, r.dz, ky), byDkz);
    r.Sz = sel3(r.ivx, r.ivy, r.ivz, kz);
}

// Reference mode: direction = normalize(pos) (hlsl:52); canonical form p / sqrtf((xx+yy)+zz).
// len: |o| = sqrtf((ox ox + oy oy) + oz oz) when the caller holds it already (the lists' first step computes it as the ray's start radius,
// dm_ray_point: the same expression), else negative
DXV_HD void finish_ray_reference(Ray& r, float lenKnown = -1.0f)
{
    const float len = lenKnown >= 0.0f ? lenKnown : sqrt_in_range((r.ox * r.ox + r.oy * r.oy) + r.oz * r.oz);
    const RcpRefined byLen = rcp_refined(len);
    r.dx = div_by(r.ox, byLen); r.dy = div_by(r.oy, byLen); r.dz = div_by(r.oz, byLen);
    r.ivx = div_by(1.0f, rcp_refined(r.dx)); r.ivy = div_by(1.0f, rcp_refined(r.dy)); r.ivz = div_by(1.0f, rcp_refined(r.dz));
    r.nox = -(r.ox * r.ivx); r.noy = -(r.oy * r.ivy); r.noz = -(r.oz * r.ivz);
    r.kz = -1;   // shear constants are set up by the first triangle test (most rays never need them)
}

DXV_HD Ray make_ray_reference(uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz)
{
    Ray r;
    ray_origin(N, ix, iy, iz, r.ox, r.oy, r.oz);
    finish_ray_reference(r);
    return r;
}

// Parity mode: +X axis ray from the same origin.
DXV_HD Ray make_ray_parity(uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz)
{
    Ray r;
    ray_origin(N, ix, iy, iz, r.ox, r.oy, r.oz);
    r.dx = 1.0f; r.dy = 0.0f; r.dz = 0.0f;
    r.ivx = 1.0f; r.ivy = 0.0f; r.ivz = 0.0f;   // unused in parity mode
    r.nox = r.noy = r.noz = 0.0f;
    r.kz = 0; r.kx = 1; r.ky = 2;
    r.Sx = 0.0f; r.Sy = 0.0f; r.Sz = 1.0f;
    return r;
}

// ------------------------------------------------------------------------------------------
// Slab test.  t(plane) = fma(plane, 1/d, -(o/d)) is monotone in `plane`, so a box that contains
// another never fails where the inner one passes and never reports a later entry: the LBVH can
// not cull a triangle whose own box passes (BVH-topology independence).
// ------------------------------------------------------------------------------------------
DXV_HD bool slab(const Ray& r, float lox, float loy, float loz, float hix, float hiy, float hiz, float& tn)
{
    const float t0x = fma_(lox, r.ivx, r.nox), t1x = fma_(hix, r.ivx, r.nox);
    const float t0y = fma_(loy, r.ivy, r.noy), t1y = fma_(hiy, r.ivy, r.noy);
    const float t0z = fma_(loz, r.ivz, r.noz), t1z = fma_(hiz, r.ivz, r.noz);
    tn = max_(max_(min_(t0x, t1x), min_(t0y, t1y)), max_(min_(t0z, t1z), 0.0f));
    const float tf = min_(min_(max_(t0x, t1x), max_(t0y, t1y)), max_(t0z, t1z));
    return tn <= tf;
}

// The same test with the planes already sorted by the ray: `in*` is the plane of each axis the
// ray enters through (lo when that direction component is positive, hi when it is negative),
// `out*` the other one.  t(plane) is monotone in `plane` with the sign of 1/d, so
// min(t(lo), t(hi)) == t(in) and max(t(lo), t(hi)) == t(out) bit for bit: slab_sorted returns what
// slab returns (direction components are never zero: voxel centres do not lie on an axis plane).
DXV_HD bool slab_sorted(const Ray& r, float inx, float iny, float inz, float outx, float outy, float outz, float& tn)
{
    const float tix = fma_(inx, r.ivx, r.nox), tox = fma_(outx, r.ivx, r.nox);
    const float tiy = fma_(iny, r.ivy, r.noy), toy = fma_(outy, r.ivy, r.noy);
    const float tiz = fma_(inz, r.ivz, r.noz), toz = fma_(outz, r.ivz, r.noz);
    tn = max_(max_(tix, tiy), max_(tiz, 0.0f));
    const float tf = min_(min_(tox, toy), toz);
    return tn <= tf;
}

// +X axis ray against a box (parity mode): origin inside the box's y/z extent and box not behind.
DXV_HD bool slab_parity(const Ray& r, float loy, float loz, float hix, float hiy, float hiz)
{
    return loy <= r.oy && r.oy <= hiy && loz <= r.oz && r.oz <= hiz && hix >= r.ox;
}

// ------------------------------------------------------------------------------------------
// Watertight ray/triangle test (Woop, Benthin, Wald: "Watertight Ray/Triangle Intersection",
// JCGT 2013), both faces, strict 0 < t < TMax (DXR triangle rule; hlsl:76-77).  Edge functions
// are rounded products subtracted without fusion, so the value an edge gets from its two
// incident triangles is exactly antisymmetric; exact zeros are re-evaluated in double.
// Barycentrics as DXR reports them: b1 = weight of vertex 1, b2 = weight of vertex 2
// (hlsl:110-116).
// FILL (parity mode): an edge function that is exactly zero takes the sign it has at the
// symbolically perturbed origin o + (eps, eps^2), so a ray through a shared edge or vertex is
// counted for exactly one incident triangle.
// ------------------------------------------------------------------------------------------
// DEFER: b1, b2 come back as the undivided V, W and *detOut as det; the caller divides once, for the closest hit only
// (same operands, same correctly rounded divisions: the same b1, b2 bit for bit).
template <bool FILL, bool DEFER = false>
DXV_HD bool tri_test(const Ray& r, const F4& v0, const F4& v1, const F4& v2, float& t, float& b1, float& b2, float* detOut = nullptr)
{
    const float ax = v0.x - r.ox, ay = v0.y - r.oy, az = v0.z - r.oz;
    const float bx = v1.x - r.ox, by = v1.y - r.oy, bz = v1.z - r.oz;
    const float cx = v2.x - r.ox, cy = v2.y - r.oy, cz = v2.z - r.oz;
    const float akz = sel3(ax, ay, az, r.kz), bkz = sel3(bx, by, bz, r.kz), ckz = sel3(cx, cy, cz, r.kz);
    const float nSx = -r.Sx, nSy = -r.Sy;
    const float Ax = fma_(nSx, akz, sel3(ax, ay, az, r.kx)), Ay = fma_(nSy, akz, sel3(ax, ay, az, r.ky));
    const float Bx = fma_(nSx, bkz, sel3(bx, by, bz, r.kx)), By = fma_(nSy, bkz, sel3(bx, by, bz, r.ky));
    const float Cx = fma_(nSx, ckz, sel3(cx, cy, cz, r.kx)), Cy = fma_(nSy, ckz, sel3(cx, cy, cz, r.ky));
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    float su = U, sv = V, sw = W;
    if (FILL) {
        // U's edge runs C->B, V's A->C, W's B->A; perturbed sign = (ey != 0) ? -ey : ex
        if (su == 0.0f) { const float ey = By - Cy, ex = Bx - Cx; su = ey != 0.0f ? -ey : ex; }
        if (sv == 0.0f) { const float ey = Cy - Ay, ex = Cx - Ax; sv = ey != 0.0f ? -ey : ex; }
        if (sw == 0.0f) { const float ey = Ay - By, ex = Ax - Bx; sw = ey != 0.0f ? -ey : ex; }
        if (su == 0.0f || sv == 0.0f || sw == 0.0f) return false;
    }
    if ((su < 0.0f || sv < 0.0f || sw < 0.0f) && (su > 0.0f || sv > 0.0f || sw > 0.0f)) return false;
    const float det = (U + V) + W;
    if (det == 0.0f) return false;
    const float Az = r.Sz * akz, Bz = r.Sz * bkz, Cz = r.Sz * ckz;
    const float T = (U * Az + V * Bz) + W * Cz;
    t = T / det;
    if (!(t > 0.0f && t < kTMax)) return false;
    if (DEFER) { b1 = V; b2 = W; *detOut = det; }
    else { b1 = V / det; b2 = W / det; }
    return true;
}

// ------------------------------------------------------------------------------------------
// Parity mode, row form.  All voxels of a grid row (fixed iy, iz) fire the same +X line; for a
// triangle everything tri_test<true> computes from the y/z coordinates -- the sheared vertices
// (with Sx = Sy = 0 they are just v.yz - o.yz; the fma against -0 can only change the sign of a
// zero, which no comparison below observes), the edge functions U, V, W with their exact-zero
// fallback and fill rule, the sign test and det -- is the same for the whole row.  Only
// t = ((U*(v0.x-ox) + V*(v1.x-ox)) + W*(v2.x-ox)) / det and the box test hi.x >= ox depend on the
// voxel.  parity_row_setup is evaluated once per (row, triangle), parity_row_voxel per voxel:
// bit-identical to the per-voxel definition, a row shares one BVH walk.
// ------------------------------------------------------------------------------------------
struct ParityRowTri { float U, V, W, det, v0x, v1x, v2x, hix; bool hit; };
DXV_HD void tri_box(const F4& a, const F4& b, const F4& c, float lo[3], float hi[3]);

DXV_HD ParityRowTri parity_row_setup(float oy, float oz, const F4& v0, const F4& v1, const F4& v2)
{
    ParityRowTri s;
    float lo[3], hi[3];
    tri_box(v0, v1, v2, lo, hi);
    s.hit = lo[1] <= oy && oy <= hi[1] && lo[2] <= oz && oz <= hi[2];
    s.hix = hi[0]; s.v0x = v0.x; s.v1x = v1.x; s.v2x = v2.x;
    s.U = s.V = s.W = s.det = 0.0f;
    if (!s.hit) return s;       // the row misses the triangle's own box: nothing below can turn hit back on
    const float Ax = v0.y - oy, Ay = v0.z - oz, Bx = v1.y - oy, By = v1.z - oz, Cx = v2.y - oy, Cy = v2.z - oz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    float su = U, sv = V, sw = W;
    if (su == 0.0f) { const float ey = By - Cy, ex = Bx - Cx; su = ey != 0.0f ? -ey : ex; }
    if (sv == 0.0f) { const float ey = Cy - Ay, ex = Cx - Ax; sv = ey != 0.0f ? -ey : ex; }
    if (sw == 0.0f) { const float ey = Ay - By, ex = Ax - Bx; sw = ey != 0.0f ? -ey : ex; }
    if (su == 0.0f || sv == 0.0f || sw == 0.0f) s.hit = false;
    if ((su < 0.0f || sv < 0.0f || sw < 0.0f) && (su > 0.0f || sv > 0.0f || sw > 0.0f)) s.hit = false;
    s.U = U; s.V = V; s.W = W;
    s.det = (U + V) + W;
    if (s.det == 0.0f) s.hit = false;
    return s;
}

DXV_HD bool parity_row_voxel(const ParityRowTri& s, float ox)
{
    if (!(s.hix >= ox)) return false;
    const float Az = s.v0x - ox, Bz = s.v1x - ox, Cz = s.v2x - ox;      // Sz = 1
    const float T = (s.U * Az + s.V * Bz) + s.W * Cz;
    const float t = T / s.det;
    return t > 0.0f && t < kTMax;
}

// ------------------------------------------------------------------------------------------
// Closest-hit predicate: hlsl:110-116 (normal interpolation), :137-138 (test), :5 (threshold).
// ------------------------------------------------------------------------------------------
DXV_HD bool predicate(const Ray& r, const F4& n0, const F4& n1, const F4& n2, float b1, float b2,
                      float& nx, float& ny, float& nz)
{
    nx = (n0.x + b1 * (n1.x - n0.x)) + b2 * (n2.x - n0.x);
    ny = (n0.y + b1 * (n1.y - n0.y)) + b2 * (n2.y - n0.y);
    nz = (n0.z + b1 * (n1.z - n0.z)) + b2 * (n2.z - n0.z);
    const float l = __builtin_sqrtf((nx * nx + ny * ny) + nz * nz);
    nx = nx / l; ny = ny / l; nz = nz / l;
    return ((nx * r.dx + ny * r.dy) + nz * r.dz) > kThreshold;
}

// ------------------------------------------------------------------------------------------
// Per-triangle shortcut of the predicate (reference rule).  Every ray of the rule is radial, so its direction is the
// direction of its hit point: on one triangle the predicate is a function of the hit point alone,
//     angle(N(b), p(b)) < acos(0.12),   N(b), p(b) = the interpolated normal and position.
// N(b) is a non-negative combination of the vertex normals, p(b) of the vertices, so each stays inside the spherical cap
// around its generators' mean direction that contains the generators (a cap of less than 90 degrees is convex); with alpha
// the angle between the two cap axes and theta_n, theta_x the cap radii the predicate's angle lies in
// [alpha - theta_n - theta_x, alpha + theta_n + theta_x].  A triangle whose whole interval is on one side of the
// threshold by more than kClassMargin is classified once, at build time (double precision), and its hits need neither the
// 48-byte normal record nor the barycentrics; everything else -- the band of a surface where the angle is near 83 degrees,
// big or badly shaped triangles, vertices near the grid centre, zero or wildly unequal normals -- takes the canonical
// predicate above.  The margin (2e-3 rad, 2e-3 of the cosine) is three orders of magnitude above what the canonical
// float evaluation, a barycentric a few ulps outside [0, 1] or the watertight test's slack (the ray passes within 1e-5 of
// the triangle, the vertices are at least 1e-2 from the centre) can move the angle.  0 = undecided, 2 = in, 3 = out.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kClassShift = 28u, kClassIn = 2u, kClassOut = 3u;
DXV_HD bool class_cap(const double g[3][3], double maxRatio, double minLen, double axis[3], double& cosRadius)
{
    double u[3][3], lmin = 1e300, lmax = 0.0;
    for (int i = 0; i < 3; ++i) {
        const double l = __builtin_sqrt(g[i][0] * g[i][0] + g[i][1] * g[i][1] + g[i][2] * g[i][2]);
        if (!(l > minLen && l < 1e30)) return false;
        if (l < lmin) lmin = l;
        if (l > lmax) lmax = l;
        for (int a = 0; a < 3; ++a) u[i][a] = g[i][a] / l;
    }
    if (!(lmax <= maxRatio * lmin)) return false;
    double c[3] = {u[0][0] + u[1][0] + u[2][0], u[0][1] + u[1][1] + u[2][1], u[0][2] + u[1][2] + u[2][2]};
    const double cl = __builtin_sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    if (!(cl > 0.5)) return false;
    cosRadius = 1.0;
    for (int a = 0; a < 3; ++a) axis[a] = c[a] / cl;
    for (int i = 0; i < 3; ++i) {
        const double d = axis[0] * u[i][0] + axis[1] * u[i][1] + axis[2] * u[i][2];
        if (d < cosRadius) cosRadius = d;
    }
    return cosRadius > 0.5;                                        // caps of less than 60 degrees only
}
DXV_HD uint32_t normal_class(const F4& v0, const F4& v1, const F4& v2, const F4& n0, const F4& n1, const F4& n2)
{
    const double P[3][3] = {{v0.x, v0.y, v0.z}, {v1.x, v1.y, v1.z}, {v2.x, v2.y, v2.z}};
    const double N[3][3] = {{n0.x, n0.y, n0.z}, {n1.x, n1.y, n1.z}, {n2.x, n2.y, n2.z}};
    double an[3], ax[3], cn, cx;
    if (!class_cap(N, 4.0, 1e-30, an, cn) || !class_cap(P, 1e30, 1e-2, ax, cx)) return 0u;
    if (cn > 1.0) cn = 1.0;
    if (cx > 1.0) cx = 1.0;
    const double sn = __builtin_sqrt(1.0 - cn * cn), sx = __builtin_sqrt(1.0 - cx * cx);
    const double cb = cn * cx - sn * sx, sb = sn * cx + cn * sx;            // beta = theta_n + theta_x < 120 degrees
    const double ca = an[0] * ax[0] + an[1] * ax[1] + an[2] * ax[2];        // alpha
    // cos / sin of acos(0.12) -+ 2e-3
    constexpr double cosLo = 0.12198530645974026, sinLo = 0.9925319062921469;   // threshold - margin
    constexpr double cosHi = 0.1180142135404197, sinHi = 0.9930119059721471;   // threshold + margin
    if (cb > cosLo && ca > cosLo * cb + sinLo * sb) return kClassIn;        // alpha + beta < threshold - margin
    if (cb > -cosHi && ca < cosHi * cb - sinHi * sb) return kClassOut;      // alpha - beta > threshold + margin (and < 180 degrees)
    return 0u;
}

// float4(Normal, 1) stored to R10G10B10A2_UNORM (hlsl:84, Content/Voxelizer.cpp:65):
// D3D float->UNORM = clamp to [0,1] (NaN -> 0), scale, round to nearest.
DXV_HD uint32_t unorm10(float v)
{
    if (!(v > 0.0f)) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return (uint32_t)(v * 1023.0f + 0.5f);
}
DXV_HD uint32_t pack_texel(float nx, float ny, float nz)
{
    return unorm10(nx) | (unorm10(ny) << 10) | (unorm10(nz) << 20) | (3u << 30);
}

// ------------------------------------------------------------------------------------------
// Triangle preparation: normalising transform (Content/Voxelizer.cpp:52-57, :304-306) and the
// canonical padded box.
// ------------------------------------------------------------------------------------------
DXV_HD F4 normalise_pos(const float* p, const float* bound)
{
    F4 q;
    q.x = (p[0] - bound[0]) / bound[3];
    q.y = (p[1] - bound[1]) / bound[3];
    q.z = (p[2] - bound[2]) / bound[3];
    q.w = 0.0f;
    return q;
}

DXV_HD void tri_box(const F4& a, const F4& b, const F4& c, float lo[3], float hi[3])
{
    lo[0] = min_(min_(a.x, b.x), c.x) - kPad; hi[0] = max_(max_(a.x, b.x), c.x) + kPad;
    lo[1] = min_(min_(a.y, b.y), c.y) - kPad; hi[1] = max_(max_(a.y, b.y), c.y) + kPad;
    lo[2] = min_(min_(a.z, b.z), c.z) - kPad; hi[2] = max_(max_(a.z, b.z), c.z) + kPad;
}

// ------------------------------------------------------------------------------------------
// Directed float -> half conversion for the compressed traversal nodes (integer arithmetic only,
// identical on host and device).  half_down(x) <= x <= half_up(x) for every finite x;
// beyond the half range the bound saturates to +-65504 on the inner side and +-inf on the outer.
// ------------------------------------------------------------------------------------------
DXV_HD float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else {                      // subnormal half: m * 2^-24
            const float v = (float)m * 5.9604644775390625e-08f;
            bits = sign | __builtin_bit_cast(uint32_t, v);
        }
    } else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    return __builtin_bit_cast(float, bits);
}

// magnitude truncated toward zero; *inexact tells whether bits were dropped
DXV_HD uint16_t half_trunc_mag(float a /* >= 0, finite */, bool& inexact)
{
    const uint32_t bits = __builtin_bit_cast(uint32_t, a) & 0x7fffffffu;
    const int32_t e = (int32_t)(bits >> 23) - 127;
    const uint32_t m = (bits & 0x7fffffu) | 0x800000u;
    if (bits == 0) { inexact = false; return 0; }
    if (e > 15) { inexact = true; return 0x7bffu; }                 // above 65504: largest finite half
    if (e >= -14) {                                                  // normal half
        inexact = (m & 0x1fffu) != 0;
        return (uint16_t)(((uint32_t)(e + 15) << 10) | ((m >> 13) & 1023u));
    }
    const int32_t shift = 13 + (-14 - e);                            // subnormal half
    if (shift > 24) { inexact = true; return 0; }
    inexact = (m & ((1u << shift) - 1u)) != 0;
    return (uint16_t)(m >> shift);
}

DXV_HD uint16_t half_down(float x)
{
    bool inexact;
    if (!(x == x)) return 0xfc00u;                                   // NaN: -inf is a valid lower bound
    if (x >= 0.0f) { if (x > 3.0e38f) return 0x7bffu; return half_trunc_mag(x, inexact); }
    if (x < -3.0e38f) return 0xfc00u;
    const uint16_t h = half_trunc_mag(-x, inexact);
    return (uint16_t)(0x8000u | (inexact ? h + 1u : h));             // magnitude up (0x7bff + 1 = inf)
}

DXV_HD uint16_t half_up(float x)
{
    bool inexact;
    if (!(x == x)) return 0x7c00u;
    if (x <= 0.0f) { if (x < -3.0e38f) return 0xfbffu; const uint16_t h = half_trunc_mag(-x, inexact); return (uint16_t)(h ? (0x8000u | h) : 0u); }
    if (x > 3.0e38f) return 0x7c00u;
    const uint16_t h = half_trunc_mag(x, inexact);
    return (uint16_t)(inexact ? h + 1u : h);
}

DXV_HD Node32 compress_node(const Node& n)
{
    Node32 c;
    c.b[0] = half_down(n.lo0x); c.b[1] = half_down(n.lo1x); c.b[2] = half_up(n.hi0x);  c.b[3] = half_up(n.hi1x);
    c.b[4] = half_down(n.lo0y); c.b[5] = half_down(n.lo1y); c.b[6] = half_up(n.hi0y);  c.b[7] = half_up(n.hi1y);
    c.b[8] = half_down(n.lo0z); c.b[9] = half_down(n.lo1z); c.b[10] = half_up(n.hi0z); c.b[11] = half_up(n.hi1z);
    c.c0 = n.c0; c.c1 = n.c1;
    return c;
}

// Wide node of binary node i (dxv_types.h Node64).  `nodes` are the refitted exact nodes.
DXV_HD Node64 widen_node(const Node* nodes, int32_t i)
{
    Node64 w;
    int slot = 0;
    auto emit = [&](const float* box, int32_t link) {      // box = lo.xyz hi.xyz
        for (int a = 0; a < 3; ++a) {
            w.b[a * 8 + slot] = half_down(box[a]);
            w.b[a * 8 + 4 + slot] = half_up(box[3 + a]);
        }
        w.c[slot++] = link;
    };
    const Node n = nodes[i];
    const float* nb = reinterpret_cast<const float*>(&n);  // child 0: floats 0..5, child 1: floats 6..11
    const int32_t link[2] = {n.c0, n.c1};
    for (int side = 0; side < 2; ++side) {
        if (link[side] >= 0) {
            const Node m = nodes[link[side]];
            const float* mb = reinterpret_cast<const float*>(&m);
            emit(mb, m.c0);
            emit(mb + 6, m.c1);
        } else emit(nb + 6 * side, link[side]);
    }
    for (; slot < 4; ++slot) {
        for (int a = 0; a < 3; ++a) { w.b[a * 8 + slot] = 0x7c00u; w.b[a * 8 + 4 + slot] = 0xfc00u; }   // [+inf, -inf]
        w.c[slot] = kNoChild;
    }
    return w;
}

// ------------------------------------------------------------------------------------------
// Morton keys (30 bits over the centre of the padded box) and the Karras 2012 hierarchy rule.
// Keys are (morton << 32) | triangle index: unique, so the tree is deterministic.
// ------------------------------------------------------------------------------------------
DXV_HD uint32_t expand_bits10(uint32_t v)
{
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

DXV_HD uint32_t quant10(float c)
{
    float u = (c * 0.5f + 0.5f) * 1024.0f;
    if (!(u > 0.0f)) u = 0.0f;
    if (u > 1023.0f) u = 1023.0f;
    return (uint32_t)u;
}

DXV_HD uint64_t morton_key(const float lo[3], const float hi[3], uint32_t k)
{
    const uint32_t qx = quant10((lo[0] + hi[0]) * 0.5f);
    const uint32_t qy = quant10((lo[1] + hi[1]) * 0.5f);
    const uint32_t qz = quant10((lo[2] + hi[2]) * 0.5f);
    const uint32_t m = (expand_bits10(qx) << 2) | (expand_bits10(qy) << 1) | expand_bits10(qz);
    return ((uint64_t)m << 32) | (uint64_t)k;
}

DXV_HD int key_delta(const uint64_t* keys, int64_t T, int64_t i, int64_t j)
{
    if (j < 0 || j >= T) return -1;
    return __builtin_clzll(keys[i] ^ keys[j]);
}

// Internal node i of T-1 (T >= 2): children as links (>= 0 internal, < 0 ~leaf).
// left/right: children links (>= 0 internal node, < 0 ~leaf); other: the far end of the node's
// range of leaves [min(i, other), max(i, other)]; the left child ends at gamma = left >= 0 ? left : ~left.
DXV_HD void karras_node(const uint64_t* keys, int64_t T, int64_t i, int32_t& left, int32_t& right, uint32_t& other)
{
    const int d = key_delta(keys, T, i, i + 1) - key_delta(keys, T, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = key_delta(keys, T, i, i - d);
    int64_t lmax = 2;
    while (key_delta(keys, T, i, i + lmax * d) > dmin) lmax *= 2;
    int64_t l = 0;
    for (int64_t t = lmax / 2; t >= 1; t /= 2)
        if (key_delta(keys, T, i, i + (l + t) * d) > dmin) l += t;
    const int64_t j = i + l * d;
    const int dnode = key_delta(keys, T, i, j);
    int64_t s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (key_delta(keys, T, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int64_t gamma = i + s * d + (d < 0 ? -1 : 0);
    const int64_t lo = i < j ? i : j, hi = i < j ? j : i;
    left = lo == gamma ? ~(int32_t)gamma : (int32_t)gamma;
    right = hi == gamma + 1 ? ~(int32_t)(gamma + 1) : (int32_t)(gamma + 1);
    other = (uint32_t)j;
}

DXV_HD void karras_node(const uint64_t* keys, int64_t T, int64_t i, int32_t& left, int32_t& right)
{
    uint32_t other;
    karras_node(keys, T, i, left, right, other);
}

} // namespace dxv
