// dxv_dirmap.h -- direction-space candidate lists for the reference rule (shared by the kernels,
// the device builder and tests/hostcheck).
//
// Every ray of the reference rule is radial: origin p = voxel centre, direction p / |p|
// (Content/Shaders/DXRVoxelizer.hlsl:44-53), so the points of a ray are (rho + t) * dir with
// rho = |p|: in direction space a ray is one POINT plus a start radius.  A cube map around the grid
// centre (6 faces of R x R texels, face coordinates u = p_b / |p_a|, v = p_c / |p_a| on the face of
// the dominant axis a) lists, per texel, the triangles whose projection can reach it.  A ray reads
// the list of its texel and hands the few entries whose footprint box contains (u, v) and whose
// radial range lies beyond rho to the same canonical triangle step as the tree walks
// (leaf_reference: padded-box slab test, watertight test, closest = min (t, k)).  No stack, no tree.
//
// Identical results need only one property: every triangle that leaf_reference would accept for a
// ray is listed in that ray's texel with a footprint containing the ray and a radial range that
// passes.  leaf_reference accepts a triangle when the ray passes the slab test of its box padded by
// 2^-16 AND the fp32 watertight test: the edge functions of a triangle that is not degenerate only
// agree in sign within ~1e-6 of it (inputs are differences of coordinates in [-1, 1], rounding
// ~1e-7), and for a degenerate sliver, whose edge functions all vanish along its line, the padded
// box keeps the ray within sqrt(3) 2^-16 = 2.64e-5 of it.  Footprints are those of the triangle
// dilated by kDmDelta = 2^-15 = 3.05e-5, clipped against face frusta widened by 2^-10, and then
// rounded outward to halfs; triangles closer than 64 kDmDelta to the centre along the face axis
// take the whole face.  tests/ (fuzz against the oracle, lattice-snapped adversarial meshes) and the GPU
// soak check the claim the same way they check the padded leaf boxes.
#pragma once
#include "dxv_trace.h"

namespace dxv {

struct alignas(16) DirEntry {
    uint16_t u0, u1, v0, v1;   // footprint box in face coordinates (halfs, rounded outward); u0 > u1: no footprint
    uint16_t r0, r1;           // radial range of the triangle part inside the face frustum (halfs, rounded outward)
    uint32_t tri;              // position in the scene's triangle order (TriPos index)
};
static_assert(sizeof(DirEntry) == 16, "one 16-byte load per entry");

struct alignas(16) DirCell {                        // one 16-byte load per ray
    uint32_t begin, end;                            // entries [begin, end) of a texel
    uint32_t r1max;                                 // far radius of its last entry (half bits): a ray that starts beyond it has no candidate
    uint32_t thick;                                 // largest radial extent r1 - r0 of its entries (half bits, rounded up): where a scan may stop
};

struct DirMapView {
    const DirCell* cells;      // 6 * R * R, cell = (face * R + j) * R + i; NULL: no map
    const DirEntry* entries;
    uint32_t R;                // texels per face side, a power of two
};

constexpr float kDmDelta = 3.0517578125e-5f;        // 2^-15: dilation of the triangles
constexpr double kDmFrustum = 1.0 + 1.0 / 1024.0;   // face frusta are widened by this factor

// texel index of face coordinate u (monotone in u: add, multiply by a power of two, floor)
DXV_HD uint32_t dm_texel(float u, uint32_t R)
{
    const float x = (u + 1.0f) * (0.5f * (float)R);
    if (!(x > 0.0f)) return 0u;
    const uint32_t i = (uint32_t)x;
    return i < R ? i : R - 1u;
}

// face = 2 * axis + (negative side); in-face axes (b, c) = ((axis + 1) % 3, (axis + 2) % 3)
DXV_HD void dm_ray_point(float ox, float oy, float oz, uint32_t& face, float& u, float& v, float& rho)
{
    const float ax = __builtin_fabsf(ox), ay = __builtin_fabsf(oy), az = __builtin_fabsf(oz);
    if (ax >= ay && ax >= az) { face = ox < 0.0f ? 1u : 0u; u = oy / ax; v = oz / ax; }
    else if (ay >= az) { face = oy < 0.0f ? 3u : 2u; u = oz / ay; v = ox / ay; }
    else { face = oz < 0.0f ? 5u : 4u; u = ox / az; v = oy / az; }
    rho = __builtin_sqrtf((ox * ox + oy * oy) + oz * oz);
}

// next float towards +inf / -inf (finite inputs)
DXV_HD float dm_up(float x)
{
    if (x == 0.0f) return 1.401298464e-45f;
    const uint32_t b = __builtin_bit_cast(uint32_t, x);
    return __builtin_bit_cast(float, x > 0.0f ? b + 1u : b - 1u);
}
DXV_HD float dm_down(float x) { return -dm_up(-x); }

// Footprint of triangle tp on one face, or false when it cannot be seen through that face.
// Builder side only (one call per triangle and face): double precision, nothing canonical here --
// the result only has to be a superset.
struct DirFootprint { float u0, u1, v0, v1, r0, r1; };

DXV_HD bool dm_footprint(const TriPos& tp, uint32_t face, DirFootprint& out)
{
    const uint32_t a = face >> 1, b = (a + 1u) % 3u, c = (a + 2u) % 3u;
    const double s = (face & 1u) ? -1.0 : 1.0;
    const float vx[3][3] = {{tp.v0.x, tp.v0.y, tp.v0.z}, {tp.v1.x, tp.v1.y, tp.v1.z}, {tp.v2.x, tp.v2.y, tp.v2.z}};
    double poly[2][10][3];
    int n = 3, cur = 0;
    for (int i = 0; i < 3; ++i) { poly[0][i][0] = vx[i][b]; poly[0][i][1] = vx[i][c]; poly[0][i][2] = s * (double)vx[i][a]; }
    const double delta = (double)kDmDelta;
    // the four side planes of the widened frustum, pushed out by the dilation: kF * d +- b + 2 delta >= 0
    const double planes[4][2] = {{1.0, 0.0}, {-1.0, 0.0}, {0.0, 1.0}, {0.0, -1.0}};
    // most (triangle, face) pairs end here: all three vertices outside one side plane (what the clip
    // below would find, without its arrays)
    for (int pl = 0; pl < 4; ++pl) {
        bool anyIn = false;
        for (int i = 0; i < 3; ++i)
            anyIn = anyIn || kDmFrustum * poly[0][i][2] + planes[pl][0] * poly[0][i][0] + planes[pl][1] * poly[0][i][1] + 2.0 * delta >= 0.0;
        if (!anyIn) return false;
    }
    for (int pl = 0; pl < 4 && n > 0; ++pl) {
        const double nb = planes[pl][0], nc = planes[pl][1];
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const double* A = poly[cur][i];
            const double* B = poly[cur][(i + 1) % n];
            const double fa = kDmFrustum * A[2] + nb * A[0] + nc * A[1] + 2.0 * delta;
            const double fb = kDmFrustum * B[2] + nb * B[0] + nc * B[1] + 2.0 * delta;
            if (fa >= 0.0) { for (int k = 0; k < 3; ++k) poly[cur ^ 1][m][k] = A[k]; ++m; }
            if ((fa >= 0.0) != (fb >= 0.0)) {
                const double w = fa / (fa - fb);
                for (int k = 0; k < 3; ++k) poly[cur ^ 1][m][k] = A[k] + (B[k] - A[k]) * w;
                ++m;
            }
        }
        cur ^= 1;
        n = m;
    }
    if (n == 0) return false;
    double dmin = poly[cur][0][2], rmax = 0.0;
    for (int i = 0; i < n; ++i) {
        const double* q = poly[cur][i];
        if (q[2] < dmin) dmin = q[2];
        const double r = __builtin_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        if (r > rmax) rmax = r;
    }
    const double full = kDmFrustum + 1.0 / 256.0;
    double u0 = -full, u1 = full, v0 = -full, v1 = full;
    if (dmin >= 64.0 * delta) {
        u0 = v0 = 1e300; u1 = v1 = -1e300;
        for (int i = 0; i < n; ++i) {
            const double* q = poly[cur][i];
            const double u = q[0] / q[2], v = q[1] / q[2];
            if (u < u0) u0 = u;
            if (u > u1) u1 = u;
            if (v < v0) v0 = v;
            if (v > v1) v1 = v;
        }
        // a point moved by delta sideways and in depth at depth >= dmin >= 64 delta, |u| <= 1.006:
        // du <= (1 + |u|) delta / (dmin - delta) <= 2.04 delta / dmin
        const double pad = 2.25 * delta / dmin + 1e-6;
        u0 -= pad; u1 += pad; v0 -= pad; v1 += pad;
        if (u0 < -full) u0 = -full;
        if (v0 < -full) v0 = -full;
        if (u1 > full) u1 = full;
        if (v1 > full) v1 = full;
    }
    // radial range: the farthest point of a convex polygon is a vertex; the nearest is the foot of the perpendicular from
    // the centre onto its plane when that lies inside the polygon, else the nearest point of its boundary.  (A tight near
    // radius matters twice: it is what culls entries behind a hit, and the thickest entry of a texel decides how far
    // behind a hit the scan of that texel goes on.)
    double rmin = 1e300;
    for (int i = 0; i < n; ++i) {                                      // nearest point of every edge (covers degenerate polygons)
        const double* A = poly[cur][i];
        const double* B = poly[cur][(i + 1) % n];
        const double d[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
        const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        double w = dd > 0.0 ? -(A[0] * d[0] + A[1] * d[1] + A[2] * d[2]) / dd : 0.0;
        w = w < 0.0 ? 0.0 : w > 1.0 ? 1.0 : w;
        const double x = A[0] + w * d[0], y = A[1] + w * d[1], z = A[2] + w * d[2];
        const double dist = __builtin_sqrt(x * x + y * y + z * z);
        if (dist < rmin) rmin = dist;
    }
    if (n >= 3) {
        // plane through the polygon (its vertices are coplanar: clipped from one triangle), in the polygon's own coordinates
        const double* A = poly[cur][0];
        double nx = 0.0, ny = 0.0, nz = 0.0;
        for (int i = 1; i + 1 < n; ++i) {                              // summed fan normals: robust for thin clips
            const double* B = poly[cur][i];
            const double* C = poly[cur][i + 1];
            const double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
            nx += e1[1] * e2[2] - e1[2] * e2[1]; ny += e1[2] * e2[0] - e1[0] * e2[2]; nz += e1[0] * e2[1] - e1[1] * e2[0];
        }
        const double nn = nx * nx + ny * ny + nz * nz;
        if (nn > 1e-60) {
            const double k = (nx * A[0] + ny * A[1] + nz * A[2]) / nn;
            const double P[3] = {k * nx, k * ny, k * nz};               // foot of the perpendicular from the centre
            bool inside = true;
            for (int i = 0; i < n && inside; ++i) {
                const double* E = poly[cur][i];
                const double* F = poly[cur][(i + 1) % n];
                const double d[3] = {F[0] - E[0], F[1] - E[1], F[2] - E[2]}, q[3] = {P[0] - E[0], P[1] - E[1], P[2] - E[2]};
                const double cx = d[1] * q[2] - d[2] * q[1], cy = d[2] * q[0] - d[0] * q[2], cz = d[0] * q[1] - d[1] * q[0];
                inside = cx * nx + cy * ny + cz * nz >= 0.0;            // same turn as the polygon's own orientation (the fan normal)
            }
            const double dist = __builtin_fabs(k) * __builtin_sqrt(nn);
            if (inside && dist < rmin) rmin = dist;
        }
    }
    rmin *= 1.0 - 1e-6;
    rmin -= 4.0 * delta;
    if (rmin < 0.0) rmin = 0.0;
    rmax += 4.0 * delta;
    out.u0 = (float)u0; out.u1 = (float)u1; out.v0 = (float)v0; out.v1 = (float)v1;
    out.r0 = (float)rmin; out.r1 = (float)rmax;
    // float conversion rounds to nearest: one more ulp outward
    out.u0 = dm_down(out.u0); out.v0 = dm_down(out.v0); out.u1 = dm_up(out.u1); out.v1 = dm_up(out.v1);
    out.r0 = out.r0 > 0.0f ? dm_down(out.r0) : 0.0f;
    out.r1 = dm_up(out.r1);
    return true;
}

// the record of (triangle, face) as it is stored in the lists; u0 > u1 when there is no footprint
DXV_HD DirEntry dm_entry(const TriPos& tp, uint32_t face, uint32_t tri)
{
    DirEntry e;
    DirFootprint f;
    e.tri = tri;
    if (!dm_footprint(tp, face, f)) { e.u0 = 0x3c00u; e.u1 = 0u; e.v0 = 0x3c00u; e.v1 = 0u; e.r0 = 0u; e.r1 = 0u; return e; }   // 1 > 0
    e.u0 = half_down(f.u0); e.u1 = half_up(f.u1); e.v0 = half_down(f.v0); e.v1 = half_up(f.v1);
    e.r0 = half_down(f.r0); e.r1 = half_up(f.r1);
    return e;
}

// texel rectangle [i0, i1] x [j0, j1] of an entry (false: none)
DXV_HD bool dm_rect(const DirEntry& e, uint32_t R, uint32_t& i0, uint32_t& i1, uint32_t& j0, uint32_t& j1)
{
    const float u0 = half_bits_to_float(e.u0), u1 = half_bits_to_float(e.u1);
    const float v0 = half_bits_to_float(e.v0), v1 = half_bits_to_float(e.v1);
    if (u0 > u1) return false;
    if (u1 < -1.0f || u0 > 1.0f || v1 < -1.0f || v0 > 1.0f) return false;      // rays only have |u|, |v| <= 1
    i0 = dm_texel(u0, R); i1 = dm_texel(u1, R); j0 = dm_texel(v0, R); j1 = dm_texel(v1, R);
    return true;
}

// List order: by texel, inside a texel by far radius r1 ascending (then by triangle).  Sort key:
//   cell (cellBits = bits of 6 R R - 1) | r1 as half (16 bits; positive halfs order like integers) | triangle (the rest)
struct DirKeyLayout { uint32_t cellBits, triBits; };
DXV_HD DirKeyLayout dm_key_layout(uint32_t R)
{
    DirKeyLayout k;
    k.cellBits = 1u;
    while ((6ull * R * R - 1ull) >> k.cellBits) ++k.cellBits;
    k.triBits = 64u - 16u - k.cellBits;
    return k;
}
DXV_HD uint64_t dm_key(const DirKeyLayout& k, uint32_t cell, uint16_t r1, uint32_t tri)
{
    return ((uint64_t)cell << (64u - k.cellBits)) | ((uint64_t)r1 << k.triBits) | (uint64_t)tri;
}
DXV_HD uint32_t dm_key_cell(const DirKeyLayout& k, uint64_t key) { return (uint32_t)(key >> (64u - k.cellBits)); }
DXV_HD uint32_t dm_key_tri(const DirKeyLayout& k, uint64_t key) { return (uint32_t)(key & ((1ull << k.triBits) - 1ull)); }

// closest hit of the reference rule through the lists
// Two steps like the postponed-leaf walks: scanning entries is short and cheap, the triangle step is
// long, so the triangles an entry scan selects are queued in the thread's LDS column (cap entries)
// and tested when some lane's queue is full or every lane has finished scanning -- all lanes with
// work test together.
// ABL (timing-only builds, tools/ablate.py; 0 in every shipped path): 8 = stop before the texel lookup, 1 = stop after it,
// 2 = scan the entries but test no triangle
template <class Stack, int ABL = 0>
DXV_HD void trace_reference_dm(Ray& r, const DirMapView& dm, const TriPos* tris, const Stack& stk, int cap, Hit& best)
{
    best.t = kTMax; best.b1 = 0.0f; best.b2 = 0.0f; best.k = 0xffffffffu; best.leaf = -1;
    if (ABL & 8) return;
    uint32_t face;
    float u, v, rho;
    dm_ray_point(r.ox, r.oy, r.oz, face, u, v, rho);
    const DirCell cell = dm.cells[(face * dm.R + dm_texel(v, dm.R)) * dm.R + dm_texel(u, dm.R)];
    const float near = rho * 0.999f;
    // entries wholly nearer the centre than the ray's start (r1 < near: t < 0) come first: skip them --
    // all of them at once for a ray that starts beyond the texel's last triangle
    uint32_t i = cell.begin, hi = cell.end;
    if (half_bits_to_float(cell.r1max) < near) i = hi;
    if (ABL & 1) { if (i == 0xffffffffu) best.k = 0u; return; }
    while (hi - i > 8u) {
        const uint32_t mid = i + ((hi - i) >> 1);
        if (half_bits_to_float(dm.entries[mid].r1) < near) i = mid + 1u; else hi = mid;
    }
    int qn = 0;
    const float thick = half_bits_to_float(cell.thick);
    auto consider = [&](const DirEntry& e) {
        if (!(half_bits_to_float(e.r1) < near) &&
            !(u < half_bits_to_float(e.u0) || u > half_bits_to_float(e.u1)) &&
            !(v < half_bits_to_float(e.v0) || v > half_bits_to_float(e.v1)) &&
            !(half_bits_to_float(e.r0) > (rho + best.t) * 1.001f + 1e-4f))          // wholly beyond the closest hit so far
            stk.put(qn++, (int32_t)e.tri);
    };
    for (;;) {
        // four entries per round, all four loads in flight before the first is looked at (two per round:
        // +13 % on the 1 M-triangle scene, one: +40 %)
        if (i < cell.end) {
            const uint32_t last = cell.end - 1u;
            // (the third and fourth load are skipped when no lane of the wave has that many entries left:
            // -5 % on the 1 M-triangle scene; voting on the second one as well: +7 %)
            const bool wide = wave_any(i + 2u <= last);
            const DirEntry e0 = dm.entries[i], e1 = dm.entries[i + 1u < last ? i + 1u : last];
            DirEntry e2 = e0, e3 = e0;
            if (wide) { e2 = dm.entries[i + 2u < last ? i + 2u : last]; e3 = dm.entries[i + 3u < last ? i + 3u : last]; }
            // The list is sorted by far radius and no entry of the texel is thicker than `thick`: once an entry ends more
            // than that beyond the closest hit so far, it and everything behind it START beyond the hit.  (Surface meshes
            // have short lists and gain little; in a deep soup a ray stops after the first few of hundreds of entries.)
            if (half_bits_to_float(e0.r1) - thick > (rho + best.t) * 1.001f + 1e-4f) i = cell.end;
            else {
                consider(e0);
                if (i + 1u <= last) consider(e1);
                if (i + 2u <= last) consider(e2);
                if (i + 3u <= last) consider(e3);
                i += 4u;
            }
        }
        const bool scanning = wave_any(i < cell.end);
        if (scanning && !wave_any(qn + 4 > cap)) continue;
        if (ABL & 2) { if (qn > 100) best.k = 0u; }
        else
        for (int k = 0; wave_any(k < qn); ++k)
            if (k < qn) leaf_reference(r, tris, stk.get(k), best);
        qn = 0;
        if (!scanning) break;
    }
}

template <class Stack, int ABL>
DXV_HD void trace_reference_lists(Ray& r, const SceneView& sc, const Stack& stk, int cap, Hit& best)
{
    const DirMapView dm{static_cast<const DirCell*>(sc.dmCells), static_cast<const DirEntry*>(sc.dmEntries), sc.dmR};
    trace_reference_dm<Stack, ABL>(r, dm, sc.triPos, stk, cap, best);
}

} // namespace dxv
