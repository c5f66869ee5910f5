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

// One entry per (triangle, texel), everything in small integers so that a ray decides in a handful of integer
// instructions whether the triangle is worth the 48-byte fetch and the watertight test:
//   box   bytes (x0, y0, 127 - x1, 127 - y1): the footprint's box, cut to the texel, in texel-local cells (a texel is
//         128 x 128 cells).  A ray in cell (x, y) passes iff every byte of (x, y, 127 - x, 127 - y) is >= this word's
//         byte: one subtraction of packed bytes (dm_local_pass).
//   edge  signed bytes (a, b, c1, c2): the ONE edge of the projected, dilated triangle that cuts most off that box.
//         The ray passes iff a x + b y + 127 c1 + c2 >= 0 (one v_dot4c_i32_i8); 0 = no edge (clipped polygons, or no
//         edge crosses the box).  A triangle fills at most half of its box: this halves the triangle tests.
//   rr    (0x7fff - r0) | r1 << 16 | 0x80008000, r0 / r1 the outward-rounded radial range as halfs: positive halfs
//         order like integers, so r1 >= near and r0 <= (start + closest hit) are one packed subtraction as well.
//   tri   position in the scene's triangle order (TriPos index) in the low 26 bits; the high 6 bits say how far behind this
//         entry's far radius the EARLIEST start of any later entry of the texel lies, in 63rds of the texel's thickest
//         entry (rounded up): the scan stops at the first entry for which that point is beyond the closest hit
//         (dm_stop_radius).  Lists therefore serve scenes of up to 2^26 triangles.
struct alignas(16) DirEntry {
    uint32_t box, edge, rr, tri;
};
static_assert(sizeof(DirEntry) == 16, "one 16-byte load per entry");
constexpr uint32_t kDmCells = 128u;                 // cells per texel side (7 bits)

// Record of a (triangle, face) pair, builder side: the footprint as before plus the projected triangle, from which
// the per-texel entries are cut.
struct alignas(16) DirRecord {
    float u0, u1, v0, v1;      // footprint box in face coordinates, rounded outward; u0 > u1: no footprint
    uint32_t rr;               // radial range of the whole footprint, halfs rounded outward: r0 | r1 << 16
    uint32_t hasTri;           // bit 0: px / py / pw hold the projected triangle (all three vertices in front of the face plane); bit 1: dm_record_on_map
    float pad;                 // dilation of the projected triangle in face coordinates
    float px[3], py[3];        // projected vertices
    float pw[3];               // 1 / depth of the vertices along the face axis: linear over the projected triangle's plane (dm_local_radial)
};
static_assert(sizeof(DirRecord) == 64, "record layout");

struct alignas(16) DirCell {                        // one 16-byte load per ray
    uint32_t begin;                                 // the texel's entries: [begin, begin + count)
    uint16_t count;                                 // (a scene with more than 65,535 entries in one texel keeps the tree walk)
    uint16_t r1max;                                 // far radius of its last entry (half bits): a ray that starts beyond it has no candidate
    uint16_t thick;                                 // largest radial extent r1 - r0 of its entries (half bits, rounded up): the unit of the entries' stop codes
    // The first two steps of the binary search for a ray's start need no load: the far radii (half bits) of the entries
    // the search would look at -- the middle one, and the middles of both halves (dm_search_hints).  Lists of up to 35
    // entries (all of a surface mesh's) are then down to the eight entries a scan round starts with anyway.
    uint16_t q1, q2, q3;
};
static_assert(sizeof(DirCell) == 16, "one 16-byte load per texel");

// entries the start search looks at first in a list of n entries (relative indices; n > 8): the middle of [0, n), and the
// middles of [0, m2) and [m2 + 1, n) where those halves are still longer than 8 -- exactly the sequence of the loop
// `while (hi - lo > 8) { mid = lo + (hi - lo) / 2; r1[mid] < near ? lo = mid + 1 : hi = mid; }`
struct DirSearchHints { uint32_t m1, m2, m3; bool has1, has3; };
DXV_HD DirSearchHints dm_search_hints(uint32_t n)
{
    DirSearchHints h;
    h.m2 = n >> 1;
    h.has1 = h.m2 > 8u;                       // left half [0, m2)
    h.m1 = h.m2 >> 1;
    h.has3 = n - (h.m2 + 1u) > 8u;            // right half [m2 + 1, n)
    h.m3 = h.m2 + 1u + ((n - (h.m2 + 1u)) >> 1);
    return h;
}

struct DirMapView {
    const DirCell* cells;      // 6 * R * R, cell = (face * R + j) * R + i; NULL: no map
    const DirEntry* entries;
    uint32_t R;                // texels per face side, a power of two
    uint32_t coop = 0u;        // device: 1 = a lone lane's long list is scanned by its whole wave (trace_reference_dm_from)
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

// texel AND texel-local cell (0 .. 127) of face coordinate u: lexicographically monotone in u, evaluated identically by
// builder and kernel (x - i is exact for x in [i, i + 1), the scaling a power of two)
DXV_HD void dm_local(float u, uint32_t R, uint32_t& texel, uint32_t& cell)
{
    const float x = (u + 1.0f) * (0.5f * (float)R);
    if (!(x > 0.0f)) { texel = 0u; cell = 0u; return; }
    const uint32_t i = (uint32_t)x;
    if (i >= R) { texel = R - 1u; cell = kDmCells - 1u; return; }
    const uint32_t c = (uint32_t)((x - (float)i) * (float)kDmCells);
    texel = i;
    cell = c < kDmCells ? c : kDmCells - 1u;
}

// a . b over four signed bytes (v_dot4c_i32_i8 on the device)
DXV_HD int32_t dm_dot4(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sdot4((int)a, (int)b, 0, false);
#else
    int32_t s = 0;
    for (int k = 0; k < 4; ++k) s += (int32_t)(int8_t)(a >> (8 * k)) * (int32_t)(int8_t)(b >> (8 * k));
    return s;
#endif
}

// The ray's side of an entry test: q = bytes (x, y, 127 - x, 127 - y) | 0x80808080, p = bytes (x, y, 127, 1),
// rc = (0x7fff - half_down(bound)) | half_up(near) << 16 (dm_radial_word)
struct DirRayLocal { uint32_t q, p; };
DXV_HD DirRayLocal dm_ray_local(uint32_t x, uint32_t y)
{
    DirRayLocal l;
    l.q = (x | (y << 8) | ((kDmCells - 1u - x) << 16) | ((kDmCells - 1u - y) << 24)) | 0x80808080u;
    l.p = x | (y << 8) | (127u << 16) | (1u << 24);
    return l;
}
// Directed float -> half for the values a ray brings to the integer radial test: positive, inside the half's normal
// range [2^-14, 65504) (a ray starts at least half a voxel diagonal, 4e-4 even at 2048^3, from the centre).  There the
// half's bits are the float's, rebiased and cut: truncation rounds down, adding the dropped bits' mask first rounds up
// (the carry runs into the exponent as it should).  Equal to half_down / half_up (dxv_math.h) on that range (tests).
DXV_HD uint32_t dm_half_down_pos(float x) { return (__builtin_bit_cast(uint32_t, x) - 0x38000000u) >> 13; }
DXV_HD uint32_t dm_half_up_pos(float x) { return (__builtin_bit_cast(uint32_t, x) - 0x38000000u + 0x1fffu) >> 13; }
DXV_HD uint32_t dm_radial_word(float near, float bound)
{
    const uint32_t b = bound < 65504.0f ? dm_half_down_pos(bound) : 0x7bffu;
    return (0x7fffu - b) | (dm_half_up_pos(near) << 16);
}
// box, edge and radial range of an entry against a ray: true = fetch and test the triangle
DXV_HD bool dm_local_pass(const DirEntry& e, const DirRayLocal& l, uint32_t rc)
{
    return ((l.q - e.box) & 0x80808080u) == 0x80808080u && ((e.rr - rc) & 0x80008000u) == 0x80008000u && dm_dot4(e.edge, l.p) >= 0;
}
constexpr uint32_t kDmTriBits = 26u, kDmTriMask = (1u << kDmTriBits) - 1u;
DXV_HD uint32_t dm_entry_tri(const DirEntry& e) { return e.tri & kDmTriMask; }
// The lists are sorted by far radius.  No entry from this one on starts before  r1 - q * (thick / 63):  q is chosen by
// the builder (dm_stop_code) with this very expression, so the float arithmetic is the same on both sides.
DXV_HD float dm_stop_step(float thick) { return thick * 0.015873017f; }                     // a little more than 1 / 63
DXV_HD float dm_stop_radius(const DirEntry& e, float step) { return half_bits_to_float((e.rr >> 16) & 0x7fffu) - (float)(e.tri >> kDmTriBits) * step; }
// smallest q in [0, 63] with r1 - q * step <= suffixMinR0 (r1 - 63 * step <= r1 - thick <= every later start: always found)
DXV_HD uint32_t dm_stop_code(float r1, float suffixMinR0, float step)
{
    // start two below the quotient (every smaller q misses by more than a whole step, far beyond rounding) and count up
    const float x = step > 0.0f ? (r1 - suffixMinR0) / step : 0.0f;
    uint32_t q = x > 2.0f ? (x < 65.0f ? (uint32_t)x - 2u : 63u) : 0u;
    while (q < 63u && r1 - (float)q * step > suffixMinR0) ++q;
    return q;
}
DXV_HD float dm_entry_r1(const DirEntry& e) { return half_bits_to_float((e.rr >> 16) & 0x7fffu); }
DXV_HD float dm_entry_r0(const DirEntry& e) { return half_bits_to_float(0x7fffu - (e.rr & 0x7fffu)); }

// face = 2 * axis + (negative side); in-face axes (b, c) = ((axis + 1) % 3, (axis + 2) % 3)
DXV_HD void dm_ray_point(float ox, float oy, float oz, uint32_t& face, float& u, float& v, float& rho)
{
    const float ax = __builtin_fabsf(ox), ay = __builtin_fabsf(oy), az = __builtin_fabsf(oz);
    // (the two quotients share their denominator: div_by, dxv_math.h -- the same bits as `/` for these operands, checked exhaustively)
    float a, nb, nc;
    if (ax >= ay && ax >= az) { face = ox < 0.0f ? 1u : 0u; a = ax; nb = oy; nc = oz; }
    else if (ay >= az) { face = oy < 0.0f ? 3u : 2u; a = ay; nb = oz; nc = ox; }
    else { face = oz < 0.0f ? 5u : 4u; a = az; nb = ox; nc = oy; }
    const RcpRefined byA = rcp_refined(a);
    u = div_by(nb, byA); v = div_by(nc, byA);
    rho = sqrt_in_range((ox * ox + oy * oy) + oz * oz);
}

// The first step of a ray through the lists -- its texel, its cell inside the texel, and whether it has any candidate at
// all: a ray whose texel is empty, or that starts beyond the far radius of its texel's last entry, is a miss after this
// one load.  ONE definition: the kernel (trace_reference_dm), the plan's exact checker (k_plan_check, traverse.hip) and
// the exhaustive list check make this decision through it, and the brick test below bounds it from above.
struct DirRayStart { DirCell cell; uint32_t cx, cy; float rho, near; bool live; };
DXV_HD DirRayStart dm_ray_start(float ox, float oy, float oz, const DirMapView& dm)
{
    DirRayStart s;
    uint32_t face, ti, tj;
    float u, v;
    dm_ray_point(ox, oy, oz, face, u, v, s.rho);
    dm_local(u, dm.R, ti, s.cx); dm_local(v, dm.R, tj, s.cy);
    s.cell = dm.cells[(face * dm.R + tj) * dm.R + ti];
    s.near = s.rho * 0.999f;
    s.live = s.cell.count != 0u && !(half_bits_to_float(s.cell.r1max) < s.near);
    return s;
}

// ---------------------------------------------------------------------------------------------
// Max-mip of the texels' far radii: which 4 x 4 x 4-voxel bricks can hold a live ray at all (the launch's work queue,
// k_plan_bricks in traverse.hip).  Key of a texel = its r1max as half bits (positive halfs order like integers), 0 for an
// empty texel; level l holds, for 6 faces of (R >> l)^2 cells, the maximum over the 2^l x 2^l texels below a cell.
// Built from the cells (dirmap.hip: dirmap_mip), never exported: a function of the cells alone.
// ---------------------------------------------------------------------------------------------
DXV_HD uint32_t dm_mip_levels(uint32_t R) { uint32_t l = 1u; while ((R >> (l - 1u)) > 1u) ++l; return l; }   // R = 256: 9 (256 .. 1)
DXV_HD uint32_t dm_mip_offset(uint32_t R, uint32_t level)
{
    uint32_t off = 0;
    for (uint32_t l = 0; l < level; ++l) off += 6u * (R >> l) * (R >> l);
    return off;
}
DXV_HD uint32_t dm_mip_words(uint32_t R) { return dm_mip_offset(R, dm_mip_levels(R)); }    // 16-bit words
DXV_HD uint32_t dm_mip_key(const DirCell& c) { return c.count ? (uint32_t)c.r1max : 0u; }
DXV_HD uint32_t dm_mip_count_key(const DirCell& c) { return (uint32_t)c.count; }       // the second mip (behind the first): list lengths

// maximum key over texels [i0, i1] x [j0, j1] of a face (a superset: the coarsest level at which the rectangle is at most
// 2 x 2 cells; four loads)
DXV_HD uint32_t dm_mip_max(const uint16_t* mip, uint32_t R, uint32_t face, uint32_t i0, uint32_t i1, uint32_t j0, uint32_t j1)
{
    uint32_t l = 0, off = 0, r = R;
    while (((i1 >> l) - (i0 >> l)) > 1u || ((j1 >> l) - (j0 >> l)) > 1u) { off += 6u * r * r; r >>= 1; ++l; }
    const uint16_t* base = mip + off + face * r * r;
    const uint32_t a0 = i0 >> l, a1 = i1 >> l, b0 = j0 >> l, b1 = j1 >> l;
    uint32_t m = base[b0 * r + a0], t = base[b0 * r + a1];
    if (t > m) m = t;
    t = base[b1 * r + a0]; if (t > m) m = t;
    t = base[b1 * r + a1]; if (t > m) m = t;
    return m;
}

// Can a voxel whose centre lies in the box [x0, x1] x [y0, y1] x [z0, z1] (the hull of a brick's voxel centres; x0 <= x1 ...)
// hold a live ray, i.e. pass origin_leaves_root and dm_ray_start().live?  False negatives are not allowed, false positives
// cost a wave that finds nothing to do.  Every step bounds the per-voxel arithmetic through its monotonicity in fp32
// (correctly rounded operations are monotone): coordinates of one sign, |u| = |p_b| / |p_a| grows with |p_b| and falls
// with |p_a|, the texel index grows with u, rho grows with every |coordinate|, near = rho * 0.999f with rho.
//  * root box: on an axis whose coordinates are all of one sign, axis_leaves_root is monotone -- all of them leave iff the
//    one nearest to zero does;
//  * a box that straddles a centre plane (grids whose half is no multiple of the brick) is kept;
//  * faces: in one octant a box meets the cone of face A iff its corner with the largest |a| and smallest |b|, |c| lies in
//    it, and likewise for the tie-breaking order of dm_ray_point -- the faces present among its voxels are exactly the
//    faces of three of its corners; per present face the texel rectangle of the box comes from the extreme quotients;
//  * every ray of the box starts at or beyond rhoMin: dead iff every texel it can look into is empty or ends before
//    rhoMin * 0.999f.
// (the faces a box of one octant can look into, each with the rectangle of texels its rays can fall on: visit(face, i0, i1, j0, j1)
// returns true to stop; the box must not straddle a centre plane)
template <class Visit>
DXV_HD bool dm_box_faces(float x0, float x1, float y0, float y1, float z0, float z1, uint32_t R, Visit&& visit)
{
    const bool nx = x1 < 0.0f, ny = y1 < 0.0f, nz = z1 < 0.0f;
    const float xa = nx ? -x1 : x0, xb = nx ? -x0 : x1, ya = ny ? -y1 : y0, yb = ny ? -y0 : y1, za = nz ? -z1 : z0, zb = nz ? -z0 : z1;
    // rectangle of texels for numerator range [na, nb] (sign neg) over denominator range [da, db]
    auto range = [&](float na, float nb, bool neg, float da, float db, uint32_t& t0, uint32_t& t1) {
        const float lo = na / db, hi = nb / da;
        t0 = dm_texel(neg ? -hi : lo, R); t1 = dm_texel(neg ? -lo : hi, R);
    };
    uint32_t i0, i1, j0, j1;
    if (xb >= ya && xb >= za) {                                         // face X: u = y / |x|, v = z / |x|
        range(ya, yb, ny, xa, xb, i0, i1); range(za, zb, nz, xa, xb, j0, j1);
        if (visit(nx ? 1u : 0u, i0, i1, j0, j1)) return true;
    }
    if (!(xa >= yb && xa >= za) && yb >= za) {                          // face Y: u = z / |y|, v = x / |y|
        range(za, zb, nz, ya, yb, i0, i1); range(xa, xb, nx, ya, yb, j0, j1);
        if (visit(ny ? 3u : 2u, i0, i1, j0, j1)) return true;
    }
    if (!(xa >= ya && xa >= zb) && !(ya >= zb)) {                       // face Z: u = x / |z|, v = y / |z|
        range(xa, xb, nx, za, zb, i0, i1); range(ya, yb, ny, za, zb, j0, j1);
        if (visit(nz ? 5u : 4u, i0, i1, j0, j1)) return true;
    }
    return false;
}
DXV_HD bool dm_box_straddles(float x0, float x1, float y0, float y1, float z0, float z1)
{
    return !(x0 > 0.0f || x1 < 0.0f) || !(y0 > 0.0f || y1 < 0.0f) || !(z0 > 0.0f || z1 < 0.0f);
}
DXV_HD bool dm_box_may_be_live(float x0, float x1, float y0, float y1, float z0, float z1, const float* rootLo, const float* rootHi,
                               const uint16_t* mip, uint32_t R)
{
    if ((x0 > 0.0f && axis_leaves_root(x0, rootLo[0], rootHi[0])) || (x1 < 0.0f && axis_leaves_root(x1, rootLo[0], rootHi[0]))) return false;
    if ((y0 > 0.0f && axis_leaves_root(y0, rootLo[1], rootHi[1])) || (y1 < 0.0f && axis_leaves_root(y1, rootLo[1], rootHi[1]))) return false;
    if ((z0 > 0.0f && axis_leaves_root(z0, rootLo[2], rootHi[2])) || (z1 < 0.0f && axis_leaves_root(z1, rootLo[2], rootHi[2]))) return false;
    if (dm_box_straddles(x0, x1, y0, y1, z0, z1)) return true;          // straddles a centre plane
    const bool nx = x1 < 0.0f, ny = y1 < 0.0f, nz = z1 < 0.0f;
    const float xa = nx ? -x1 : x0, ya = ny ? -y1 : y0, za = nz ? -z1 : z0;
    const float rhoMin = __builtin_sqrtf((xa * xa + ya * ya) + za * za), nearMin = rhoMin * 0.999f;
    return dm_box_faces(x0, x1, y0, y1, z0, z1, R, [&](uint32_t face, uint32_t i0, uint32_t i1, uint32_t j0, uint32_t j1) {
        const uint32_t key = dm_mip_max(mip, R, face, i0, i1, j0, j1);
        return key != 0u && !(half_bits_to_float(key) < nearMin);
    });
}
// The longest list any ray of the box can look into (an upper bound: the same rectangles through the max-mip of the texels' entry
// counts, which lies behind the first mip in memory).  Not a matter of results: it decides which bricks a launch starts with --
// a brick is as long as its longest list, one in a hundred takes three to seven times the mean, and a launch whose last bricks
// are of that kind ends with a handful of waves on an idle GPU (k_plan_bricks).  Boxes that straddle a centre plane: 0.
DXV_HD uint32_t dm_box_max_count(float x0, float x1, float y0, float y1, float z0, float z1, const uint16_t* countMip, uint32_t R)
{
    if (dm_box_straddles(x0, x1, y0, y1, z0, z1)) return 0u;
    uint32_t most = 0;
    (void)dm_box_faces(x0, x1, y0, y1, z0, z1, R, [&](uint32_t face, uint32_t i0, uint32_t i1, uint32_t j0, uint32_t j1) {
        const uint32_t c = dm_mip_max(countMip, R, face, i0, i1, j0, j1);
        if (c > most) most = c;
        return false;
    });
    return most;
}
// memory of the two mips, the per-level "long list" words behind them (16 words) and the build's scratch behind those (dirmap.hip:
// k_dm_mip_tiles leaves (sum, non-empty cells) per tile and level there for k_dm_mip_top), 16-bit words
DXV_HD uint32_t dm_mip_partials_at(uint32_t R) { return (2u * dm_mip_words(R) + 16u + 3u) & ~3u; }
DXV_HD uint32_t dm_mip_buffer_words(uint32_t R)
{
    const uint32_t tile = R < 32u ? R : 32u, tiles = 6u * (R / tile) * (R / tile);
    return dm_mip_partials_at(R) + 4u * tiles * 8u;                   // (two 32-bit words per tile and level, up to 8 levels inside a tile)
}
// the level of the count mip whose cells are about the patch of texels a 4^3-voxel brick of an N^3 grid looks into
// (4 voxels of 2 / N at a typical distance of 0.7 from the centre: ~5.6 R / N texels across)
constexpr uint32_t kDmHeavyLevelMin = 2u;           // (the levels below have no "long list" word: k_dm_heavy_thresholds)
DXV_HD uint32_t dm_heavy_level(uint32_t R, uint32_t N)
{
    uint32_t l = 0;
    while ((1u << (l + 1u)) * N <= 8u * R && l + 1u < dm_mip_levels(R)) ++l;      // 2^l <= 8 R / N < 2^(l + 1):  512 / 512 -> 3, 512 / 256 -> 4, 512 / 1024 -> 2
    const uint32_t top = dm_mip_levels(R) - 1u;
    return l < kDmHeavyLevelMin ? (kDmHeavyLevelMin < top ? kDmHeavyLevelMin : top) : l;
}

// the hull of the voxel centres of brick (bx, by, bz) of 4 x 4 x 4 voxels in a partition's local brick grid (x0 <= x1 ...; y falls
// with iy, hlsl:49).  Local slice lz <-> global slice z0 + (lz >> zShift) * zPeriod + (lz & (zBlock - 1)) (zBlock == nz: a slab);
// the map is increasing, so the hull's z range comes from the brick's first and last slice.
DXV_HD void dm_brick_hull(uint32_t N, uint32_t nz, uint32_t z0, uint32_t zBlock, uint32_t zShift, uint32_t zPeriod, uint32_t bx, uint32_t by,
                          uint32_t bz, float& x0, float& x1, float& y0, float& y1, float& zlo, float& zhi)
{
    const uint32_t ix0 = bx * 4u, iy0 = by * 4u, lz0 = bz * 4u;
    const uint32_t ix1 = ix0 + 3u < N ? ix0 + 3u : N - 1u, iy1 = iy0 + 3u < N ? iy0 + 3u : N - 1u, lz1 = lz0 + 3u < nz ? lz0 + 3u : nz - 1u;
    const uint32_t iz0 = zBlock == nz ? z0 + lz0 : z0 + (lz0 >> zShift) * zPeriod + (lz0 & (zBlock - 1u));
    const uint32_t iz1 = zBlock == nz ? z0 + lz1 : z0 + (lz1 >> zShift) * zPeriod + (lz1 & (zBlock - 1u));
    ray_origin(N, ix0, iy1, iz0, x0, y0, zlo);
    ray_origin(N, ix1, iy0, iz1, x1, y1, zhi);
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
struct DirFootprint { float u0, u1, v0, v1, r0, r1, pad, px[3], py[3], pw[3]; bool hasTri; };

// The polygon a face sees of a triangle (its vertices in the face's own coordinates b, c, +-a) -> footprint.  FIXED = 3: the
// triangle itself, whole inside the face's frustum -- by far the common case; all loops have constant bounds then and the
// polygon stays in registers (the general case indexes it dynamically: scratch memory on the device).
template <int FIXED>
DXV_HD void dm_footprint_finish(const double (*poly)[3], int n, const double (*tri)[3], DirFootprint& out)
{
    const int m = FIXED ? FIXED : n;
    const double delta = (double)kDmDelta;
    double dmin = poly[0][2], rmax2 = 0.0;                        // (radii squared until the end: the root is monotone)
    for (int i = 0; i < m; ++i) {
        const double* q = poly[i];
        if (q[2] < dmin) dmin = q[2];
        const double r2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
        if (r2 > rmax2) rmax2 = r2;
    }
    const double full = kDmFrustum + 1.0 / 256.0;
    double u0 = -full, u1 = full, v0 = -full, v1 = full;
    out.hasTri = false; out.pad = 0.0f;
    for (int i = 0; i < 3; ++i) out.px[i] = out.py[i] = out.pw[i] = 0.0f;
    if (dmin >= 64.0 * delta) {
        u0 = v0 = 1e300; u1 = v1 = -1e300;
        for (int i = 0; i < m; ++i) {
            const double* q = poly[i];
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
        // the projected triangle itself, when all of it lies in front of the face plane: its edges, pushed out by the same
        // pad, bound the footprint too (the part cut off by the frustum's side planes only makes the polygon smaller)
        const double d0 = tri[0][2], d1 = tri[1][2], d2 = tri[2][2];           // (tri: the unclipped triangle in the face's coordinates)
        if (d0 >= 64.0 * delta && d1 >= 64.0 * delta && d2 >= 64.0 * delta) {
            const double dd[3] = {d0, d1, d2};
            for (int i = 0; i < 3; ++i) { out.px[i] = (float)(tri[i][0] / dd[i]); out.py[i] = (float)(tri[i][1] / dd[i]); out.pw[i] = (float)(1.0 / dd[i]); }
            out.hasTri = true;
            out.pad = (float)(pad + 1e-6);                      // (+ the rounding of the stored vertices)
        }
        if (u0 < -full) u0 = -full;
        if (v0 < -full) v0 = -full;
        if (u1 > full) u1 = full;
        if (v1 > full) v1 = full;
    }
    // radial range: the farthest point of a convex polygon is a vertex; the nearest is the foot of the perpendicular from
    // the centre onto its plane when that lies inside the polygon, else the nearest point of its boundary.  (A tight near
    // radius matters twice: it is what culls entries behind a hit, and the thickest entry of a texel decides how far
    // behind a hit the scan of that texel goes on.)
    double rmin2 = 1e300;
    for (int i = 0; i < m; ++i) {                                      // nearest point of every edge (covers degenerate polygons)
        const double* A = poly[i];
        const double* B = poly[(i + 1) % m];
        const double d[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
        const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        double w = dd > 0.0 ? -(A[0] * d[0] + A[1] * d[1] + A[2] * d[2]) / dd : 0.0;
        w = w < 0.0 ? 0.0 : w > 1.0 ? 1.0 : w;
        const double x = A[0] + w * d[0], y = A[1] + w * d[1], z = A[2] + w * d[2];
        const double dist2 = x * x + y * y + z * z;
        if (dist2 < rmin2) rmin2 = dist2;
    }
    if (m >= 3) {
        // plane through the polygon (its vertices are coplanar: clipped from one triangle), in the polygon's own coordinates
        const double* A = poly[0];
        double nx = 0.0, ny = 0.0, nz = 0.0;
        for (int i = 1; i + 1 < m; ++i) {                              // summed fan normals: robust for thin clips
            const double* B = poly[i];
            const double* C = poly[i + 1];
            const double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
            nx += e1[1] * e2[2] - e1[2] * e2[1]; ny += e1[2] * e2[0] - e1[0] * e2[2]; nz += e1[0] * e2[1] - e1[1] * e2[0];
        }
        const double nn = nx * nx + ny * ny + nz * nz;
        if (nn > 1e-60) {
            const double k = (nx * A[0] + ny * A[1] + nz * A[2]) / nn;
            const double P[3] = {k * nx, k * ny, k * nz};               // foot of the perpendicular from the centre
            bool inside = true;
            for (int i = 0; i < m && inside; ++i) {
                const double* E = poly[i];
                const double* F = poly[(i + 1) % m];
                const double d[3] = {F[0] - E[0], F[1] - E[1], F[2] - E[2]}, q[3] = {P[0] - E[0], P[1] - E[1], P[2] - E[2]};
                const double cx = d[1] * q[2] - d[2] * q[1], cy = d[2] * q[0] - d[0] * q[2], cz = d[0] * q[1] - d[1] * q[0];
                // same turn as the polygon's own orientation (the fan normal).  The clip leaves coincident vertices behind
                // (an edge through a frustum corner is cut twice at one point): such an edge's product is rounding noise of
                // either sign, so the test forgives 1e-10 of the operands' scale.  Erring towards "inside" is the safe side:
                // the distance to the plane never exceeds the distance to any point of the polygon.
                const double qq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
                inside = cx * nx + cy * ny + cz * nz >= -1e-10 * (1.0 + rmax2) * __builtin_sqrt(qq * nn);
            }
            const double dist2 = k * k * nn;
            if (inside && dist2 < rmin2) rmin2 = dist2;
        }
    }
    double rmin = __builtin_sqrt(rmin2), rmax = __builtin_sqrt(rmax2);
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
}


DXV_HD bool dm_footprint(const TriPos& tp, uint32_t face, DirFootprint& out)
{
    const uint32_t a = face >> 1, b = (a + 1u) % 3u, c = (a + 2u) % 3u;
    const double s = (face & 1u) ? -1.0 : 1.0;
    // the triangle in the face's coordinates (b, c, +-a); the axes are picked by selects, not by indexing (no scratch memory)
    const float X[3] = {tp.v0.x, tp.v1.x, tp.v2.x}, Y[3] = {tp.v0.y, tp.v1.y, tp.v2.y}, Z[3] = {tp.v0.z, tp.v1.z, tp.v2.z};
    double tri[3][3];
    for (int i = 0; i < 3; ++i) {
        tri[i][0] = b == 0u ? X[i] : b == 1u ? Y[i] : Z[i];
        tri[i][1] = c == 0u ? X[i] : c == 1u ? Y[i] : Z[i];
        tri[i][2] = s * (double)(a == 0u ? X[i] : a == 1u ? Y[i] : Z[i]);
    }
    const double delta = (double)kDmDelta;
    // the four side planes of the widened frustum, pushed out by the dilation: kF * d +- b + 2 delta >= 0
    const double planes[4][2] = {{1.0, 0.0}, {-1.0, 0.0}, {0.0, 1.0}, {0.0, -1.0}};
    // most (triangle, face) pairs end here: all three vertices outside one side plane (what the clip
    // below would find, without its arrays)
    bool allIn = true;
    for (int pl = 0; pl < 4; ++pl) {
        bool anyIn = false;
        for (int i = 0; i < 3; ++i) {
            const bool in = kDmFrustum * tri[i][2] + planes[pl][0] * tri[i][0] + planes[pl][1] * tri[i][1] + 2.0 * delta >= 0.0;
            anyIn = anyIn || in;
            allIn = allIn && in;
        }
        if (!anyIn) return false;
    }
    if (allIn) {                                                        // nothing to clip: the polygon is the triangle
        dm_footprint_finish<3>(tri, 3, tri, out);
        return true;
    }
    double poly[2][10][3];
    int n = 3, cur = 0;
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) poly[0][i][k] = tri[i][k];
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
    dm_footprint_finish<0>(poly[cur], n, tri, out);
    return true;
}

// the record of a (triangle, face) pair; u0 > u1 when there is no footprint
DXV_HD DirRecord dm_record(const DirFootprint& f)
{
    DirRecord r;
    r.u0 = f.u0; r.u1 = f.u1; r.v0 = f.v0; r.v1 = f.v1;
    r.rr = (uint32_t)half_down(f.r0) | ((uint32_t)half_up(f.r1) << 16);
    r.hasTri = f.hasTri ? 1u : 0u; r.pad = f.pad;
    for (int i = 0; i < 3; ++i) { r.px[i] = f.px[i]; r.py[i] = f.py[i]; r.pw[i] = f.pw[i]; }
    return r;
}
DXV_HD DirRecord dm_record(const TriPos& tp, uint32_t face)
{
    DirFootprint f;
    if (!dm_footprint(tp, face, f)) {
        DirRecord r;
        r.u0 = 1.0f; r.u1 = 0.0f; r.v0 = 1.0f; r.v1 = 0.0f; r.rr = 0u; r.hasTri = 0u; r.pad = 0.0f;
        for (int i = 0; i < 3; ++i) r.px[i] = r.py[i] = r.pw[i] = 0.0f;
        return r;
    }
    return dm_record(f);
}

// texel rectangle [i0, i1] x [j0, j1] of a record (false: none)
DXV_HD bool dm_rect(const DirRecord& e, uint32_t R, uint32_t& i0, uint32_t& i1, uint32_t& j0, uint32_t& j1)
{
    if (e.u0 > e.u1) return false;
    if (e.u1 < -1.0f || e.u0 > 1.0f || e.v1 < -1.0f || e.v0 > 1.0f) return false;      // rays only have |u|, |v| <= 1
    i0 = dm_texel(e.u0, R); i1 = dm_texel(e.u1, R); j0 = dm_texel(e.v0, R); j1 = dm_texel(e.v1, R);
    return true;
}

// Radial range of a record INSIDE texel (i, j), as halfs rounded outward: a triangle that spans several texels (every triangle of
// a coarse mesh, of a soup) is radially much thinner over one of them than as a whole, and the radial range is what culls an entry
// before its triangle is fetched, what orders a texel's list and what lets a scan stop behind a hit.
// 1 / depth along the face axis is LINEAR in the face coordinates (u, v) over the triangle's plane: w(u, v) = pw0 + gu (u - px0)
// + gv (v - py0), its extremes over a rectangle lie at the corners, and a point of the plane in direction (u, v) has radius
// sqrt(1 + u^2 + v^2) / w(u, v).  The rectangle is the texel cut by the record's box, widened by the record's pad (the rays of the
// texel, the dilated triangle's outline); the plane extrapolated beyond the triangle only widens the range.  Points within the
// dilation delta of the plane lie within delta rho / |d| of it along a ray (d: the plane's distance from the centre, 1 / d^2 =
// w(0, 0)^2 + gu^2 + gv^2): taken twice, plus the whole footprint's own margins.  The result is cut with the whole footprint's
// range; records without a projected triangle, seen edge-on, or whose plane passes the centre keep that range.
// Which texels of a record's rectangle get an entry at all.  The rectangle is the bounding box of the footprint: a triangle fills
// half of its box, and a quarter to a third of the (triangle, texel) pairs of a mesh are texels that lie entirely outside one
// edge of the projected, dilated triangle -- entries no ray could ever select (measured on the host: 28 % bunny, 35 % dragon, 38 % a
// soup, 11 % a mesh of texel-sized triangles).  dm_texel_test makes the three edge functions of a record once (inside edge k <=>
// nx u + ny v + c >= 0, the dilation `pad` and every rounding of this single-precision arithmetic on the safe side of c);
// dm_texel_outside says whether a texel's rectangle (widened by more than the rays' own rounding) lies wholly outside one of them.
// Counting pass, key pass and the host's replica call the same two functions.  Records without a projected triangle, seen
// (nearly) edge-on, or with more than kDmTexelTestMax texels keep their whole rectangle.
constexpr uint32_t kDmTexelTestMax = 1024u;          // (beyond: a triangle close to the centre -- rare, and one thread walks the rectangle in the counting pass)
struct DirTexelTest { float nx[3], ny[3], c[3]; bool on; };
DXV_HD DirTexelTest dm_texel_test(const DirRecord& rec, uint32_t texels)
{
    DirTexelTest t;
    t.on = false;
    for (int k = 0; k < 3; ++k) { t.nx[k] = 0.0f; t.ny[k] = 0.0f; t.c[k] = 0.0f; }
    if (!(rec.hasTri & 1u) || texels < 4u || texels > kDmTexelTestMax) return t;   // (a strip of two or three texels is touched in all of them)
    const float ax = rec.px[1] - rec.px[0], ay = rec.py[1] - rec.py[0], bx = rec.px[2] - rec.px[0], by = rec.py[2] - rec.py[0];
    const float area2 = ax * by - ay * bx;
    const float scale = (ax * ax + ay * ay) + (bx * bx + by * by);
    if (!(__builtin_fabsf(area2) > 1e-3f * scale) || !(scale > 1e-24f)) return t;
    const float sgn = area2 > 0.0f ? 1.0f : -1.0f;
    for (int k = 0; k < 3; ++k) {
        const int k1 = k == 2 ? 0 : k + 1;
        const float ex = rec.px[k1] - rec.px[k], ey = rec.py[k1] - rec.py[k];
        const float nx = -ey * sgn, ny = ex * sgn;                      // inward normal (not unit: |n| <= |nx| + |ny| pays for that)
        const float l1 = __builtin_fabsf(nx) + __builtin_fabsf(ny);
        const float px = rec.px[k] * nx, py = rec.py[k] * ny;
        t.nx[k] = nx; t.ny[k] = ny;
        t.c[k] = -(px + py) + (rec.pad + 1e-5f) * l1 + 1e-6f * (__builtin_fabsf(px) + __builtin_fabsf(py)) + 1e-30f;
    }
    t.on = true;
    return t;
}
DXV_HD bool dm_texel_outside(const DirTexelTest& t, uint32_t R, uint32_t i, uint32_t j)
{
    if (!t.on) return false;
    const float inv = 2.0f / (float)R;                                  // (R a power of two: exact)
    const float u0 = (float)i * inv - 1.0f - 4e-6f, u1 = (float)(i + 1u) * inv - 1.0f + 4e-6f;
    const float v0 = (float)j * inv - 1.0f - 4e-6f, v1 = (float)(j + 1u) * inv - 1.0f + 4e-6f;
    for (int k = 0; k < 3; ++k) {
        const float pu = t.nx[k] > 0.0f ? u1 : u0, pv = t.ny[k] > 0.0f ? v1 : v0;       // the corner farthest inside edge k
        const float a = t.nx[k] * pu, b = t.ny[k] * pv;
        if ((a + b) + t.c[k] + 1e-6f * (__builtin_fabsf(a) + __builtin_fabsf(b)) < 0.0f) return true;
    }
    return false;
}

constexpr uint32_t kDmLocalRadialFrom = 12u;         // footprints of more texels than this get per-texel radial ranges
// hasTri bit 1: the record's footprint covers more than kDmLocalRadialFrom texels of the map the lists are built on (`texels`: the
// area of its rectangle there) and has a projected triangle -- set by whoever makes the record for a map (k_dm_records, hostcheck)
DXV_HD void dm_record_on_map(DirRecord& rec, uint32_t texels)
{
    if ((rec.hasTri & 1u) && texels > kDmLocalRadialFrom && texels <= kDmTexelTestMax) rec.hasTri |= 2u;
}
DXV_HD void dm_local_radial(const DirRecord& rec, uint32_t R, uint32_t i, uint32_t j, uint32_t& r0h, uint32_t& r1h)
{
    r0h = rec.rr & 0xffffu; r1h = rec.rr >> 16;
    // (a footprint of a few texels is hardly thinner over one of them: the whole range, for nothing -- every record of a mesh whose
    // triangles are as small as the texels ends here; the flag is set where the record's rectangle on this map is known: dm_record_on_map)
    if (!(rec.hasTri & 2u)) return;
    // (single precision throughout -- this runs once per entry in two passes of the build -- with every rounding paid for below: the fit
    // of a triangle seen within 1e-3 of edge-on is not trusted at all, the fitted w carries 2e-3 of its own variation over the
    // rectangle and 1e-5 of its size, the radii 3e-5 of theirs.  Plain IEEE operations, no contraction: host and device agree.)
    const float ax = rec.px[1] - rec.px[0], ay = rec.py[1] - rec.py[0], aw = rec.pw[1] - rec.pw[0];
    const float bx = rec.px[2] - rec.px[0], by = rec.py[2] - rec.py[0], bw = rec.pw[2] - rec.pw[0];
    const float det = ax * by - ay * bx;
    const float scale = (ax * ax + ay * ay) + (bx * bx + by * by);
    if (!(__builtin_fabsf(det) > 1e-3f * scale) || !(scale > 1e-24f)) return;   // seen (nearly) edge-on, or degenerate: no plane to speak of
    const float gu = (aw * by - bw * ay) / det, gv = (bw * ax - aw * bx) / det;
    const float w00 = rec.pw[0] - gu * rec.px[0] - gv * rec.py[0];
    const float invd = __builtin_sqrtf(w00 * w00 + gu * gu + gv * gv);
    if (!(invd * kDmDelta < 0.125f)) return;                                    // the plane passes (nearly) through the centre
    const float eps = rec.pad + 4e-6f;
    float ua = 2.0f * (float)i / (float)R - 1.0f, ub = 2.0f * (float)(i + 1u) / (float)R - 1.0f;
    float va = 2.0f * (float)j / (float)R - 1.0f, vb = 2.0f * (float)(j + 1u) / (float)R - 1.0f;
    if (rec.u0 > ua) ua = rec.u0;
    if (rec.u1 < ub) ub = rec.u1;
    if (rec.v0 > va) va = rec.v0;
    if (rec.v1 < vb) vb = rec.v1;
    ua -= eps; ub += eps; va -= eps; vb += eps;
    if (!(ua <= ub && va <= vb)) return;
    const float w0 = w00 + gu * ua + gv * va, w1 = w00 + gu * ub + gv * va, w2 = w00 + gu * ua + gv * vb, w3 = w00 + gu * ub + gv * vb;
    float wmin = w0 < w1 ? w0 : w1, wmax = w0 > w1 ? w0 : w1;
    if (w2 < wmin) wmin = w2;
    if (w3 < wmin) wmin = w3;
    if (w2 > wmax) wmax = w2;
    if (w3 > wmax) wmax = w3;
    const float terms = __builtin_fabsf(w00) + __builtin_fabsf(gu) * (__builtin_fabsf(ua) > __builtin_fabsf(ub) ? __builtin_fabsf(ua) : __builtin_fabsf(ub)) +
                        __builtin_fabsf(gv) * (__builtin_fabsf(va) > __builtin_fabsf(vb) ? __builtin_fabsf(va) : __builtin_fabsf(vb));
    const float slack = 2e-3f * (wmax - wmin) + 1e-5f * terms;
    wmin -= slack; wmax += slack;
    const float ulo = ua > 0.0f ? ua : ub < 0.0f ? -ub : 0.0f, vlo = va > 0.0f ? va : vb < 0.0f ? -vb : 0.0f;
    const float uhi = -ua > ub ? -ua : ub, vhi = -va > vb ? -va : vb;
    const float thick = 2.0f * kDmDelta * invd + 3e-5f;
    if (wmax > 0.0f) {
        float rmin = __builtin_sqrtf(1.0f + ulo * ulo + vlo * vlo) / wmax;
        rmin = rmin * (1.0f - thick) - 4.0f * kDmDelta;
        if (rmin > 0.0f) { const uint32_t h = half_down(rmin); if (h > r0h) r0h = h; }
    }
    if (wmin > 1e-6f) {
        float rmax = __builtin_sqrtf(1.0f + uhi * uhi + vhi * vhi) / wmin;
        rmax = rmax * (1.0f + thick) + 4.0f * kDmDelta;
        if (rmax < 60000.0f) { const uint32_t h = half_up(rmax); if (h < r1h) r1h = h; }
    }
    if (r0h > r1h) r0h = r1h;                                                   // (cannot happen for a range that holds a point; keeps the words ordered)
}

// The entry of a record in texel (i, j) of its rectangle: the box cut to the texel in local cells, the radial word, and
// the edge of the projected triangle that removes most of that box.
// Edge arithmetic (builder side, double): inside the dilated triangle means (p - P_k) . n_k + pad >= 0 for the unit inward
// normal n_k of every edge k.  With texel-local coordinates X, Y (u = 2 i / R - 1 + X / (64 R)) that is A X + B Y + C >= 0;
// scaled so that the larger of |A|, |B| is 127 and rounded to bytes a, b.  For a ray in integer cell (x, y), i.e. anywhere
// in [x, x + 1) x [y, y + 1), a x + b y differs from the scaled A X + B Y by at most |A| + |B| (position inside the cell)
// + 127 (rounding of a and b, x, y <= 127) + a little for the ray's own float arithmetic: c takes all of that, so the
// byte test passes wherever the real one does (it gives away two to three cells of 128, 2 % of a texel).
DXV_HD DirEntry dm_local_entry(const DirRecord& rec, uint32_t R, uint32_t i, uint32_t j, uint32_t tri)
{
    DirEntry e;
    e.tri = tri;
    uint32_t r0h, r1h;
    dm_local_radial(rec, R, i, j, r0h, r1h);
    e.rr = ((0x7fffu - r0h) | (r1h << 16)) | 0x80008000u;
    uint32_t t0, c0, t1, c1, x0, x1, y0, y1;
    dm_local(rec.u0, R, t0, c0); dm_local(rec.u1, R, t1, c1);
    x0 = t0 < i ? 0u : c0; x1 = t1 > i ? kDmCells - 1u : c1;
    dm_local(rec.v0, R, t0, c0); dm_local(rec.v1, R, t1, c1);
    y0 = t0 < j ? 0u : c0; y1 = t1 > j ? kDmCells - 1u : c1;
    e.box = x0 | (y0 << 8) | ((kDmCells - 1u - x1) << 16) | ((kDmCells - 1u - y1) << 24);
    e.edge = 0u;
    if (!(rec.hasTri & 1u)) return e;
    const double area2 = ((double)rec.px[1] - rec.px[0]) * ((double)rec.py[2] - rec.py[0]) - ((double)rec.py[1] - rec.py[0]) * ((double)rec.px[2] - rec.px[0]);
    if (!(__builtin_fabs(area2) > 1e-14)) return e;                     // seen edge-on: the box has to do
    const double sgn = area2 > 0.0 ? 1.0 : -1.0;
    const double ou = 2.0 * i / R - 1.0, ov = 2.0 * j / R - 1.0, perCell = 1.0 / (64.0 * R);
    int bestCut = 0;
    for (int k = 0; k < 3; ++k) {
        const double ex = (double)rec.px[(k + 1) % 3] - rec.px[k], ey = (double)rec.py[(k + 1) % 3] - rec.py[k];
        const double len2 = ex * ex + ey * ey;
        if (!(len2 > 1e-24)) continue;
        // inward normal, unit to 3e-7 (single-precision root and reciprocal: the test is homogeneous in the normal except for
        // `pad`, which carries 10 % of slack)
        const double rl = (double)(1.0f / __builtin_sqrtf((float)len2)) * sgn;
        const double nx = -ey * rl, ny = ex * rl;
        const double A = nx * perCell, B = ny * perCell;
        const double C = (ou - rec.px[k]) * nx + (ov - rec.py[k]) * ny + rec.pad;
        const double big = __builtin_fabs(A) > __builtin_fabs(B) ? __builtin_fabs(A) : __builtin_fabs(B);
        const double sc = (double)(127.0f / (float)big);                // (one scale for A, B and C: its last bits do not matter)
        const double ka = A * sc, kb = B * sc;
        const int a = (int)__builtin_floor(ka + 0.5), b = (int)__builtin_floor(kb + 0.5);
        double cd = __builtin_ceil(C * sc + __builtin_fabs(ka) + __builtin_fabs(kb) + 127.0 + 13.0);
        if (!(cd < 16256.0)) continue;                                   // this edge cuts nothing representable off the texel
        if (cd < -16256.0) cd = -16256.0;                               // (more permissive: safe)
        const int c = (int)cd;
        const int c1b = c / 127, c2b = c - 127 * c1b;                   // truncation: |c2b| < 127
        // how much of the box does it remove?  Nothing when the box's corner farthest outside is still inside; else 3 x 3 samples
        const int worst = a * (int)(a < 0 ? x1 : x0) + b * (int)(b < 0 ? y1 : y0) + c;
        if (worst >= 0) continue;
        int cut = 1;
        for (int sy = 0; sy < 3; ++sy)
            for (int sx = 0; sx < 3; ++sx) {
                const int x = (int)x0 + (int)((x1 - x0) * (2 * sx + 1) / 6), y = (int)y0 + (int)((y1 - y0) * (2 * sy + 1) / 6);
                if (a * x + b * y + c < 0) ++cut;
            }
        if (cut > bestCut) {
            bestCut = cut;
            e.edge = (uint32_t)(uint8_t)(int8_t)a | ((uint32_t)(uint8_t)(int8_t)b << 8) | ((uint32_t)(uint8_t)(int8_t)c1b << 16) | ((uint32_t)(uint8_t)(int8_t)c2b << 24);
        }
    }
    return e;
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

constexpr uint32_t kDmCoopLanes = 2u, kDmCoopMin = 24u;     // the cooperative scan of a lone lane's long list (trace_reference_dm_from)
// Diagnostic build (-DDXV_PHASE_TIMES, tools/phase_times.py): where a brick's TIME goes -- 100 MHz ticks between stamps of the wave's own
// instruction stream, summed over all bricks per phase: [0] ray set-up and the texel's cell, [1] start search, [2] scan rounds,
// [3] direction / shear at a flush, [4] triangle rounds, [5] predicate and stores; [6] bricks, [7] scan rounds counted, [8] triangle
// rounds counted, [9] flushes.  A stall on a load is charged to the phase that first uses the value.
#if defined(DXV_PHASE_TIMES) && defined(__HIPCC__)
constexpr uint32_t kPhaseSlots = 1u << 16;                   // (a workgroup adds to the slot of its number: no two waves in flight share a word for long)
static __device__ unsigned long long g_dxvPhase[kPhaseSlots * 16u];
#endif
#if defined(DXV_PHASE_TIMES) && defined(__HIP_DEVICE_COMPILE__)
#define DXV_PHASE_BEGIN() unsigned long long tPhase_ = __builtin_amdgcn_s_memrealtime(), ph_[10] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}
#define DXV_PHASE(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); ph_[k] += now_ - tPhase_; tPhase_ = now_; } while (0)
#define DXV_COUNT(k) do { ph_[k] += 1ull; } while (0)
#define DXV_PHASE_END() do { if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) { unsigned long long* slot_ = g_dxvPhase + (size_t)(blockIdx.x & (kPhaseSlots - 1u)) * 16u; \
    for (int k_ = 1; k_ < 10; ++k_) if (ph_[k_]) atomicAdd(slot_ + k_, ph_[k_]); } } while (0)
#else
#define DXV_PHASE_BEGIN() do { } while (0)
#define DXV_PHASE(k) do { } while (0)
#define DXV_COUNT(k) do { } while (0)
#define DXV_PHASE_END() do { } while (0)
#endif
// closest hit of the reference rule through the lists
// Two steps like the postponed-leaf walks: scanning entries is short and cheap, the triangle step is
// long, so the triangles an entry scan selects are queued in the thread's LDS column (cap entries)
// and tested when some lane's queue is full or every lane has finished scanning -- all lanes with
// work test together.
// ABL (timing-only builds, tools/ablate.py; 0 in every shipped path): 8 = stop before the texel lookup, 1 = stop after it,
// 2 = scan the entries but test no triangle, 16 = the scan's loads wave-uniform (with 2: 18), 32 / 64 = two / four more loads per scan
// round with their results unused: what one of the scan's load instructions costs
// (split at the ray's first step: the work-queue kernel makes that step for all 64 lanes of a brick at once and collects the brick
// it asked for in advance behind it -- k_voxelize_queue, traverse.hip)
// HITLDS: the closest hit's V, W, det and index live in column words cap .. cap + 3 (leaf_reference_deferred_lds); best.t and best.leaf are
// all of `best` that is meaningful then (best.leaf == -1: miss), bestDet is untouched.
// HITLDS 2: nothing but best.t and best.leaf is kept at all (leaf_reference_min, shade_reference_again).
template <class Stack, int ABL = 0, int HITLDS = 0>
DXV_HD void trace_reference_dm_from(Ray& r, const DirMapView& dm, const DirRayStart& start, const TriPos* tris, const Stack& stk, int cap, Hit& best,
                                    float& bestDet)
{
    best.t = kTMax; best.b1 = 0.0f; best.b2 = 0.0f; best.k = 0xffffffffu; best.leaf = -1;
    if (ABL & 8) return;
    DXV_PHASE_BEGIN();
    const DirCell cell = start.cell;
    const uint32_t cx = start.cx, cy = start.cy;
    const float rho = start.rho, near = start.near;
    // entries wholly nearer the centre than the ray's start (r1 < near: t < 0) come first: skip them --
    // all of them at once for a ray that starts beyond the texel's last triangle
    uint32_t end = cell.begin + cell.count;
    uint32_t i = cell.begin, hi = end;
    if (!start.live) i = hi;
    if (ABL & 1) { if (i == 0xffffffffu) best.k = 0u; return; }
    if (hi - i > 8u) {                                                  // the first two steps of the search from the texel's own words
        const uint32_t mid = i + ((hi - i) >> 1);
        const bool right = half_bits_to_float(cell.q2) < near;
        const float q = half_bits_to_float(right ? cell.q3 : cell.q1);
        if (right) i = mid + 1u; else hi = mid;
        if (hi - i > 8u) {
            const uint32_t mid2 = i + ((hi - i) >> 1);
            if (q < near) i = mid2 + 1u; else hi = mid2;
        }
    }
    while (hi - i > 8u) {
        const uint32_t mid = i + ((hi - i) >> 1);
        if (dm_entry_r1(dm.entries[mid]) < near) i = mid + 1u; else hi = mid;
    }
    DXV_PHASE(1);
    int qn = 0;
    bestDet = 1.0f;                                                     // divisor of the closest hit's barycentrics (the caller's finish_hit)
    const float step = dm_stop_step(half_bits_to_float(cell.thick));
    DirRayLocal loc = dm_ray_local(cx, cy);
    float bound = (rho + best.t) * 1.001f + 1e-4f;                      // how far out an entry may start and still matter: behind the closest hit so far
    uint32_t rc = dm_radial_word(near, bound);                          // radial cut: r1 >= near, r0 not beyond `bound`
    for (;;) {
        bool coop = false;                                                  // this round was a cooperative one (wave-uniform; the host: never)
#if defined(__HIP_DEVICE_COMPILE__)
        // A LONG list scanned by a LONE lane: a brick is as long as its longest list (a ray that finds no hit reads its texel's list to
        // the end: hundreds of entries in a deep fold of a real mesh, four per round, while the wave's other lanes have long
        // finished: 40 - 80 us against a median brick of 10, and a short launch cannot end before its longest brick).  When at most
        // kDmCoopLanes lanes are still scanning and the first of them has kDmCoopMin entries or more ahead, the WHOLE wave scans that
        // lane's list, one entry per lane and round (one coalesced kilobyte), with that ray's words broadcast through scalar
        // registers; what passes goes into the ray's own queue, as if it had scanned alone.  Same entries, same integer test, same
        // stop rule -- entries are only LOOKED AT in another order, and the closest hit does not depend on the order of tests.
        {
            const uint64_t sm = __builtin_amdgcn_ballot_w64(i < end);
            if (dm.coop && sm != 0ull && __builtin_popcountll(sm) <= (int)kDmCoopLanes) {
                const int L = __builtin_ctzll(sm);
                const uint32_t iL = (uint32_t)__builtin_amdgcn_readlane((int)i, L), endL = (uint32_t)__builtin_amdgcn_readlane((int)end, L);
                if (endL - iL >= kDmCoopMin) {
                    const DirRayLocal locL{(uint32_t)__builtin_amdgcn_readlane((int)loc.q, L), (uint32_t)__builtin_amdgcn_readlane((int)loc.p, L)};
                    const uint32_t rcL = (uint32_t)__builtin_amdgcn_readlane((int)rc, L);
                    const float stepL = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, step), L));
                    const float boundL = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bound), L));
                    // (lanes of this wave that take part: a brick at the grid's end has left some behind)
                    const uint64_t act = __builtin_amdgcn_ballot_w64(true);
                    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                    const uint32_t rank = (uint32_t)__builtin_popcountll(act & ((1ull << lane) - 1ull)), width = (uint32_t)__builtin_popcountll(act);
                    const uint32_t at = iL + rank;
                    const bool have = at < endL;
                    const DirEntry e = dm.entries[have ? at : endL - 1u];
                    const uint64_t stop = __builtin_amdgcn_ballot_w64(have && dm_stop_radius(e, stepL) > boundL);
                    // the first entry (in list order) at which a lone scan would have stopped: nothing from there on matters
                    uint32_t stopRank = 64u;
                    if (stop) stopRank = (uint32_t)__builtin_popcountll(act & ((1ull << __builtin_ctzll(stop)) - 1ull));
                    uint64_t pm = __builtin_amdgcn_ballot_w64(have && rank < stopRank && dm_local_pass(e, locL, rcL));
                    int qnL = __builtin_amdgcn_readlane(qn, L);
                    const uint32_t triE = dm_entry_tri(e);
                    uint32_t next = stop ? endL : (iL + width < endL ? iL + width : endL);
                    while (pm) {
                        const int src = __builtin_ctzll(pm);
                        if (2 * (qnL + 1) > cap) {                           // the ray's queue is full: go on behind the last entry taken
                            next = iL + (uint32_t)__builtin_popcountll(act & ((1ull << src) - 1ull));
                            break;
                        }
                        pm &= pm - 1ull;
                        const int32_t t = __builtin_amdgcn_readlane((int)triE, src), rr = __builtin_amdgcn_readlane((int)e.rr, src);
                        if ((int)lane == L) { stk.put(2 * qn, t); stk.put(2 * qn + 1, rr); ++qn; }
                        ++qnL;
                    }
                    if ((int)lane == L) i = next;
                    coop = true;                                            // (the ray's queue may now hold up to `cap` words: no lane scans on its own in this round)
                }
            }
        }
#endif
        // four entries per round, all four loads in flight before the first is looked at (two per round:
        // +13 % on the 1 M-triangle scene, one: +40 %).  One address, four offsets: entries behind the end of the list are
        // loaded and not looked at (the next texel's, or the three spare ones behind the last list).
        if (!coop && i < end) {
            const uint32_t last = end - 1u;
            // (the third and fourth load are skipped when no lane of the wave has that many entries left:
            // -5 % on the 1 M-triangle scene; voting on the second one as well: +7 %)
            const bool wide = wave_any(i + 2u <= last);
            const DirEntry* p = dm.entries + i;
#if defined(DXV_ABLATE) && defined(__HIP_DEVICE_COMPILE__)
            // (timing only, 16: every lane loads the FIRST active lane's entries -- one line access per load instead of one per
            // lane, instruction count unchanged: what the launch would gain if the scan's loads were wave-uniform)
            if (ABL & 16) p = dm.entries + __builtin_amdgcn_readfirstlane(i);
#endif
            const DirEntry e0 = p[0], e1 = p[1];
            DirEntry e2, e3;
            if (wide && i + 2u <= last) { e2 = p[2]; e3 = p[3]; }       // (lanes whose list ends here issue no access for what they would not look at)
#if defined(DXV_ABLATE) && defined(__HIP_DEVICE_COMPILE__)
            // (timing only, 32 / 64: two / four MORE 16-byte loads per round, of the entries the next round loads anyway, results
            // unused: the slope is what one of the scan's load instructions costs)
            if (ABL & (32 | 64)) {
                const DirEntry x0 = p[4], x1 = p[5];
                asm volatile("" :: "v"(x0.box), "v"(x1.box));
                if (ABL & 64) { const DirEntry x2 = p[6], x3 = p[7]; asm volatile("" :: "v"(x2.box), "v"(x3.box)); }
            }
#endif
            // The list is sorted by far radius and every entry knows (in 63rds of the texel's thickest entry) how far behind
            // its far radius the earliest start of any LATER entry lies: once that point is beyond the closest hit so far,
            // this entry and everything behind it start beyond the hit.  (Surface meshes have short lists and gain
            // little; in a deep soup a ray stops after the first few of hundreds of entries.)
            if (dm_stop_radius(e0, step) > bound) i = end;
            else {
                // (a queued item is two words: the triangle, and the entry's radial word for the second look below)
                if (dm_local_pass(e0, loc, rc)) { stk.put(2 * qn, (int32_t)dm_entry_tri(e0)); stk.put(2 * qn + 1, (int32_t)e0.rr); ++qn; }
                if (i + 1u <= last && dm_local_pass(e1, loc, rc)) { stk.put(2 * qn, (int32_t)dm_entry_tri(e1)); stk.put(2 * qn + 1, (int32_t)e1.rr); ++qn; }
                if (wide) {
                    if (i + 2u <= last && dm_local_pass(e2, loc, rc)) { stk.put(2 * qn, (int32_t)dm_entry_tri(e2)); stk.put(2 * qn + 1, (int32_t)e2.rr); ++qn; }
                    if (i + 3u <= last && dm_local_pass(e3, loc, rc)) { stk.put(2 * qn, (int32_t)dm_entry_tri(e3)); stk.put(2 * qn + 1, (int32_t)e3.rr); ++qn; }
                }
                i += 4u;
            }
        }
        const bool scanning = wave_any(i < end);
        DXV_PHASE(2); DXV_COUNT(7);
        if (scanning && !wave_any(2 * (qn + 4) > cap)) continue;
        DXV_COUNT(9);
        if (ABL & 2) { if (qn > 100) best.k = 0u; }
        else {
            // direction, 1 / d, -o / d (hlsl:52 and the slab constants): only now, after the scan of the short lists of a
            // surface mesh is over -- two waves in five never get here, and the scan runs with a dozen registers less
            finish_ray_reference(r, rho);                               // (|o| is the ray's start radius: the same expression, dm_ray_point)
            ray_shear_finished(r);                                      // (here, not at the first triangle: the direction's registers are free through the tests)
            DXV_PHASE(3);
            // Every lane takes its queued triangles in turn, but looks at an item's near radius once more first: what starts
            // beyond a hit found since it was queued is dropped unfetched, so a round is one triangle for every lane that
            // still has a live item (the most loaded lane of a wave decides how many rounds there are).
            int k = 0;
            for (;;) {
                while (k < qn && (((uint32_t)stk.get(2 * k + 1) - rc) & 0x00008000u) == 0u) ++k;      // r0 beyond the closest hit so far
                if (!wave_any(k < qn)) break;
                if (k < qn) {
                    if (HITLDS == 2) leaf_reference_min(r, tris, stk.get(2 * k), best.t, best.leaf);
                    else if (HITLDS == 1) leaf_reference_deferred_lds(r, tris, stk.get(2 * k), best.t, best.leaf, stk, cap);
                    else leaf_reference_deferred(r, tris, stk.get(2 * k), best, bestDet);
                    bound = (rho + best.t) * 1.001f + 1e-4f;
                    rc = dm_radial_word(near, bound);
                    ++k;
                }
                DXV_COUNT(8);
            }
            DXV_PHASE(4);
        }
        qn = 0;
        if (!scanning) break;
    }
    DXV_PHASE_END();
}

// Row lists of the parity rule (dirmap.hip): the texels of the R x R grid over the (y, z) plane that a triangle's padded box --
// the box parity_row_setup tests a row against -- reaches; a row's texel is (dm_texel(oy), dm_texel(oz)), monotone in the
// coordinate, so a row inside the box lies in a texel of the rectangle.
DXV_HD void pl_rect(const TriPos& tp, uint32_t R, uint32_t& j0, uint32_t& j1, uint32_t& k0, uint32_t& k1)
{
    float lo[3], hi[3];
    tri_box(tp.v0, tp.v1, tp.v2, lo, hi);
    j0 = dm_texel(lo[1], R); j1 = dm_texel(hi[1], R); k0 = dm_texel(lo[2], R); k1 = dm_texel(hi[2], R);
}

template <class Stack, int ABL = 0>
DXV_HD void trace_reference_dm(Ray& r, const DirMapView& dm, const TriPos* tris, const Stack& stk, int cap, Hit& best, float& bestDet)
{
    if (ABL & 8) { best.t = kTMax; best.b1 = 0.0f; best.b2 = 0.0f; best.k = 0xffffffffu; best.leaf = -1; return; }
    const DirRayStart start = dm_ray_start(r.ox, r.oy, r.oz, dm);
    trace_reference_dm_from<Stack, ABL>(r, dm, start, tris, stk, cap, best, bestDet);
}

template <class Stack, int ABL>
DXV_HD void trace_reference_lists(Ray& r, const SceneView& sc, const Stack& stk, int cap, Hit& best, float& bestDet)
{
    const DirMapView dm{static_cast<const DirCell*>(sc.dmCells), static_cast<const DirEntry*>(sc.dmEntries), sc.dmR, sc.dmCoop};
    trace_reference_dm<Stack, ABL>(r, dm, sc.triPos, stk, cap, best, bestDet);
}

} // namespace dxv
