// hostcheck.cpp -- TEST-ONLY harness: drives the product's own __host__ __device__ code
// (dxrvoxelizer_amd/csrc/dxv_math.h, dxv_trace.h) on the CPU so that the arithmetic, the Karras
// hierarchy rule and the traversal can be checked against the oracle without a GPU.
// Never linked into libdxv.so; the shipped library has no CPU path.
#include "../../dxrvoxelizer_amd/csrc/dxv_trace.h"
#include "../../dxrvoxelizer_amd/csrc/dxv_dirmap.h"
#include "../../dxrvoxelizer_amd/csrc/dxv_raycast.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace dxv;

// old-style unpack for the replay tools below
static inline void load_node(const Node32* nodes, int32_t i, F4& q0, F4& q1, F4& q2, int32_t& c0, int32_t& c1)
{
    const NodePlanes n = load_node(nodes, i);
    q0 = F4{n.b[0], n.b[1], n.b[2], n.b[3]}; q1 = F4{n.b[4], n.b[5], n.b[6], n.b[7]}; q2 = F4{n.b[8], n.b[9], n.b[10], n.b[11]};
    c0 = n.c0; c1 = n.c1;
}

struct HcScene {
    uint32_t T = 0;
    float bound[4];
    std::vector<uint64_t> keys;
    std::vector<Node> nodes;
    std::vector<Node32> nodes32;
    std::vector<Node64> nodes64;
    std::vector<TriPos> triPos;
    std::vector<TriNrm> triNrm;
    uint32_t height = 0;
    std::vector<DirCell> dmCells;          // direction-space lists (hc_dirmap_build)
    std::vector<DirEntry> dmEntries;
    std::vector<uint32_t> plBegin, plEntries;   // row lists of the parity rule (CSR over R x R texels)
    uint32_t plR = 0;
    uint32_t dmR = 0;
};

static void set_child(Node& n, int side, const float lo[3], const float hi[3], uint32_t h)
{
    float* w = reinterpret_cast<float*>(&n) + 6 * side;
    w[0] = lo[0]; w[1] = lo[1]; w[2] = lo[2]; w[3] = hi[0]; w[4] = hi[1]; w[5] = hi[2];
    (side ? n.h1 : n.h0) = h;
}

static uint32_t refit(HcScene& s, int32_t link, float lo[3], float hi[3])
{
    if (link < 0) {
        const TriPos& tp = s.triPos[~link];
        tri_box(tp.v0, tp.v1, tp.v2, lo, hi);
        return 0;
    }
    Node& n = s.nodes[link];
    float l0[3], h0[3], l1[3], h1[3];
    const uint32_t a = refit(s, n.c0, l0, h0), b = refit(s, n.c1, l1, h1);
    set_child(n, 0, l0, h0, a);
    set_child(n, 1, l1, h1, b);
    for (int k = 0; k < 3; ++k) { lo[k] = min_(l0[k], l1[k]); hi[k] = max_(h0[k], h1[k]); }
    return (a > b ? a : b) + 1;
}

extern "C" {

__attribute__((visibility("default"))) void* hc_scene_create(const float* vb, uint32_t V, const uint32_t* ib, uint32_t T,
                                                               const float bound[4])
{
    (void)V;
    HcScene* s = new HcScene();
    s->T = T;
    memcpy(s->bound, bound, 16);
    s->keys.resize(T);
    auto gather = [&](uint32_t k, F4& a, F4& b, F4& c) {
        a = normalise_pos(vb + 6ull * ib[3ull * k], bound);
        b = normalise_pos(vb + 6ull * ib[3ull * k + 1], bound);
        c = normalise_pos(vb + 6ull * ib[3ull * k + 2], bound);
    };
    for (uint32_t k = 0; k < T; ++k) {
        F4 a, b, c;
        gather(k, a, b, c);
        float lo[3], hi[3];
        tri_box(a, b, c, lo, hi);
        s->keys[k] = morton_key(lo, hi, k);
    }
    std::sort(s->keys.begin(), s->keys.end());
    s->triPos.resize(T);
    s->triNrm.resize(T);
    for (uint32_t i = 0; i < T; ++i) {
        const uint32_t k = (uint32_t)(s->keys[i] & 0xffffffffull);
        F4 a, b, c;
        gather(k, a, b, c);
        a.w = __builtin_bit_cast(float, k);
        const float* n0 = vb + 6ull * ib[3ull * k] + 3;
        const float* n1 = vb + 6ull * ib[3ull * k + 1] + 3;
        const float* n2 = vb + 6ull * ib[3ull * k + 2] + 3;
        s->triNrm[i] = TriNrm{F4{n0[0], n0[1], n0[2], 0}, F4{n1[0], n1[1], n1[2], 0}, F4{n2[0], n2[1], n2[2], 0}};
        b.w = __builtin_bit_cast(float, normal_class(a, b, c, s->triNrm[i].n0, s->triNrm[i].n1, s->triNrm[i].n2) << kClassShift);
        s->triPos[i] = TriPos{a, b, c};
    }
    s->nodes.resize(T > 1 ? T - 1 : 1);
    memset(s->nodes.data(), 0xff, s->nodes.size() * sizeof(Node));
    if (T == 1) {
        float lo[3], hi[3];
        tri_box(s->triPos[0].v0, s->triPos[0].v1, s->triPos[0].v2, lo, hi);
        set_child(s->nodes[0], 0, lo, hi, 0);
        const float far[3] = {1e30f, 1e30f, 1e30f};
        set_child(s->nodes[0], 1, far, far, 0);
        s->nodes[0].c0 = ~0; s->nodes[0].c1 = ~0;
        s->height = 1;
    } else {
        for (uint32_t i = 0; i + 1 < T; ++i) karras_node(s->keys.data(), T, i, s->nodes[i].c0, s->nodes[i].c1);
        float lo[3], hi[3];
        s->height = refit(*s, 0, lo, hi);
    }
    s->nodes32.resize(s->nodes.size());
    for (size_t i = 0; i < s->nodes.size(); ++i) s->nodes32[i] = compress_node(s->nodes[i]);
    s->nodes64.resize(s->nodes.size());
    for (size_t i = 0; i < s->nodes.size(); ++i) s->nodes64[i] = widen_node(s->nodes.data(), (int32_t)i);
    return s;
}

// Direction-space lists built on the host with the product's own footprint code (dxv_dirmap.h); the
// device builder (dirmap.hip) must produce the same lists.  Returns the number of entries.
__attribute__((visibility("default"))) uint64_t hc_dirmap_build(void* p, uint32_t R)
{
    HcScene* s = static_cast<HcScene*>(p);
    std::vector<uint64_t> keys;
    std::vector<DirRecord> rec((size_t)s->T * 6);
    const DirKeyLayout lay = dm_key_layout(R);
    for (uint32_t t = 0; t < s->T; ++t)
        for (uint32_t f = 0; f < 6; ++f) {
            DirRecord e = dm_record(s->triPos[t], f);
            uint32_t i0, i1, j0, j1;
            const bool seen = dm_rect(e, R, i0, i1, j0, j1);
            if (seen) dm_record_on_map(e, (i1 - i0 + 1u) * (j1 - j0 + 1u));
            rec[(size_t)t * 6 + f] = e;
            if (!seen) continue;
            const DirTexelTest tt = dm_texel_test(e, (i1 - i0 + 1u) * (j1 - j0 + 1u));
            for (uint32_t j = j0; j <= j1; ++j)
                for (uint32_t i = i0; i <= i1; ++i) {
                    if (dm_texel_outside(tt, R, i, j)) continue;
                    uint32_t r0h, r1h;
                    dm_local_radial(e, R, i, j, r0h, r1h);
                    keys.push_back(dm_key(lay, (f * R + j) * R + i, (uint16_t)r1h, t));
                }
        }
    std::sort(keys.begin(), keys.end());
    s->dmR = R;
    s->dmCells.assign((size_t)6 * R * R, DirCell{0, 0, 0, 0, 0, 0, 0});
    s->dmEntries.assign(keys.size() + 3, DirEntry{0, 0, 0, 0});          // (+ three spare ones: a scan round loads four)
    for (size_t i = 0; i < keys.size(); ++i) {
        const uint32_t cell = dm_key_cell(lay, keys[i]), t = dm_key_tri(lay, keys[i]);
        const DirRecord& rc = rec[(size_t)t * 6 + cell / (R * R)];
        const uint32_t inFace = cell % (R * R);
        s->dmEntries[i] = dm_local_entry(rc, R, inFace % R, inFace / R, t);
        if (i == 0 || dm_key_cell(lay, keys[i - 1]) != cell) s->dmCells[cell].begin = (uint32_t)i;
        if (s->dmCells[cell].count == 0xffffu) return ~0ull;             // does not fit the 16-bit count
        s->dmCells[cell].count++;
        s->dmCells[cell].r1max = (uint16_t)((s->dmEntries[i].rr >> 16) & 0x7fffu);
        const uint32_t th = half_up(dm_entry_r1(s->dmEntries[i]) - dm_entry_r0(s->dmEntries[i]));
        if (th > s->dmCells[cell].thick) s->dmCells[cell].thick = (uint16_t)th;
    }
    for (DirCell& cell : s->dmCells) {
        const float step = dm_stop_step(half_bits_to_float(cell.thick));   // stop codes: from the far end, earliest start so far
        float smin = 3.0e38f;
        for (uint32_t k = cell.count; k-- > 0u;) {
            DirEntry& e = s->dmEntries[cell.begin + k];
            if (dm_entry_r0(e) < smin) smin = dm_entry_r0(e);
            e.tri = (e.tri & kDmTriMask) | (dm_stop_code(dm_entry_r1(e), smin, step) << kDmTriBits);
        }
        if (cell.count <= 8u) continue;
        const DirSearchHints h = dm_search_hints(cell.count);
        auto r1 = [&](uint32_t k) { return (uint16_t)((s->dmEntries[cell.begin + k].rr >> 16) & 0x7fffu); };
        cell.q2 = r1(h.m2);
        if (h.has1) cell.q1 = r1(h.m1);
        if (h.has3) cell.q3 = r1(h.m3);
    }
    return keys.size();
}
// the max-mip of the far radii from the host-built cells, level by level with the definitions of dxv_dirmap.h (the device builds
// it in tiles: dirmap.hip; same words)
static std::vector<uint16_t> host_mip(const HcScene* s)
{
    const uint32_t R = s->dmR, levels = dm_mip_levels(R);
    std::vector<uint16_t> mip(dm_mip_words(R), 0);
    for (size_t c = 0; c < s->dmCells.size(); ++c) mip[c] = (uint16_t)dm_mip_key(s->dmCells[c]);
    for (uint32_t l = 1; l < levels; ++l) {
        const uint32_t r = R >> l, rp = r << 1;
        const uint16_t* below = mip.data() + dm_mip_offset(R, l - 1);
        uint16_t* out = mip.data() + dm_mip_offset(R, l);
        for (uint32_t f = 0; f < 6; ++f)
            for (uint32_t y = 0; y < r; ++y)
                for (uint32_t x = 0; x < r; ++x) {
                    const uint16_t* a = below + (f * rp + 2 * y) * rp + 2 * x;
                    out[(f * r + y) * r + x] = std::max(std::max(a[0], a[1]), std::max(a[rp], a[rp + 1]));
                }
    }
    return mip;
}
__attribute__((visibility("default"))) void hc_dirmap_mip(void* p, void* out)
{
    const std::vector<uint16_t> mip = host_mip(static_cast<HcScene*>(p));
    memcpy(out, mip.data(), mip.size() * sizeof(uint16_t));
}
// The work queue's brick test (dm_box_may_be_live over dm_brick_hull) against the per-voxel first-step decision of the kernel
// (origin_leaves_root, dm_ray_start) for every brick of a partition: out[0] live voxels, out[1] bricks with a live voxel,
// out[2] bricks the box test keeps, out[3] violations (a live voxel in a brick the test drops: must be 0).
__attribute__((visibility("default"))) void hc_plan_check(void* p, uint32_t N, uint32_t z0, uint32_t nz, uint32_t zBlock, uint32_t zPeriod, uint64_t* out)
{
    HcScene* s = static_cast<HcScene*>(p);
    const std::vector<uint16_t> mip = host_mip(s);
    const DirMapView dm{s->dmCells.data(), s->dmEntries.data(), s->dmR};
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; }
    for (const TriPos& tp : s->triPos) {                                  // root box = union of the canonical triangle boxes
        float l[3], h[3];
        tri_box(tp.v0, tp.v1, tp.v2, l, h);
        for (int a = 0; a < 3; ++a) { lo[a] = min_(lo[a], l[a]); hi[a] = max_(hi[a], h[a]); }
    }
    uint32_t zShift = 0;
    while ((1u << zShift) < zBlock) ++zShift;
    out[0] = out[1] = out[2] = out[3] = 0;
    const uint32_t nbx = (N + 3) / 4, nbz = (nz + 3) / 4;
    for (uint32_t bz = 0; bz < nbz; ++bz)
        for (uint32_t by = 0; by < nbx; ++by)
            for (uint32_t bx = 0; bx < nbx; ++bx) {
                uint32_t live = 0;
                for (uint32_t t = 0; t < 64; ++t) {
                    const uint32_t ix = bx * 4 + (t & 3), iy = by * 4 + ((t >> 2) & 3), lz = bz * 4 + (t >> 4);
                    if (ix >= N || iy >= N || lz >= nz) continue;
                    const uint32_t iz = zBlock == nz ? z0 + lz : z0 + (lz >> zShift) * zPeriod + (lz & (zBlock - 1u));
                    float ox, oy, oz;
                    ray_origin(N, ix, iy, iz, ox, oy, oz);
                    if (origin_leaves_root(ox, oy, oz, lo, hi)) continue;
                    if (dm_ray_start(ox, oy, oz, dm).live) ++live;
                }
                float x0, x1, y0, y1, zl, zh;
                dm_brick_hull(N, nz, z0, zBlock, zShift, zPeriod, bx, by, bz, x0, x1, y0, y1, zl, zh);
                const bool kept = dm_box_may_be_live(x0, x1, y0, y1, zl, zh, lo, hi, mip.data(), s->dmR);
                out[0] += live; out[1] += live ? 1 : 0; out[2] += kept ? 1 : 0;
                if (live && !kept) ++out[3];
            }
}
// dm_footprint of one triangle (9 floats) on one face: out = u0, u1, v0, v1, r0, r1; returns 0 when the face does not see it
__attribute__((visibility("default"))) int hc_dm_footprint(const float* tri, uint32_t face, float* out)
{
    TriPos tp;
    tp.v0 = {tri[0], tri[1], tri[2], 0.0f}; tp.v1 = {tri[3], tri[4], tri[5], 0.0f}; tp.v2 = {tri[6], tri[7], tri[8], 0.0f};
    DirFootprint f;
    if (!dm_footprint(tp, face, f)) return 0;
    out[0] = f.u0; out[1] = f.u1; out[2] = f.v0; out[3] = f.v1; out[4] = f.r0; out[5] = f.r1;
    return 1;
}
// dm_local_radial of one triangle's record on one face and map, for texel (i, j): out = r0, r1 of the whole footprint and of the texel
// (as floats); returns 0 when the face does not see the triangle or the texel lies outside its rectangle, 2 when the record gets
// per-texel ranges on this map (footprints of many texels), else 1
__attribute__((visibility("default"))) int hc_dm_local_radial(const float* tri, uint32_t face, uint32_t R, uint32_t i, uint32_t j, float* out)
{
    TriPos tp;
    tp.v0 = {tri[0], tri[1], tri[2], 0.0f}; tp.v1 = {tri[3], tri[4], tri[5], 0.0f}; tp.v2 = {tri[6], tri[7], tri[8], 0.0f};
    DirRecord rec = dm_record(tp, face);
    uint32_t i0, i1, j0, j1;
    if (!dm_rect(rec, R, i0, i1, j0, j1) || i < i0 || i > i1 || j < j0 || j > j1) return 0;
    dm_record_on_map(rec, (i1 - i0 + 1u) * (j1 - j0 + 1u));
    uint32_t r0h, r1h;
    dm_local_radial(rec, R, i, j, r0h, r1h);
    out[0] = half_to_float((uint16_t)(rec.rr & 0xffffu)); out[1] = half_to_float((uint16_t)(rec.rr >> 16));
    out[2] = half_to_float((uint16_t)r0h); out[3] = half_to_float((uint16_t)r1h);
    return (rec.hasTri & 2u) ? 2 : 1;
}
// dm_texel_outside for texel (i, j) of a triangle's record on one face and map: 1 = no entry there, 0 = entry (or no record / outside
// the rectangle: -1)
__attribute__((visibility("default"))) int hc_dm_texel_outside(const float* tri, uint32_t face, uint32_t R, uint32_t i, uint32_t j)
{
    TriPos tp;
    tp.v0 = {tri[0], tri[1], tri[2], 0.0f}; tp.v1 = {tri[3], tri[4], tri[5], 0.0f}; tp.v2 = {tri[6], tri[7], tri[8], 0.0f};
    const DirRecord rec = dm_record(tp, face);
    uint32_t i0, i1, j0, j1;
    if (!dm_rect(rec, R, i0, i1, j0, j1) || i < i0 || i > i1 || j < j0 || j > j1) return -1;
    return dm_texel_outside(dm_texel_test(rec, (i1 - i0 + 1u) * (j1 - j0 + 1u)), R, i, j) ? 1 : 0;
}
__attribute__((visibility("default"))) uint32_t hc_normal_class(const float* tri, const float* nrm)
{
    return normal_class(F4{tri[0], tri[1], tri[2], 0}, F4{tri[3], tri[4], tri[5], 0}, F4{tri[6], tri[7], tri[8], 0},
                        F4{nrm[0], nrm[1], nrm[2], 0}, F4{nrm[3], nrm[4], nrm[5], 0}, F4{nrm[6], nrm[7], nrm[8], 0});
}
__attribute__((visibility("default"))) void hc_scene_tripos(void* p, void* out) { auto* s = static_cast<HcScene*>(p); memcpy(out, s->triPos.data(), s->triPos.size() * sizeof(TriPos)); }
// Row lists of the parity rule built on the host with the product's pl_rect; returns the number of entries.
__attribute__((visibility("default"))) uint64_t hc_plists_build(void* p, uint32_t R)
{
    HcScene* s = static_cast<HcScene*>(p);
    std::vector<uint32_t> count((size_t)R * R + 1, 0);
    for (uint32_t t = 0; t < s->T; ++t) {
        uint32_t j0, j1, k0, k1;
        pl_rect(s->triPos[t], R, j0, j1, k0, k1);
        for (uint32_t k = k0; k <= k1; ++k) for (uint32_t j = j0; j <= j1; ++j) count[(size_t)k * R + j + 1]++;
    }
    for (size_t c = 0; c < (size_t)R * R; ++c) count[c + 1] += count[c];
    s->plBegin = count;
    s->plEntries.assign(count.back(), 0);
    std::vector<uint32_t> cur(count.begin(), count.end() - 1);
    for (uint32_t t = s->T; t-- > 0u;) {                               // (any order inside a list: this one is the reverse of the device's usual)
        uint32_t j0, j1, k0, k1;
        pl_rect(s->triPos[t], R, j0, j1, k0, k1);
        for (uint32_t k = k0; k <= k1; ++k) for (uint32_t j = j0; j <= j1; ++j) s->plEntries[cur[(size_t)k * R + j]++] = t;
    }
    s->plR = R;
    return s->plEntries.size();
}
// Debug aid: every entry of the texel of voxel (ix, iy, iz)'s ray, with the outcome of each test of the scan.
__attribute__((visibility("default"))) void hc_dirmap_debug(void* p, uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz, uint32_t wantK)
{
    HcScene* s = static_cast<HcScene*>(p);
    Ray r;
    ray_origin(N, ix, iy, iz, r.ox, r.oy, r.oz);
    uint32_t face; float u, v, rho;
    dm_ray_point(r.ox, r.oy, r.oz, face, u, v, rho);
    uint32_t ti, tj, cx, cy;
    dm_local(u, s->dmR, ti, cx); dm_local(v, s->dmR, tj, cy);
    const DirCell cell = s->dmCells[(face * s->dmR + tj) * s->dmR + ti];
    const float near = rho * 0.999f;
    printf("o=(%g %g %g) face %u u %g v %g rho %.9g texel (%u,%u) local (%u,%u) count %u r1max %g thick %g near %.9g\n", r.ox, r.oy, r.oz, face, u, v, rho, ti, tj, cx, cy,
           cell.count, half_bits_to_float(cell.r1max), half_bits_to_float(cell.thick), near);
    const DirRayLocal loc = dm_ray_local(cx, cy);
    const uint32_t rc = dm_radial_word(near, 3.0e38f);
    const float step = dm_stop_step(half_bits_to_float(cell.thick));
    for (uint32_t k = 0; k < cell.count; ++k) {
        const DirEntry e = s->dmEntries[cell.begin + k];
        const TriPos tp = load_tri(s->triPos.data(), (int32_t)dm_entry_tri(e));
        const uint32_t kk = __builtin_bit_cast(uint32_t, tp.v0.w);
        if (wantK != 0xffffffffu && kk != wantK) continue;
        printf("  #%u tri %u r0 %.9g r1 %.9g stop %.9g box %08x edge %08x boxpass %d radial %d edge %d (dot %d)\n", k, kk, dm_entry_r0(e), dm_entry_r1(e), dm_stop_radius(e, step), e.box, e.edge,
               ((loc.q - e.box) & 0x80808080u) == 0x80808080u, ((e.rr - rc) & 0x80008000u) == 0x80008000u, dm_dot4(e.edge, loc.p) >= 0, dm_dot4(e.edge, loc.p));
    }
}
__attribute__((visibility("default"))) void hc_dirmap_get(void* p, void* cells, void* entries)
{
    HcScene* s = static_cast<HcScene*>(p);
    memcpy(cells, s->dmCells.data(), s->dmCells.size() * sizeof(DirCell));
    memcpy(entries, s->dmEntries.data(), (s->dmEntries.size() - 3) * sizeof(DirEntry));
}

__attribute__((visibility("default"))) void hc_scene_nodes32(void* p, void* out) { auto* s = static_cast<HcScene*>(p); memcpy(out, s->nodes32.data(), s->nodes32.size() * sizeof(Node32)); }
__attribute__((visibility("default"))) uint32_t hc_dm_half_pos(float x, int up) { return up ? dm_half_up_pos(x) : dm_half_down_pos(x); }
__attribute__((visibility("default"))) uint32_t hc_half_down(float x) { return half_down(x); }
__attribute__((visibility("default"))) uint32_t hc_half_up(float x) { return half_up(x); }
__attribute__((visibility("default"))) float hc_half_to_float(uint32_t h) { return half_to_float((uint16_t)h); }
__attribute__((visibility("default"))) void hc_scene_destroy(void* p) { delete static_cast<HcScene*>(p); }
__attribute__((visibility("default"))) uint32_t hc_scene_height(void* p) { return static_cast<HcScene*>(p)->height; }
__attribute__((visibility("default"))) void hc_scene_nodes64(void* p, void* out) { auto* s = static_cast<HcScene*>(p); memcpy(out, s->nodes64.data(), s->nodes64.size() * sizeof(Node64)); }
__attribute__((visibility("default"))) void hc_scene_nodes(void* p, void* out) { auto* s = static_cast<HcScene*>(p); memcpy(out, s->nodes.data(), s->nodes.size() * sizeof(Node)); }
__attribute__((visibility("default"))) void hc_scene_keys(void* p, void* out) { auto* s = static_cast<HcScene*>(p); memcpy(out, s->keys.data(), s->keys.size() * 8); }

__attribute__((visibility("default"))) int hc_voxelize(void* p, uint32_t N, int mode, uint32_t z0, uint32_t nz, int stackCap,
                                                         uint8_t* out, uint32_t* texels)
{
    HcScene* s = static_cast<HcScene*>(p);
    if (mode == 4 || mode == 7 || mode == 8 || mode == 10 || mode == 11) {   // parity, row form (what k_parity_rows computes): one walk per block of RB x RB rows
        const uint32_t RB = (mode == 4 || mode == 10) ? 1u : (mode == 7 || mode == 11) ? 2u : 4u;
        const bool wideWalk = mode >= 10;          // ... over the four-box nodes
        SceneView scr{s->nodes32.data(), s->triPos.data(), s->triNrm.data(), {0, 0, 0}, {0, 0, 0}};
        const float* w = reinterpret_cast<const float*>(&s->nodes[0]);
        for (int a = 0; a < 3; ++a) { scr.rootLo[a] = min_(w[a], w[6 + a]); scr.rootHi[a] = max_(w[3 + a], w[9 + a]); }
        const uint32_t by = (N + RB - 1) / RB, bz = (nz + RB - 1) / RB;
#pragma omp parallel for schedule(dynamic, 4)
        for (int64_t blk = 0; blk < (int64_t)by * bz; ++blk) {
            const uint32_t biy = (uint32_t)(blk % by), blz = (uint32_t)(blk / by);
            uint32_t iy[4], lz[4];
            float oy[4], oz[4], t0, t1;
            std::vector<float> ox(N);
            for (uint32_t k = 0; k < RB; ++k) {
                iy[k] = biy * RB + k < N ? biy * RB + k : N - 1;
                lz[k] = blz * RB + k < nz ? blz * RB + k : nz - 1;
                ray_origin(N, 0, iy[k], z0 + lz[k], t0, oy[k], t1);
                ray_origin(N, 0, iy[0], z0 + lz[k], t0, t1, oz[k]);
            }
            for (uint32_t ix = 0; ix < N; ++ix) ray_origin(N, ix, iy[0], z0, ox[ix], t0, t1);
            float ylo = oy[0], yhi = oy[0], zlo = oz[0], zhi = oz[0];
            for (uint32_t k = 1; k < RB; ++k) { ylo = min_(ylo, oy[k]); yhi = max_(yhi, oy[k]); zlo = min_(zlo, oz[k]); zhi = max_(zhi, oz[k]); }
            std::vector<uint32_t> cnt((size_t)RB * RB * N, 0);
            struct HostStack { int32_t e[128]; void push(int& sp, int32_t v) { e[sp++] = v; } int32_t pop(int& sp) { return e[--sp]; } } stk;
            auto tri = [&](const TriPos& tp) {
                for (uint32_t r = 0; r < RB * RB; ++r) {
                    const ParityRowTri ps = parity_row_setup(oy[r % RB], oz[r / RB], tp.v0, tp.v1, tp.v2);
                    if (ps.hit) for (uint32_t ix = 0; ix < N; ++ix) cnt[(size_t)r * N + ix] += parity_row_voxel(ps, ox[ix]) ? 1u : 0u;
                }
            };
            auto triAt = [&](int32_t leaf) { return load_tri(scr.triPos, leaf); };
            if (scr.rootLo[1] <= yhi && ylo <= scr.rootHi[1] && scr.rootLo[2] <= zhi && zlo <= scr.rootHi[2] && scr.rootHi[0] >= ox[0]) {
                if (wideWalk)
                    walk_parity_rows_wide([&](int32_t i) { return parity_rows_wide_node(s->nodes64[i], ylo, yhi, zlo, zhi, ox[0]); }, triAt, stk, tri);
                else
                    walk_parity_rows([&](int32_t i) { return parity_rows_node(load_node(scr.nodes, i), ylo, yhi, zlo, zhi, ox[0]); }, triAt, stk, tri);
            }
            for (uint32_t r = 0; r < RB * RB; ++r)
                for (uint32_t ix = 0; ix < N; ++ix) out[((size_t)lz[r / RB] * N + iy[r % RB]) * N + ix] = (uint8_t)(cnt[(size_t)r * N + ix] & 1u);
        }
        return 0;
    }
    if (mode == 13) {                              // parity through the row lists (what k_parity_rows<.., LISTS> computes): one texel per row
        if (!s->plR) return -1;
#pragma omp parallel for schedule(dynamic, 4)
        for (int64_t row = 0; row < (int64_t)nz * N; ++row) {
            const uint32_t lz = (uint32_t)(row / N), iy = (uint32_t)(row % N);
            float ox0, oy, oz, t0, t1;
            ray_origin(N, 0, iy, z0 + lz, ox0, oy, oz);
            const size_t cell = (size_t)dm_texel(oz, s->plR) * s->plR + dm_texel(oy, s->plR);
            std::vector<uint32_t> cnt(N, 0);
            for (uint32_t e = s->plBegin[cell]; e < s->plBegin[cell + 1]; ++e) {
                const TriPos& tp = s->triPos[s->plEntries[e]];
                const ParityRowTri ps = parity_row_setup(oy, oz, tp.v0, tp.v1, tp.v2);
                if (!ps.hit) continue;
                for (uint32_t ix = 0; ix < N; ++ix) {
                    float ox;
                    ray_origin(N, ix, iy, z0 + lz, ox, t0, t1);
                    cnt[ix] += parity_row_voxel(ps, ox) ? 1u : 0u;
                }
            }
            for (uint32_t ix = 0; ix < N; ++ix) out[((size_t)lz * N + iy) * N + ix] = (uint8_t)(cnt[ix] & 1u);
        }
        return 0;
    }
    int overflow = 0;
    SceneView sc{s->nodes32.data(), s->triPos.data(), s->triNrm.data(), {0, 0, 0}, {0, 0, 0}, s->nodes64.data(),
                 s->dmCells.data(), s->dmEntries.data(), s->dmR};
    if (mode == 12 && !s->dmR) return -1;
    {   // root box = union of the root node's two child boxes (as k_root_info computes it)
        const float* w = reinterpret_cast<const float*>(&s->nodes[0]);
        for (int a = 0; a < 3; ++a) { sc.rootLo[a] = min_(w[a], w[6 + a]); sc.rootHi[a] = max_(w[3 + a], w[9 + a]); }
    }
#pragma omp parallel for schedule(dynamic, 4) reduction(| : overflow)
    for (int64_t row = 0; row < (int64_t)nz * N; ++row) {
        const uint32_t lz = (uint32_t)(row / N), iy = (uint32_t)(row % N), iz = z0 + lz;
        int32_t stack[128];
        StridedStack stk{stack, 1};
        for (uint32_t ix = 0; ix < N; ++ix) {
            const size_t id = ((size_t)lz * N + iy) * N + ix;
            bool ovf = false;
            uint32_t texel = 0;
            out[id] = mode == 0 ? voxel_reference<0>(sc, N, ix, iy, iz, stk, stackCap, texels ? &texel : nullptr, ovf)
                      : mode == 2 ? voxel_reference<1>(sc, N, ix, iy, iz, stk, stackCap, texels ? &texel : nullptr, ovf)
                      : mode == 6 ? voxel_reference<2>(sc, N, ix, iy, iz, stk, stackCap, texels ? &texel : nullptr, ovf)
                      : mode == 12 ? voxel_reference<4>(sc, N, ix, iy, iz, stk, stackCap, texels ? &texel : nullptr, ovf)
                      : mode == 3 ? voxel_parity<true>(sc, N, ix, iy, iz, stk, stackCap, ovf)
                                  : voxel_parity<false>(sc, N, ix, iy, iz, stk, stackCap, ovf);
            if (texels) texels[id] = texel;
            if (ovf) overflow |= 1;
        }
    }
    return overflow;
}

// per-ray traversal statistics over slices z0, z0+zstep, ... (tuning aid):
// out[0] rays, out[1] internal-node visits, out[2] leaf tests, out[3] max stack depth,
// out[4] rays with zero leaf tests, out[5] rays with <= 2 node visits; hist[64] of max stack depth
__attribute__((visibility("default"))) void hc_trace_stats(void* p, uint32_t N, uint32_t zstep, uint64_t* out, uint64_t* hist)
{
    HcScene* s = static_cast<HcScene*>(p);
    uint64_t rays = 0, nodes = 0, leaves = 0, maxsp = 0, noleaf = 0, trivial = 0;
    uint64_t h[64] = {0};
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays, nodes, leaves, noleaf, trivial, h[:64]) reduction(max : maxsp)
    for (int64_t row = 0; row < (int64_t)((N + zstep - 1) / zstep) * N; ++row) {
        const uint32_t iz = (uint32_t)(row / N) * zstep, iy = (uint32_t)(row % N);
        int32_t stack[128];
        StridedStack stk{stack, 1};
        for (uint32_t ix = 0; ix < N; ++ix) {
            Ray r = make_ray_reference(N, ix, iy, iz);
            Hit best;
            TraceStats st{0, 0, 0};
            trace_reference<StridedStack, true>(r, s->nodes32.data(), s->triPos.data(), stk, 128, best, &st);
            rays++; nodes += st.nodes; leaves += st.leaves; noleaf += st.leaves == 0; trivial += st.nodes <= 2;
            if (st.maxsp > maxsp) maxsp = st.maxsp;
            h[st.maxsp < 63 ? st.maxsp : 63]++;
        }
    }
    out[0] = rays; out[1] = nodes; out[2] = leaves; out[3] = maxsp; out[4] = noleaf; out[5] = trivial;
    for (int i = 0; i < 64; ++i) hist[i] = h[i];
}

static bool scells_any(const bool* b) { for (int t = 0; t < 64; ++t) if (b[t]) return true; return false; }
// Lists replay per 4x4x4 brick (design aid; tools/list_stats.py): out[0] waves with a live lane, [1] live lanes, [2] lanes
// ended by the texel's far radius, [3] waves without a scanning lane, [4] distinct texels of the scanning lanes (sum over
// waves), [5] their lists' lengths together, [6] entries scanned (start search to early stop), [7] sum over waves of the
// longest lane scan, [8] entries whose byte box contains the ray, [9] ... that also pass the edge = triangles selected,
// [10] sum over waves of the most selections of a lane (= triangle-step rounds), [11] lanes with a hit,
// [12] DISTINCT triangles among a wave's selections (sum over waves: [9] - [12] fetches of a 48-byte record are repeats),
// [13] hits whose triangle answers the normal test by its class (no 48-byte normal record fetched)
__attribute__((visibility("default"))) void hc_list_stats(void* p, uint32_t N, uint32_t bstep, uint64_t* out)
{
    HcScene* s = static_cast<HcScene*>(p);
    float rootLo[3], rootHi[3];
    {
        const float* w = reinterpret_cast<const float*>(&s->nodes[0]);
        for (int a = 0; a < 3; ++a) { rootLo[a] = min_(w[a], w[6 + a]); rootHi[a] = max_(w[3 + a], w[9 + a]); }
    }
    const uint32_t nb = N / 4, R = s->dmR;
    uint64_t o[14] = {0};
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : o[:14])
    for (int64_t bzi = 0; bzi < (int64_t)nb; bzi += bstep) {
        for (uint32_t byi = 0; byi < nb; ++byi) for (uint32_t bxi = 0; bxi < nb; ++bxi) {
            int nlive = 0;
            std::vector<uint32_t> cellsSeen, trisSel;
            uint64_t mScan = 0, mSel = 0;
            for (int t = 0; t < 64; ++t) {
                const uint32_t ix = bxi * 4 + t % 4, iy = byi * 4 + (t / 4) % 4, iz = (uint32_t)bzi * 4 + t / 16;
                Ray r;
                ray_origin(N, ix, iy, iz, r.ox, r.oy, r.oz);
                if (origin_leaves_root(r.ox, r.oy, r.oz, rootLo, rootHi)) continue;
                finish_ray_reference(r);
                nlive++;
                uint32_t face, ti, tj, cx, cy; float u, v, rho;
                dm_ray_point(r.ox, r.oy, r.oz, face, u, v, rho);
                dm_local(u, R, ti, cx); dm_local(v, R, tj, cy);
                const uint32_t ci = (face * R + tj) * R + ti;
                const DirCell cell = s->dmCells[ci];
                const float near = rho * 0.999f;
                if (half_bits_to_float(cell.r1max) < near || !cell.count) { o[2]++; continue; }
                cellsSeen.push_back(ci);
                const DirRayLocal loc = dm_ray_local(cx, cy);
                uint32_t i = cell.begin, hi = cell.begin + cell.count;
                while (hi > i) { const uint32_t mid = i + ((hi - i) >> 1); if (dm_entry_r1(s->dmEntries[mid]) < near) i = mid + 1u; else hi = mid; }
                const float thick = half_bits_to_float(cell.thick);
                Hit best; best.t = kTMax; best.k = 0xffffffffu; best.leaf = -1; best.b1 = best.b2 = 0;
                uint64_t scan = 0, sel = 0;
                for (uint32_t e = i; e < cell.begin + cell.count; ++e) {
                    const DirEntry& en = s->dmEntries[e];
                    const float bound = (rho + best.t) * 1.001f + 1e-4f;
                    if (dm_stop_radius(en, dm_stop_step(thick)) > bound) break;
                    scan++;
                    DirEntry boxOnly = en; boxOnly.edge = 0u;
                    const uint32_t rc = dm_radial_word(near, bound);
                    if (dm_local_pass(boxOnly, loc, rc)) o[8]++;
                    if (dm_local_pass(en, loc, rc)) { sel++; trisSel.push_back(dm_entry_tri(en)); leaf_reference(r, s->triPos.data(), (int32_t)dm_entry_tri(en), best); }
                }
                o[6] += scan; o[9] += sel;
                if (scan > mScan) mScan = scan;
                if (sel > mSel) mSel = sel;
                if (best.k != 0xffffffffu) { o[11]++; if (__builtin_bit_cast(uint32_t, s->triPos[best.leaf].v1.w) >> kClassShift) o[13]++; }
            }
            if (!nlive) continue;
            o[0]++; o[1] += nlive; o[7] += mScan; o[10] += mSel;
            if (cellsSeen.empty()) o[3]++;
            std::sort(cellsSeen.begin(), cellsSeen.end());
            cellsSeen.erase(std::unique(cellsSeen.begin(), cellsSeen.end()), cellsSeen.end());
            o[4] += cellsSeen.size();
            std::sort(trisSel.begin(), trisSel.end());
            o[12] += std::unique(trisSel.begin(), trisSel.end()) - trisSel.begin();
            for (uint32_t ci : cellsSeen) o[5] += s->dmCells[ci].count;
        }
    }
    for (int i = 0; i < 14; ++i) out[i] = o[i];
}

static uint64_t g_uniform = 0, g_distinct = 0;
#pragma omp threadprivate(g_uniform, g_distinct)
// Lockstep (SIMT) replay of the reference-mode loop for 4x4x4-voxel waves: every iteration each
// live lane visits one node.  out[0] waves, out[1] wave iterations, out[2] sum of live lanes over
// iterations, out[3] iterations in which >=1 lane runs a leaf test, out[4] lanes in those leaf
// tests, out[5] second-leaf executions, out[6] waves with zero iterations (all lanes culled by the
// root early-out), out[7] iterations of the heaviest wave.
__attribute__((visibility("default"))) void hc_simt_stats(void* p, uint32_t N, uint32_t bstep, uint64_t* out)
{
    HcScene* s = static_cast<HcScene*>(p);
    SceneView sc{s->nodes32.data(), s->triPos.data(), s->triNrm.data(), {0, 0, 0}, {0, 0, 0}};
    {
        const float* w = reinterpret_cast<const float*>(&s->nodes[0]);
        for (int a = 0; a < 3; ++a) { sc.rootLo[a] = min_(w[a], w[6 + a]); sc.rootHi[a] = max_(w[3 + a], w[9 + a]); }
    }
    const uint32_t nb = N / 4;
    uint64_t waves = 0, iters = 0, live = 0, leafIters = 0, leafLanes = 0, second = 0, empty = 0, heaviest = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : waves, iters, live, leafIters, leafLanes, second, empty) reduction(max : heaviest)
    for (int64_t bzi = 0; bzi < (int64_t)nb; bzi += bstep) {
        for (uint32_t byi = 0; byi < nb; ++byi) for (uint32_t bxi = 0; bxi < nb; ++bxi) {
            struct Lane { Ray r; Hit best; int32_t node; int sp; int32_t stack[64]; };
            static thread_local Lane L[64];
            int nlive = 0;
            for (int t = 0; t < 64; ++t) {
                Lane& l = L[t];
                const uint32_t ix = bxi * 4 + t % 4, iy = byi * 4 + (t / 4) % 4, iz = (uint32_t)bzi * 4 + t / 16;
                ray_origin(N, ix, iy, iz, l.r.ox, l.r.oy, l.r.oz);
                l.node = -1;
                if (origin_leaves_root(l.r.ox, l.r.oy, l.r.oz, sc.rootLo, sc.rootHi)) continue;
                finish_ray_reference(l.r);
                l.best.t = kTMax; l.best.k = 0xffffffffu; l.best.leaf = -1; l.best.b1 = l.best.b2 = 0;
                l.stack[0] = -1; l.sp = 1; l.node = 0;
                nlive++;
            }
            waves++;
            if (!nlive) { empty++; continue; }
            uint64_t myIters = 0;
            for (;;) {
                int liveNow = 0, leafNow = 0, secondNow = 0;
                {
                    int32_t first = -2; bool uni = true; int distinct = 0; int32_t seen[64];
                    for (int t = 0; t < 64; ++t) {
                        if (L[t].node < 0) continue;
                        if (first == -2) first = L[t].node; else if (L[t].node != first) uni = false;
                        bool f = false; for (int q = 0; q < distinct; ++q) if (seen[q] == L[t].node) { f = true; break; }
                        if (!f) seen[distinct++] = L[t].node;
                    }
                    if (first != -2) { g_uniform += uni ? 1 : 0; g_distinct += distinct; }
                }
                for (int t = 0; t < 64; ++t) {
                    Lane& l = L[t];
                    if (l.node < 0) continue;
                    liveNow++;
                    F4 q0, q1, q2; int32_t c0, c1;
                    load_node(sc.nodes, l.node, q0, q1, q2, c0, c1);
                    float tn0, tn1;
                    bool h0 = slab(l.r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tn0) && tn0 <= l.best.t;
                    bool h1 = slab(l.r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tn1) && tn1 <= l.best.t;
                    const bool l0 = h0 && c0 < 0, l1 = h1 && c1 < 0;
                    if (l0 || l1) {
                        leafNow++;
                        leaf_reference(l.r, sc.triPos, l0 ? ~c0 : ~c1, l.best);
                        if (l0 && l1) { secondNow++; leaf_reference(l.r, sc.triPos, ~c1, l.best); }
                    }
                    h0 = h0 && c0 >= 0 && tn0 <= l.best.t;
                    h1 = h1 && c1 >= 0 && tn1 <= l.best.t;
                    const bool both = h0 && h1, swap = tn1 < tn0;
                    if (both) l.stack[l.sp++] = swap ? c0 : c1;
                    if (h0 || h1) l.node = (h0 && !(both && swap)) ? c0 : c1;
                    else l.node = l.stack[--l.sp];
                }
                if (!liveNow) break;
                myIters++; live += liveNow;
                if (leafNow) { leafIters++; leafLanes += leafNow; }
                if (secondNow) second++;
            }
            iters += myIters;
            if (myIters > heaviest) heaviest = myIters;
        }
    }
    out[0] = waves; out[1] = iters; out[2] = live; out[3] = leafIters; out[4] = leafLanes; out[5] = second; out[6] = empty; out[7] = heaviest;
    uint64_t uni = 0, dis = 0;
#pragma omp parallel reduction(+ : uni, dis)
    { uni += g_uniform; dis += g_distinct; g_uniform = 0; g_distinct = 0; }
    out[8] = uni; out[9] = dis;
}

// Packet replay: one walk per 4x4x4-voxel wave, a node is visited when ANY live lane's ray meets its
// box within that lane's closest hit so far; every lane tests both child boxes of every visited node.
// Near child first by the vote of the lanes that hit both.  out[0] waves, out[1] node visits,
// out[2] leaf visits (a triangle tested by the lanes whose ray meets its box), out[3] lanes in them,
// out[4] heaviest wave (visits + leaves), out[5] deepest stack, out[6] grids differ (must be 0)
__attribute__((visibility("default"))) void hc_packet_stats(void* p, uint32_t N, uint32_t bstep, uint64_t* out)
{
    HcScene* s = static_cast<HcScene*>(p);
    SceneView sc{s->nodes32.data(), s->triPos.data(), s->triNrm.data(), {0, 0, 0}, {0, 0, 0}};
    {
        const float* w = reinterpret_cast<const float*>(&s->nodes[0]);
        for (int a = 0; a < 3; ++a) { sc.rootLo[a] = min_(w[a], w[6 + a]); sc.rootHi[a] = max_(w[3 + a], w[9 + a]); }
    }
    const uint32_t nb = N / 4;
    uint64_t waves = 0, visits = 0, leaves = 0, leafLanes = 0, heaviest = 0, deepest = 0, differ = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : waves, visits, leaves, leafLanes, differ) reduction(max : heaviest, deepest)
    for (int64_t bzi = 0; bzi < (int64_t)nb; bzi += bstep) {
        for (uint32_t byi = 0; byi < nb; ++byi) for (uint32_t bxi = 0; bxi < nb; ++bxi) {
            struct Lane { Ray r; Hit best; bool live; };
            Lane L[64];
            int nlive = 0;
            for (int t = 0; t < 64; ++t) {
                Lane& l = L[t];
                const uint32_t ix = bxi * 4 + t % 4, iy = byi * 4 + (t / 4) % 4, iz = (uint32_t)bzi * 4 + t / 16;
                ray_origin(N, ix, iy, iz, l.r.ox, l.r.oy, l.r.oz);
                l.live = !origin_leaves_root(l.r.ox, l.r.oy, l.r.oz, sc.rootLo, sc.rootHi);
                if (!l.live) continue;
                finish_ray_reference(l.r);
                l.best.t = kTMax; l.best.k = 0xffffffffu; l.best.leaf = -1; l.best.b1 = l.best.b2 = 0;
                nlive++;
            }
            waves++;
            if (!nlive) continue;
            int32_t stack[256]; int sp = 0; int32_t node = 0;
            uint64_t mine = 0;
            for (;;) {
                F4 q0, q1, q2; int32_t c0, c1;
                load_node(sc.nodes, node, q0, q1, q2, c0, c1);
                visits++; mine++;
                bool h0[64], h1[64]; bool any0 = false, any1 = false; int vote = 0;
                for (int t = 0; t < 64; ++t) {
                    h0[t] = h1[t] = false;
                    if (!L[t].live) continue;
                    float tn0, tn1;
                    h0[t] = slab(L[t].r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tn0) && tn0 <= L[t].best.t;
                    h1[t] = slab(L[t].r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tn1) && tn1 <= L[t].best.t;
                    any0 |= h0[t]; any1 |= h1[t];
                    if (h0[t] && h1[t]) vote += tn1 < tn0 ? 1 : -1;
                }
                for (int side = 0; side < 2; ++side) {
                    const int32_t c = side ? c1 : c0;
                    if (!(side ? any1 : any0) || c >= 0) continue;
                    leaves++; mine++;
                    for (int t = 0; t < 64; ++t)
                        if ((side ? h1[t] : h0[t])) { leafLanes++; leaf_reference(L[t].r, sc.triPos, ~c, L[t].best); }
                }
                const bool i0 = any0 && c0 >= 0, i1 = any1 && c1 >= 0;
                if (i0 && i1) {
                    const bool swap = vote > 0;
                    stack[sp++] = swap ? c0 : c1;
                    if ((uint64_t)sp > deepest) deepest = sp;
                    node = swap ? c1 : c0;
                } else if (i0 || i1) node = i0 ? c0 : c1;
                else { if (!sp) break; node = stack[--sp]; }
            }
            if (mine > heaviest) heaviest = mine;
            // same voxels as the per-lane walk
            for (int t = 0; t < 64; ++t) {
                if (!L[t].live) continue;
                Ray r = L[t].r; Hit best;
                int32_t col[64];
                trace_reference(r, sc.nodes, sc.triPos, StridedStack{col, 1}, 64, best);
                if (best.k != L[t].best.k || best.t != L[t].best.t) differ++;
            }
        }
    }
    out[0] = waves; out[1] = visits; out[2] = leaves; out[3] = leafLanes; out[4] = heaviest; out[5] = deepest; out[6] = differ;
}

// Lockstep replay of the while-while variant: phase 1 = every lane walks internal nodes until it
// has a pending leaf (or is finished), phase 2 = all lanes with a pending leaf test it together.
// out[0] waves, out[1] phase-1 iterations, out[2] live lanes in them, out[3] phase-2 executions,
// out[4] lanes in them, out[5] heaviest wave cost (70*p1 + 150*p2), out[6] total cost
__attribute__((visibility("default"))) void hc_simt_stats_ww(void* p, uint32_t N, uint32_t bstep, int carry, uint64_t* out)
{
    HcScene* s = static_cast<HcScene*>(p);
    SceneView sc{s->nodes32.data(), s->triPos.data(), s->triNrm.data(), {0, 0, 0}, {0, 0, 0}};
    {
        const float* w = reinterpret_cast<const float*>(&s->nodes[0]);
        for (int a = 0; a < 3; ++a) { sc.rootLo[a] = min_(w[a], w[6 + a]); sc.rootHi[a] = max_(w[3 + a], w[9 + a]); }
    }
    const uint32_t nb = N / 4;
    uint64_t waves = 0, p1 = 0, p1live = 0, p2 = 0, p2lanes = 0, heaviest = 0, total = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : waves, p1, p1live, p2, p2lanes, total) reduction(max : heaviest)
    for (int64_t bzi = 0; bzi < (int64_t)nb; bzi += bstep) {
        for (uint32_t byi = 0; byi < nb; ++byi) for (uint32_t bxi = 0; bxi < nb; ++bxi) {
            struct Lane { Ray r; Hit best; int32_t node; int sp; int32_t stack[64]; int32_t pa, pb; float ta, tb; };
            static thread_local Lane L[64];
            for (int t = 0; t < 64; ++t) {
                Lane& l = L[t];
                const uint32_t ix = bxi * 4 + t % 4, iy = byi * 4 + (t / 4) % 4, iz = (uint32_t)bzi * 4 + t / 16;
                ray_origin(N, ix, iy, iz, l.r.ox, l.r.oy, l.r.oz);
                l.node = -1; l.pa = l.pb = -1;
                if (origin_leaves_root(l.r.ox, l.r.oy, l.r.oz, sc.rootLo, sc.rootHi)) continue;
                finish_ray_reference(l.r);
                l.best.t = kTMax; l.best.k = 0xffffffffu; l.best.leaf = -1; l.best.b1 = l.best.b2 = 0;
                l.stack[0] = -1; l.sp = 1; l.node = 0;
            }
            waves++;
            uint64_t my1 = 0, my2 = 0;
            for (;;) {
                // phase 1
                for (;;) {
                    int liveNow = 0;
                    for (int t = 0; t < 64; ++t) {
                        Lane& l = L[t];
                        if (l.node < 0 || l.pa >= 0) continue;
                        liveNow++;
                        F4 q0, q1, q2; int32_t c0, c1;
                        load_node(sc.nodes, l.node, q0, q1, q2, c0, c1);
                        float tn0, tn1;
                        bool h0 = slab(l.r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tn0) && tn0 <= l.best.t;
                        bool h1 = slab(l.r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tn1) && tn1 <= l.best.t;
                        const bool l0 = h0 && c0 < 0, l1 = h1 && c1 < 0;
                        if (l0) { l.pa = ~c0; l.ta = tn0; if (l1) { l.pb = ~c1; l.tb = tn1; } }
                        else if (l1) { l.pa = ~c1; l.ta = tn1; }
                        h0 = h0 && c0 >= 0; h1 = h1 && c1 >= 0;
                        const bool both = h0 && h1, swap = tn1 < tn0;
                        if (both) l.stack[l.sp++] = swap ? c0 : c1;
                        if (h0 || h1) l.node = (h0 && !(both && swap)) ? c0 : c1;
                        else l.node = l.stack[--l.sp];
                    }
                    if (!liveNow) break;
                    my1++; p1live += liveNow;
                }
                // phase 2
                int pend = 0;
                for (int t = 0; t < 64; ++t) if (L[t].pa >= 0) pend++;
                if (!pend) break;
                my2++; p2lanes += pend;
                int sec = 0;
                for (int t = 0; t < 64; ++t) {
                    Lane& l = L[t];
                    if (l.pa < 0) continue;
                    leaf_reference(l.r, sc.triPos, l.pa, l.best);
                    if (carry) { l.pa = l.pb; l.ta = l.tb; l.pb = -1; }
                    else { l.pa = -1; if (l.pb >= 0) { sec++; leaf_reference(l.r, sc.triPos, l.pb, l.best); l.pb = -1; } }
                }
                if (sec) { my2++; p2lanes += sec; }
            }
            p1 += my1; p2 += my2;
            const uint64_t cost = 70 * my1 + 150 * my2;
            total += cost;
            if (cost > heaviest) heaviest = cost;
        }
    }
    out[0] = waves; out[1] = p1; out[2] = p1live; out[3] = p2; out[4] = p2lanes; out[5] = heaviest; out[6] = total;
}

// Lockstep replay of the postponed-leaf variant: every live lane keeps walking nodes; hit leaves go to
// a per-lane queue (capacity Q).  The wave flushes (rounds of one leaf test per lane) when a lane's
// queue cannot take two more entries, when >= thr lanes hold leaves, or when no lane can walk.
__attribute__((visibility("default"))) void hc_simt_stats_q(void* p, uint32_t N, uint32_t bstep, int Q, int thr, uint64_t* out)
{
    HcScene* s = static_cast<HcScene*>(p);
    SceneView sc{s->nodes32.data(), s->triPos.data(), s->triNrm.data(), {0, 0, 0}, {0, 0, 0}};
    {
        const float* w = reinterpret_cast<const float*>(&s->nodes[0]);
        for (int a = 0; a < 3; ++a) { sc.rootLo[a] = min_(w[a], w[6 + a]); sc.rootHi[a] = max_(w[3 + a], w[9 + a]); }
    }
    const uint32_t nb = N / 4;
    uint64_t waves = 0, p1 = 0, p1live = 0, p2 = 0, p2lanes = 0, heaviest = 0, total = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : waves, p1, p1live, p2, p2lanes, total) reduction(max : heaviest)
    for (int64_t bzi = 0; bzi < (int64_t)nb; bzi += bstep) {
        for (uint32_t byi = 0; byi < nb; ++byi) for (uint32_t bxi = 0; bxi < nb; ++bxi) {
            struct Lane { Ray r; Hit best; int32_t node; int sp; int32_t stack[64]; int32_t ql[8]; float qt[8]; int qn; };
            static thread_local Lane L[64];
            for (int t = 0; t < 64; ++t) {
                Lane& l = L[t];
                const uint32_t ix = bxi * 4 + t % 4, iy = byi * 4 + (t / 4) % 4, iz = (uint32_t)bzi * 4 + t / 16;
                ray_origin(N, ix, iy, iz, l.r.ox, l.r.oy, l.r.oz);
                l.node = -1; l.qn = 0;
                if (origin_leaves_root(l.r.ox, l.r.oy, l.r.oz, sc.rootLo, sc.rootHi)) continue;
                finish_ray_reference(l.r);
                l.best.t = kTMax; l.best.k = 0xffffffffu; l.best.leaf = -1; l.best.b1 = l.best.b2 = 0;
                l.stack[0] = -1; l.sp = 1; l.node = 0;
            }
            waves++;
            uint64_t my1 = 0, my2 = 0;
            for (;;) {
                int liveNow = 0, holders = 0, full = 0;
                for (int t = 0; t < 64; ++t) {
                    Lane& l = L[t];
                    if (l.node >= 0) {
                        liveNow++;
                        F4 q0, q1, q2; int32_t c0, c1;
                        load_node(sc.nodes, l.node, q0, q1, q2, c0, c1);
                        float tn0, tn1;
                        bool h0 = slab(l.r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tn0) && tn0 <= l.best.t;
                        bool h1 = slab(l.r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tn1) && tn1 <= l.best.t;
                        if (h0 && c0 < 0) { l.ql[l.qn] = ~c0; l.qt[l.qn++] = tn0; }
                        if (h1 && c1 < 0) { l.ql[l.qn] = ~c1; l.qt[l.qn++] = tn1; }
                        h0 = h0 && c0 >= 0; h1 = h1 && c1 >= 0;
                        const bool both = h0 && h1, swap = tn1 < tn0;
                        if (both) l.stack[l.sp++] = swap ? c0 : c1;
                        if (h0 || h1) l.node = (h0 && !(both && swap)) ? c0 : c1;
                        else l.node = l.stack[--l.sp];
                    }
                    if (l.qn) holders++;
                    if (l.qn > Q - 2) full++;
                }
                if (liveNow) { my1++; p1live += liveNow; }
                int stillLive = 0;
                for (int t = 0; t < 64; ++t) if (L[t].node >= 0) stillLive++;
                if (holders && (full || holders >= thr || !stillLive)) {
                    for (;;) {
                        int n = 0;
                        for (int t = 0; t < 64; ++t) {
                            Lane& l = L[t];
                            if (!l.qn) continue;
                            n++;
                            l.qn--;
                            if (l.qt[l.qn] <= l.best.t) leaf_reference(l.r, sc.triPos, l.ql[l.qn], l.best);
                        }
                        if (!n) break;
                        my2++; p2lanes += n;
                    }
                } else if (!stillLive) break;
            }
            p1 += my1; p2 += my2;
            const uint64_t cost = 70 * my1 + 150 * my2;
            total += cost;
            if (cost > heaviest) heaviest = cost;
        }
    }
    out[0] = waves; out[1] = p1; out[2] = p1live; out[3] = p2; out[4] = p2lanes; out[5] = heaviest; out[6] = total;
}

// display pass (N3): the product's UpdateFrame constants and per-pixel march on the host
__attribute__((visibility("default"))) int hc_update_frame(const float* bound, const float* ps, const float* eye, const float* vp,
                                                           float w, float h, float* cb22)
{
    RayCastCB cb;
    if (!update_frame(bound, ps, eye, vp, w, h, cb)) return 1;
    memcpy(cb22, cb.lightPt, 12); memcpy(cb22 + 3, cb.eyePt, 12); memcpy(cb22 + 6, cb.screenToLocal, 64);
    return 0;
}

__attribute__((visibility("default"))) void hc_render(const uint8_t* grid, uint32_t N, const float* cb22, uint32_t width,
                                                      uint32_t height, uint8_t* rgba8)
{
    RayCastCB cb;
    memcpy(cb.lightPt, cb22, 12); memcpy(cb.eyePt, cb22 + 3, 12); memcpy(cb.screenToLocal, cb22 + 6, 64);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t py = 0; py < (int64_t)height; ++py)
        for (uint32_t px = 0; px < width; ++px) {
            float c[4];
            raycast_pixel(cb, grid, N, (float)px + 0.5f, (float)py + 0.5f, c);
            for (int k = 0; k < 4; ++k) {
                float v = c[k];
                if (!(v > 0.0f)) v = 0.0f;
                if (v > 1.0f) v = 1.0f;
                rgba8[((size_t)py * width + px) * 4 + k] = (uint8_t)(v * 255.0f + 0.5f);
            }
        }
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// The host side's decisions (dxrvoxelizer_amd/csrc/dxv_policy.h: pure functions, the same header the library compiles), for
// tests/test_policy.py: the launch-history rules of the lists and the work queue as a table walked on the CPU.
// ---------------------------------------------------------------------------------------------
#include "../../dxrvoxelizer_amd/csrc/dxv_policy.h"
struct HcListsState { int32_t optLists, optListRes, listOpt, listState; uint32_t listRes, numTris, launchesOfScene, refitted, floorTried; uint64_t listEntries; };
static ListsState hc_ls(const HcListsState* h)
{
    ListsState s{};
    s.optLists = h->optLists; s.optListRes = h->optListRes; s.listOpt = h->listOpt; s.listState = h->listState; s.listRes = h->listRes;
    s.numTris = h->numTris; s.launchesOfScene = h->launchesOfScene; s.refitted = h->refitted != 0; s.floorTried = h->floorTried != 0; s.listEntries = h->listEntries;
    return s;
}
extern "C" {
__attribute__((visibility("default"))) int hc_lists_step(const HcListsState* h, uint64_t voxels, int relaunch) { return (int)lists_step(hc_ls(h), voxels, relaunch != 0); }
__attribute__((visibility("default"))) int hc_lists_used(const HcListsState* h, int relaunch) { return lists_used(hc_ls(h), relaunch != 0) ? 1 : 0; }
__attribute__((visibility("default"))) int hc_lists_static_fine(const HcListsState* h, uint32_t floorNow) { return lists_static_scene_takes_fine_map(hc_ls(h), floorNow) ? 1 : 0; }
__attribute__((visibility("default"))) uint32_t hc_lists_base_map(uint32_t numTris, int optListRes) { return lists_base_map(numTris, optListRes); }
__attribute__((visibility("default"))) uint32_t hc_lists_recount_on(uint32_t R, uint64_t entries, int oneLaunch, int optListRes, int coarser) { return lists_recount_on(R, entries, oneLaunch != 0, optListRes, coarser != 0); }
__attribute__((visibility("default"))) int hc_lists_sample_first(uint32_t numTris, int optListRes) { return lists_sample_first(numTris, optListRes) ? 1 : 0; }
__attribute__((visibility("default"))) uint32_t hc_lists_sample_stride(void) { return kListsSampleStride; }
__attribute__((visibility("default"))) uint32_t hc_queue_waves_sevenths(uint32_t numTris, uint32_t R, uint32_t N) { return queue_waves_sevenths(numTris, R, N); }
__attribute__((visibility("default"))) int hc_lists_over_the_caps(uint64_t entries, uint32_t numTris) { return lists_over_the_caps(entries, numTris) ? 1 : 0; }
__attribute__((visibility("default"))) int hc_lists_pay(uint64_t voxels, uint64_t entries, uint32_t R) { return lists_pay_on_first_launch(voxels, entries, R) ? 1 : 0; }
__attribute__((visibility("default"))) int hc_queue_policy(int optPlan, int optDispatch, int ptrExposed, uint64_t keptSig, uint64_t lensSig, uint32_t queuedBricks, uint64_t sig, uint64_t voxels)
{
    QueueState q{};
    q.optPlan = optPlan; q.optDispatch = optDispatch; q.ptrExposed = ptrExposed != 0; q.keptSig = keptSig; q.lensSig = lensSig; q.queuedBricks = queuedBricks;
    q.optPrepared = 1; q.prepared = false;
    return (int)queue_policy(q, sig, voxels);
}
__attribute__((visibility("default"))) int hc_far_map_build_now(int haveForScene, uint32_t boxLaunchesOfScene) { return far_map_build_now(haveForScene != 0, boxLaunchesOfScene) ? 1 : 0; }
// ... with a queue PREPARED for the launch (dxv_prepare_launch) in the context
__attribute__((visibility("default"))) int hc_queue_policy_prepared(int optPlan, int optDispatch, int optPrepared, int prepared, int ptrExposed, uint64_t keptSig, uint64_t lensSig,
                                                                    uint32_t queuedBricks, uint64_t sig, uint64_t voxels)
{
    QueueState q{};
    q.optPlan = optPlan; q.optDispatch = optDispatch; q.ptrExposed = ptrExposed != 0; q.keptSig = keptSig; q.lensSig = lensSig; q.queuedBricks = queuedBricks;
    q.optPrepared = optPrepared; q.prepared = prepared != 0;
    return (int)queue_policy(q, sig, voxels);
}
}
