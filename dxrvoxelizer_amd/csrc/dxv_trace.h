// dxv_trace.h -- per-ray traversal of the LBVH (one ray = one voxel = one thread).
//
// Replaces TraceRay(g_scene, RAY_FLAG_NONE, ~0, 0, 1, 0, ray, payload)
// (Content/Shaders/DXRVoxelizer.hlsl:80) and the driver traversal behind it.
//
// Canonical acceptance of triangle k for a ray (topology independent):
//   reference mode: the triangle's own padded box passes slab() with entry tn, the watertight
//     test reports 0 < t < TMax, and tn <= t.  Closest hit = lexicographic min of (t, k).
//   parity mode: the box passes slab_parity() and the watertight test with fill rule reports
//     t > 0; every accepted triangle counts once.
// Because slab() is monotone under box inclusion, culling a node whose entry distance exceeds
// the current closest t can never drop an acceptable triangle.
//
// Leaves never enter the stack: a hit leaf child is intersected while its parent is visited
// (its entry distance is at hand there), so the stack holds internal nodes only: at most one
// entry per level plus the sentinel, i.e. treeHeight entries always suffice.
#pragma once
#include "dxv_math.h"

namespace dxv {

struct Hit {
    float t, b1, b2;
    uint32_t k;     // triangle index in the caller's index buffer; 0xffffffff = miss
    int32_t leaf;   // position in Morton order
};

// Stack policy: entry e of this thread lives at base[e * stride] (LDS column on the device).
struct StridedStack {
    int32_t* base;
    int stride;
    DXV_HD void put(int e, int32_t v) const { base[e * stride] = v; }
    DXV_HD int32_t get(int e) const { return base[e * stride]; }
    DXV_HD int32_t getv(int e) const { return *const_cast<volatile int32_t*>(base + e * stride); }   // a load the compiler may not forward from a store
};

DXV_HD float half_bits_to_float(uint32_t h16)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (float)__builtin_bit_cast(_Float16, (uint16_t)h16);      // v_cvt_f32_f16, exact
#else
    return half_to_float((uint16_t)h16);
#endif
}

// Compressed node (32 B = two 16-B loads) -> twelve float planes + two links.  32-bit byte offsets
// from a wave-uniform base: the device build addresses nodes and triangles as
// base(SGPR pair) + offset(VGPR).
struct NodePlanes { float b[12]; int32_t c0, c1; };
DXV_HD NodePlanes load_node(const Node32* nodes, int32_t i)
{
    const uint32_t* p = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(nodes) + ((uint32_t)i << 5));
    struct alignas(16) U4 { uint32_t x, y, z, w; };
    const U4 a = *reinterpret_cast<const U4*>(p), d = *reinterpret_cast<const U4*>(p + 4);
    NodePlanes n;                       // b: child 0 lo.xyz hi.xyz, child 1 lo.xyz hi.xyz (Node32 is axis-major)
    n.b[0] = half_bits_to_float(a.x & 0xffffu); n.b[6] = half_bits_to_float(a.x >> 16);    // x lo
    n.b[3] = half_bits_to_float(a.y & 0xffffu); n.b[9] = half_bits_to_float(a.y >> 16);    // x hi
    n.b[1] = half_bits_to_float(a.z & 0xffffu); n.b[7] = half_bits_to_float(a.z >> 16);    // y lo
    n.b[4] = half_bits_to_float(a.w & 0xffffu); n.b[10] = half_bits_to_float(a.w >> 16);   // y hi
    n.b[2] = half_bits_to_float(d.x & 0xffffu); n.b[8] = half_bits_to_float(d.x >> 16);    // z lo
    n.b[5] = half_bits_to_float(d.y & 0xffffu); n.b[11] = half_bits_to_float(d.y >> 16);   // z hi
    n.c0 = (int32_t)d.z; n.c1 = (int32_t)d.w;
    return n;
}

// The six plane words of node i as stored (x lo, x hi, y lo, y hi, z lo, z hi; child 0 in the low
// half, child 1 in the high half) and its links.
struct NodeWords { uint32_t w[6]; int32_t c0, c1; };
DXV_HD NodeWords load_node_words(const Node32* nodes, int32_t i)
{
    const uint32_t* p = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(nodes) + ((uint32_t)i << 5));
    struct alignas(16) U4 { uint32_t x, y, z, w; };
    const U4 a = *reinterpret_cast<const U4*>(p), d = *reinterpret_cast<const U4*>(p + 4);
    return NodeWords{{a.x, a.y, a.z, a.w, d.x, d.y}, (int32_t)d.z, (int32_t)d.w};
}

DXV_HD TriPos load_tri(const TriPos* tris, int32_t leaf)
{
    return *reinterpret_cast<const TriPos*>(reinterpret_cast<const char*>(tris) + (uint32_t)leaf * 48u);
}

// The triangle's own (exact, canonical) padded box decides candidacy: node boxes are outward
// rounded supersets and only steer the walk.
DXV_HD void leaf_reference(Ray& r, const TriPos* tris, int32_t leaf, Hit& best)
{
    const TriPos tp = load_tri(tris, leaf);
    float lo[3], hi[3], tn;
    tri_box(tp.v0, tp.v1, tp.v2, lo, hi);
    if (!(slab(r, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], tn) && tn <= best.t)) return;
    if (r.kz < 0) ray_shear(r);
    float t, b1, b2;
    if (!tri_test<false>(r, tp.v0, tp.v1, tp.v2, t, b1, b2)) return;
    if (tn > t) return;
    const uint32_t k = __builtin_bit_cast(uint32_t, tp.v0.w);
    if (t < best.t || (t == best.t && k < best.k)) { best.t = t; best.b1 = b1; best.b2 = b2; best.k = k; best.leaf = leaf; }
}

// The same step for the lists kernel: the two barycentric divisions wait until the closest hit is known (b1, b2 hold the
// undivided V, W meanwhile, `det` their divisor; finish_hit divides once).
DXV_HD void leaf_reference_deferred(Ray& r, const TriPos* tris, int32_t leaf, Hit& best, float& bestDet)
{
    const TriPos tp = load_tri(tris, leaf);
    // the slot travels with the triangle's class bits (normal_class: what the closest hit needs afterwards); joined here,
    // where both are at hand, so that one register instead of two lives through the test
    const int32_t tagged = leaf | (int32_t)__builtin_bit_cast(uint32_t, tp.v1.w);
    float lo[3], hi[3], tn;
    tri_box(tp.v0, tp.v1, tp.v2, lo, hi);
    if (!(slab(r, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], tn) && tn <= best.t)) return;
    float t, V, W, det;                                                 // (the caller has sheared the ray: ray_shear)
    if (!tri_test<false, true>(r, tp.v0, tp.v1, tp.v2, t, V, W, &det)) return;
    if (tn > t) return;
    const uint32_t k = __builtin_bit_cast(uint32_t, tp.v0.w);
    if (t < best.t || (t == best.t && k < best.k)) { best.t = t; best.b1 = V; best.b2 = W; bestDet = det; best.k = k; best.leaf = tagged; }
}
// ... and with the closest hit's rarely read words -- the undivided V, W, their divisor det and the triangle's index (read only on an exact
// tie of t) -- in the thread's LDS column (words hitAt .. hitAt + 3) instead of four vector registers held through the scan and the
// triangle rounds: written when a hit is accepted, read once behind the walk (shade_reference_lds).  bestLeaf == -1: no hit yet.
// Same tests, same comparisons, same winner as leaf_reference_deferred.
template <class Stack>
DXV_HD void leaf_reference_deferred_lds(Ray& r, const TriPos* tris, int32_t leaf, float& bestT, int32_t& bestLeaf, const Stack& stk, int hitAt)
{
    const TriPos tp = load_tri(tris, leaf);
    const int32_t tagged = leaf | (int32_t)__builtin_bit_cast(uint32_t, tp.v1.w);
    float lo[3], hi[3], tn;
    tri_box(tp.v0, tp.v1, tp.v2, lo, hi);
    if (!(slab(r, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], tn) && tn <= bestT)) return;
    float t, V, W, det;                                                 // (the caller has sheared the ray: ray_shear)
    if (!tri_test<false, true>(r, tp.v0, tp.v1, tp.v2, t, V, W, &det)) return;
    if (tn > t) return;
    const uint32_t k = __builtin_bit_cast(uint32_t, tp.v0.w);
    bool take = t < bestT;
    if (!take && t == bestT) take = k < (uint32_t)stk.get(hitAt + 3);   // (t < kTMax strictly: an equal bestT is a hit's, its index is in the column)
    if (take) {
        bestT = t; bestLeaf = tagged;
        stk.put(hitAt, __builtin_bit_cast(int32_t, V)); stk.put(hitAt + 1, __builtin_bit_cast(int32_t, W));
        stk.put(hitAt + 2, __builtin_bit_cast(int32_t, det)); stk.put(hitAt + 3, (int32_t)k);
    }
}
// ... and with NOTHING of the closest hit kept but its t and its tagged slot: the triangle's index, needed on an exact tie of t only, is read
// from the hit triangle's record then, and V, W, det -- needed for the few hits whose triangle has no class (shade_reference_again) -- are
// computed again behind the walk from the same ray and the same record: the same bits.  Four registers less through scan and rounds.
DXV_HD void leaf_reference_min(Ray& r, const TriPos* tris, int32_t leaf, float& bestT, int32_t& bestLeaf)
{
    const TriPos tp = load_tri(tris, leaf);
    const int32_t tagged = leaf | (int32_t)__builtin_bit_cast(uint32_t, tp.v1.w);
    float lo[3], hi[3], tn;
    tri_box(tp.v0, tp.v1, tp.v2, lo, hi);
    if (!(slab(r, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], tn) && tn <= bestT)) return;
    float t, V, W, det;                                                 // (the caller has sheared the ray: ray_shear)
    if (!tri_test<false, true>(r, tp.v0, tp.v1, tp.v2, t, V, W, &det)) return;
    if (tn > t) return;
    bool take = t < bestT;
    if (!take && t == bestT)                                            // (t < kTMax strictly: an equal bestT is a hit's)
        take = __builtin_bit_cast(uint32_t, tp.v0.w) < __builtin_bit_cast(uint32_t, tris[bestLeaf & (int32_t)((1u << kClassShift) - 1u)].v0.w);
    if (take) { bestT = t; bestLeaf = tagged; }
}
DXV_HD void finish_hit(Hit& best, float bestDet)
{
    if (best.k != 0xffffffffu) { best.b1 = best.b1 / bestDet; best.b2 = best.b2 / bestDet; }
}

// Optional per-ray counters (tests / tuning only; compiled out of the shipped kernels).
struct TraceStats { uint32_t nodes, leaves, maxsp; };

// Returns false when the stack capacity was exceeded (caller reports the error).
// Entry 0 of the stack holds a negative sentinel, so "pop" needs no emptiness test and the loop has
// a single exit (node < 0); capacity for real entries is cap - 1.
template <class Stack, bool STATS = false>
DXV_HD bool trace_reference(Ray& r, const Node32* nodes, const TriPos* tris, const Stack& stk, int cap, Hit& best,
                            TraceStats* st = nullptr)
{
    best.t = kTMax; best.b1 = 0.0f; best.b2 = 0.0f; best.k = 0xffffffffu; best.leaf = -1;
    stk.put(0, -1);
    int sp = 1;
    bool ok = true;
    int32_t node = 0;
    while (node >= 0) {
        const NodePlanes n = load_node(nodes, node);
        const int32_t c0 = n.c0, c1 = n.c1;
        if (STATS) st->nodes++;
        float tn0, tn1;
        bool h0 = slab(r, n.b[0], n.b[1], n.b[2], n.b[3], n.b[4], n.b[5], tn0) && tn0 <= best.t;
        bool h1 = slab(r, n.b[6], n.b[7], n.b[8], n.b[9], n.b[10], n.b[11], tn1) && tn1 <= best.t;
        const bool l0 = h0 && c0 < 0, l1 = h1 && c1 < 0;
        if (STATS) st->leaves += l0 + l1;
        if (l0 || l1) {
            // lanes with a leaf in either child test it together; a second leaf is the rare case
            leaf_reference(r, tris, l0 ? ~c0 : ~c1, best);
            if (l0 && l1) leaf_reference(r, tris, ~c1, best);
        }
        h0 = h0 && c0 >= 0 && tn0 <= best.t;
        h1 = h1 && c1 >= 0 && tn1 <= best.t;
        const bool both = h0 && h1;
        const bool swap = tn1 < tn0;
        if (both) {
            if (sp >= cap) { ok = false; break; }
            stk.put(sp++, swap ? c0 : c1);
            if (STATS && sp - 1 > (int)st->maxsp) st->maxsp = (uint32_t)(sp - 1);
        }
        if (h0 || h1) node = (h0 && !(both && swap)) ? c0 : c1;
        else node = stk.get(--sp);
    }
    return ok;
}

// wave-level vote.  Device: true when the predicate holds in any active lane of the wavefront;
// host (tests/hostcheck runs one ray at a time): the ray's own predicate.
DXV_HD bool wave_any(bool x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(x) != 0ull;
#else
    return x;
#endif
}

// wave_any(a && b): the AND of two ballots (lane-wise) costs two compares and one scalar AND, the
// ballot of the combined predicate an extra select and compare.
DXV_HD bool wave_any_both(bool a, bool b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (__builtin_amdgcn_ballot_w64(a) & __builtin_amdgcn_ballot_w64(b)) != 0ull;
#else
    return a && b;
#endif
}

constexpr int32_t kNodeOverflow = -2;   // a ray whose column ran out (-1 = walk finished)

// Postponed-leaf traversal (default).  Walking internal nodes is cheap and coherent across the
// wave; triangle tests are long and, done on the spot, run with a handful of lanes.  So hit leaf
// children are only QUEUED (leaf index in the same LDS column as the stack: stack grows up from
// entry 0, queue down from entry cap-1) while every lane keeps walking; the wave tests queued
// triangles together when some lane's column is nearly full or no lane can walk any more.  Order of
// tests does not matter for the result (closest = min (t, k)); a queued triangle's own box entry
// distance is recomputed from its vertices exactly as the refit computed the leaf box.
// One internal-node visit of the postponed-leaf walk: two slab tests, hit leaves queued, near child
// next, far child pushed.  The six plane words (Node32 order: child 0 in the low half, child 1 in
// the high half) and two links arrive by value so that the device build can feed them from SGPRs
// (wave-uniform visit, scalar load) or VGPRs (divergent visit).  Reference rule: the caller has
// sorted them by the ray -- in* is the plane the ray enters a box through (lo where the direction
// component is positive, hi where it is negative), out* the other: slab_sorted.  Parity rule:
// in* = lo, out* = hi.
template <bool PARITY, class Stack>
DXV_HD void node_step(const Ray& r, uint32_t inx, uint32_t outx, uint32_t iny, uint32_t outy, uint32_t inz, uint32_t outz,
                      int32_t c0, int32_t c1, const Stack& stk, int cap, float bestT, int32_t& node, int& sp, int& qn)
{
    float tn0 = 0.0f, tn1 = 0.0f;
    bool h0, h1;
    if (PARITY) {
        h0 = slab_parity(r, half_bits_to_float(iny & 0xffffu), half_bits_to_float(inz & 0xffffu), half_bits_to_float(outx & 0xffffu),
                         half_bits_to_float(outy & 0xffffu), half_bits_to_float(outz & 0xffffu));
        h1 = slab_parity(r, half_bits_to_float(iny >> 16), half_bits_to_float(inz >> 16), half_bits_to_float(outx >> 16),
                         half_bits_to_float(outy >> 16), half_bits_to_float(outz >> 16));
    } else {
        h0 = slab_sorted(r, half_bits_to_float(inx & 0xffffu), half_bits_to_float(iny & 0xffffu), half_bits_to_float(inz & 0xffffu),
                         half_bits_to_float(outx & 0xffffu), half_bits_to_float(outy & 0xffffu),
                         half_bits_to_float(outz & 0xffffu), tn0) && tn0 <= bestT;
        h1 = slab_sorted(r, half_bits_to_float(inx >> 16), half_bits_to_float(iny >> 16), half_bits_to_float(inz >> 16),
                         half_bits_to_float(outx >> 16), half_bits_to_float(outy >> 16), half_bits_to_float(outz >> 16), tn1) &&
             tn1 <= bestT;
    }
    // (Unconditional stores to the next free slots + an unconditional pop were tried to get rid of
    // the exec-mask juggling around these small blocks: 15-20 % SLOWER on MI355X, the extra LDS
    // operations cost more than the branches.)
    if (h0 && c0 < 0) stk.put(cap - 1 - qn++, ~c0);
    if (h1 && c1 < 0) stk.put(cap - 1 - qn++, ~c1);
    h0 = h0 && c0 >= 0;
    h1 = h1 && c1 >= 0;
    const bool both = h0 && h1;
    const bool swap = tn1 < tn0;
    if (both) stk.put(sp++, swap ? c0 : c1);
    if (h0 || h1) node = (h0 && !(both && swap)) ? c0 : c1;
    else node = stk.get(--sp);
}

#if defined(__HIP_DEVICE_COMPILE__)
// 32-B node through the scalar cache into 8 SGPRs (nodes are read-only during the kernel).
// Four 64-bit outputs: plain scalar-pair operands, which hipcc tracks reliably (512- and 128-bit
// SGPR tuples as asm outputs were mis-tracked by ROCm 7.2's hipcc: elements folded together).
struct NodeSgpr { uint64_t w[4]; };
__device__ __forceinline__ NodeSgpr load_node_scalar(const Node32* nodes, int32_t uniformIndex)
{
    const char* p = reinterpret_cast<const char*>(nodes) + ((uint64_t)(uint32_t)uniformIndex << 5);
    NodeSgpr n;
    asm volatile("s_load_dwordx2 %0, %4, 0x0\n\ts_load_dwordx2 %1, %4, 0x8\n\ts_load_dwordx2 %2, %4, 0x10\n\t"
                 "s_load_dwordx2 %3, %4, 0x18\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(n.w[0]), "=&s"(n.w[1]), "=&s"(n.w[2]), "=&s"(n.w[3]) : "s"(p) : "memory");
    return n;
}
#endif

// The walk shared by both occupancy rules.  `Leaf` is called once per queued triangle:
// leaf(leafIndex, TriPos).
template <bool PARITY, class Stack, class Leaf, bool STATS = false>
DXV_HD bool walk_queued(const Ray& r, const Node32* nodes, const TriPos* tris, const Stack& stk, int cap, const float& bestT,
                        Leaf&& leaf, TraceStats* st = nullptr)
{
    stk.put(0, -1);
    int sp = 1, qn = 0;
    int32_t node = 0;
    // The plane of each axis a box is entered through: hi where the direction is negative.  The
    // radial direction has the signs of the origin (never zero); the parity ray enters through lo.
    const bool negx = !PARITY && r.ox < 0.0f, negy = !PARITY && r.oy < 0.0f, negz = !PARITY && r.oz < 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    // A brick lies in one octant of the grid unless it straddles a centre plane (grid sizes that
    // are not a multiple of twice the brick): then the three choices are wave-uniform and a
    // wave-uniform visit sorts the planes on the scalar unit as well.
    const uint64_t active = __builtin_amdgcn_ballot_w64(true);
    const uint64_t bx = __builtin_amdgcn_ballot_w64(negx), by = __builtin_amdgcn_ballot_w64(negy), bz = __builtin_amdgcn_ballot_w64(negz);
    const bool octant = (bx == 0ull || bx == active) && (by == 0ull || by == active) && (bz == 0ull || bz == active);
    const bool unx = bx != 0ull, uny = by != 0ull, unz = bz != 0ull;
#endif
    for (;;) {
        // A step needs one free stack slot and two free queue slots: sp + qn + 3 <= cap.  The wave
        // flushes the queue whenever a queue is what is in the way (below), so no room here means
        // the stack alone is too deep for this column: stop the ray and report it.
        if (node >= 0 && sp + qn + 3 > cap) node = kNodeOverflow;
        if (node >= 0) {
            if (STATS) st->nodes++;
#if defined(__HIP_DEVICE_COMPILE__)
            // all lanes that are still walking sit on the same node (40-50 % of the visits): one
            // scalar load instead of 64 lanes x 32 B through the vector L1
            const int32_t n0 = __builtin_amdgcn_readfirstlane(node);
            if (octant && __builtin_amdgcn_ballot_w64(node != n0) == 0ull) {
                const NodeSgpr n = load_node_scalar(nodes, n0);
                const uint32_t xl = (uint32_t)n.w[0], xh = (uint32_t)(n.w[0] >> 32), yl = (uint32_t)n.w[1], yh = (uint32_t)(n.w[1] >> 32);
                const uint32_t zl = (uint32_t)n.w[2], zh = (uint32_t)(n.w[2] >> 32);
                node_step<PARITY>(r, unx ? xh : xl, unx ? xl : xh, uny ? yh : yl, uny ? yl : yh, unz ? zh : zl, unz ? zl : zh,
                                  (int32_t)(uint32_t)n.w[3], (int32_t)(uint32_t)(n.w[3] >> 32), stk, cap, bestT, node, sp, qn);
            } else
#endif
            {
                const NodeWords n = load_node_words(nodes, node);
                node_step<PARITY>(r, negx ? n.w[1] : n.w[0], negx ? n.w[0] : n.w[1], negy ? n.w[3] : n.w[2], negy ? n.w[2] : n.w[3],
                                  negz ? n.w[5] : n.w[4], negz ? n.w[4] : n.w[5], n.c0, n.c1, stk, cap, bestT, node, sp, qn);
            }
            if (STATS && sp - 1 > (int)st->maxsp) st->maxsp = (uint32_t)(sp - 1);
        }
        const bool walking = wave_any(node >= 0);
        if (walking && !wave_any_both(sp + qn + 3 > cap, qn > 0)) continue;
        for (int i = 0; wave_any(i < qn); ++i) {
            if (i < qn) {
                const int32_t l = stk.get(cap - 1 - i);
                if (STATS) st->leaves++;
                leaf(l, load_tri(tris, l));
            }
        }
        qn = 0;
        if (!walking) break;
    }
    return node != kNodeOverflow;
}

// reference rule: candidacy by the triangle's own exact box, closest = min (t, k)
struct LeafReference {
    Ray& r;
    Hit& best;
    DXV_HD void operator()(int32_t leaf, const TriPos& tp) const
    {
        float lo[3], hi[3], tn;
        tri_box(tp.v0, tp.v1, tp.v2, lo, hi);
        if (!(slab(r, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], tn) && tn <= best.t)) return;
        if (r.kz < 0) ray_shear(r);
        float t, b1, b2;
        if (!(tri_test<false>(r, tp.v0, tp.v1, tp.v2, t, b1, b2) && tn <= t)) return;
        const uint32_t k = __builtin_bit_cast(uint32_t, tp.v0.w);
        if (t < best.t || (t == best.t && k < best.k)) { best.t = t; best.b1 = b1; best.b2 = b2; best.k = k; best.leaf = leaf; }
    }
};

// parity rule: every accepted triangle counts once
struct LeafParity {
    const Ray& r;
    uint32_t& count;
    DXV_HD void operator()(int32_t, const TriPos& tp) const
    {
        float lo[3], hi[3];
        tri_box(tp.v0, tp.v1, tp.v2, lo, hi);
        if (!slab_parity(r, lo[1], lo[2], hi[0], hi[1], hi[2])) return;
        float t, b1, b2;
        if (tri_test<true>(r, tp.v0, tp.v1, tp.v2, t, b1, b2)) count++;
    }
};

template <class Stack, bool STATS = false>
DXV_HD bool trace_reference_q(Ray& r, const Node32* nodes, const TriPos* tris, const Stack& stk, int cap, Hit& best,
                              TraceStats* st = nullptr)
{
    best.t = kTMax; best.b1 = 0.0f; best.b2 = 0.0f; best.k = 0xffffffffu; best.leaf = -1;
    return walk_queued<false, Stack, LeafReference, STATS>(r, nodes, tris, stk, cap, best.t, LeafReference{r, best}, st);
}

// ------------------------------------------------------------------------------------------
// The same walk over the WIDE nodes (dxv_types.h Node64): up to four boxes per visit, so a ray
// waits for half as many dependent node fetches.  A visit queues the leaves it hits, goes on
// with the nearest internal child it hits and pushes the others (their order does not change the
// result and, measured, not the number of visits either).  Each of the four children ends up in
// at most one slot of the column, so a step needs sp + qn + 4 < cap.
// Plane words arrive as 64-bit pairs {children 0,1 | children 2,3}, already sorted by the ray.
// ------------------------------------------------------------------------------------------
constexpr int kWideRoom = 5;

template <class Stack>
DXV_HD void wide_step(const Ray& r, uint64_t inx, uint64_t outx, uint64_t iny, uint64_t outy, uint64_t inz, uint64_t outz,
                      int32_t c0, int32_t c1, int32_t c2, int32_t c3, const Stack& stk, int cap, float bestT, int32_t& node,
                      int& sp, int& qn)
{
    const int32_t c[4] = {c0, c1, c2, c3};
    float key[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; ++k) {
        float tn;
        const bool h = slab_sorted(r, half_bits_to_float((uint32_t)(inx >> (16 * k)) & 0xffffu),
                                   half_bits_to_float((uint32_t)(iny >> (16 * k)) & 0xffffu),
                                   half_bits_to_float((uint32_t)(inz >> (16 * k)) & 0xffffu),
                                   half_bits_to_float((uint32_t)(outx >> (16 * k)) & 0xffffu),
                                   half_bits_to_float((uint32_t)(outy >> (16 * k)) & 0xffffu),
                                   half_bits_to_float((uint32_t)(outz >> (16 * k)) & 0xffffu), tn) &&
                       tn <= bestT;
        if (h && c[k] < 0) stk.put(cap - 1 - qn++, ~c[k]);                  // leaf: postponed
        key[k] = (h && c[k] >= 0) ? tn : __builtin_inff();                  // internal: entry distance
    }
    int32_t next = c[0];
    float nearest = key[0];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 1; k < 4; ++k)
        if (key[k] < nearest) { nearest = key[k]; next = c[k]; }
    if (nearest < __builtin_inff()) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 0; k < 4; ++k)
            if (key[k] < __builtin_inff() && c[k] != next) stk.put(sp++, c[k]);   // links of one node are distinct
        node = next;
    } else node = stk.get(--sp);
}

#if defined(__HIP_DEVICE_COMPILE__)
struct WideSgpr { uint64_t w[8]; };
__device__ __forceinline__ WideSgpr load_wide_scalar(const Node64* nodes, int32_t uniformIndex)
{
    const char* p = reinterpret_cast<const char*>(nodes) + ((uint64_t)(uint32_t)uniformIndex << 6);
    WideSgpr n;
    asm volatile("s_load_dwordx2 %0, %8, 0x0\n\ts_load_dwordx2 %1, %8, 0x8\n\ts_load_dwordx2 %2, %8, 0x10\n\t"
                 "s_load_dwordx2 %3, %8, 0x18\n\ts_load_dwordx2 %4, %8, 0x20\n\ts_load_dwordx2 %5, %8, 0x28\n\t"
                 "s_load_dwordx2 %6, %8, 0x30\n\ts_load_dwordx2 %7, %8, 0x38\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(n.w[0]), "=&s"(n.w[1]), "=&s"(n.w[2]), "=&s"(n.w[3]), "=&s"(n.w[4]), "=&s"(n.w[5]), "=&s"(n.w[6]),
                   "=&s"(n.w[7])
                 : "s"(p) : "memory");
    return n;
}
#endif

template <class Stack, class Leaf, bool STATS = false>
DXV_HD bool walk_queued_wide(const Ray& r, const Node64* nodes, const TriPos* tris, const Stack& stk, int cap, const float& bestT,
                             Leaf&& leaf, TraceStats* st = nullptr)
{
    stk.put(0, -1);
    int sp = 1, qn = 0;
    int32_t node = 0;
    const bool negx = r.ox < 0.0f, negy = r.oy < 0.0f, negz = r.oz < 0.0f;   // as in walk_queued
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t active = __builtin_amdgcn_ballot_w64(true);
    const uint64_t bx = __builtin_amdgcn_ballot_w64(negx), by = __builtin_amdgcn_ballot_w64(negy), bz = __builtin_amdgcn_ballot_w64(negz);
    const bool octant = (bx == 0ull || bx == active) && (by == 0ull || by == active) && (bz == 0ull || bz == active);
    const bool unx = bx != 0ull, uny = by != 0ull, unz = bz != 0ull;
#endif
    for (;;) {
        if (node >= 0 && sp + qn + kWideRoom > cap) node = kNodeOverflow;
        if (node >= 0) {
            if (STATS) st->nodes++;
#if defined(__HIP_DEVICE_COMPILE__)
            const int32_t n0 = __builtin_amdgcn_readfirstlane(node);
            if (octant && __builtin_amdgcn_ballot_w64(node != n0) == 0ull) {
                const WideSgpr n = load_wide_scalar(nodes, n0);
                wide_step(r, unx ? n.w[1] : n.w[0], unx ? n.w[0] : n.w[1], uny ? n.w[3] : n.w[2], uny ? n.w[2] : n.w[3],
                          unz ? n.w[5] : n.w[4], unz ? n.w[4] : n.w[5], (int32_t)(uint32_t)n.w[6], (int32_t)(uint32_t)(n.w[6] >> 32),
                          (int32_t)(uint32_t)n.w[7], (int32_t)(uint32_t)(n.w[7] >> 32), stk, cap, bestT, node, sp, qn);
            } else
#endif
            {
                const uint64_t* p = reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(nodes) + ((uint32_t)node << 6));
                struct alignas(16) U2 { uint64_t a, b; };
                const U2 x = *reinterpret_cast<const U2*>(p), y = *reinterpret_cast<const U2*>(p + 2);
                const U2 z = *reinterpret_cast<const U2*>(p + 4), l = *reinterpret_cast<const U2*>(p + 6);
                wide_step(r, negx ? x.b : x.a, negx ? x.a : x.b, negy ? y.b : y.a, negy ? y.a : y.b, negz ? z.b : z.a,
                          negz ? z.a : z.b, (int32_t)(uint32_t)l.a, (int32_t)(uint32_t)(l.a >> 32), (int32_t)(uint32_t)l.b,
                          (int32_t)(uint32_t)(l.b >> 32), stk, cap, bestT, node, sp, qn);
            }
            if (STATS && sp - 1 > (int)st->maxsp) st->maxsp = (uint32_t)(sp - 1);
        }
        const bool walking = wave_any(node >= 0);
        if (walking && !wave_any_both(sp + qn + kWideRoom > cap, qn > 0)) continue;
        for (int i = 0; wave_any(i < qn); ++i) {
            if (i < qn) {
                const int32_t l = stk.get(cap - 1 - i);
                if (STATS) st->leaves++;
                leaf(l, load_tri(tris, l));
            }
        }
        qn = 0;
        if (!walking) break;
    }
    return node != kNodeOverflow;
}

// Hybrid: wave-uniform visits take the four-box node (scalar fetch: half as many dependent
// fetches where the wave walks together), divergent visits the binary node (the wide one would
// double their vector loads).  Both copies are indexed by the binary node, so the walk switches
// from visit to visit.  Host builds (one ray at a time) take the binary node throughout.
template <class Stack, class Leaf>
DXV_HD bool walk_queued_hybrid(const Ray& r, const Node32* nodes, const Node64* wide, const TriPos* tris, const Stack& stk, int cap,
                               const float& bestT, Leaf&& leaf)
{
    stk.put(0, -1);
    int sp = 1, qn = 0;
    int32_t node = 0;
    const bool negx = r.ox < 0.0f, negy = r.oy < 0.0f, negz = r.oz < 0.0f;   // as in walk_queued
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t active = __builtin_amdgcn_ballot_w64(true);
    const uint64_t bx = __builtin_amdgcn_ballot_w64(negx), by = __builtin_amdgcn_ballot_w64(negy), bz = __builtin_amdgcn_ballot_w64(negz);
    const bool octant = (bx == 0ull || bx == active) && (by == 0ull || by == active) && (bz == 0ull || bz == active);
    const bool unx = bx != 0ull, uny = by != 0ull, unz = bz != 0ull;
#else
    (void)wide;
#endif
    for (;;) {
        if (node >= 0 && sp + qn + 3 > cap) node = kNodeOverflow;       // room for a binary step, as in walk_queued
        if (node >= 0) {
#if defined(__HIP_DEVICE_COMPILE__)
            const int32_t n0 = __builtin_amdgcn_readfirstlane(node);
            // (a four-box step needs up to four free slots: taken only while every lane has them)
            if (octant && __builtin_amdgcn_ballot_w64(node != n0) == 0ull && !wave_any(sp + qn + kWideRoom > cap)) {
                const WideSgpr n = load_wide_scalar(wide, n0);
                wide_step(r, unx ? n.w[1] : n.w[0], unx ? n.w[0] : n.w[1], uny ? n.w[3] : n.w[2], uny ? n.w[2] : n.w[3],
                          unz ? n.w[5] : n.w[4], unz ? n.w[4] : n.w[5], (int32_t)(uint32_t)n.w[6], (int32_t)(uint32_t)(n.w[6] >> 32),
                          (int32_t)(uint32_t)n.w[7], (int32_t)(uint32_t)(n.w[7] >> 32), stk, cap, bestT, node, sp, qn);
            } else
#endif
            {
                const NodeWords n = load_node_words(nodes, node);
                node_step<false>(r, negx ? n.w[1] : n.w[0], negx ? n.w[0] : n.w[1], negy ? n.w[3] : n.w[2], negy ? n.w[2] : n.w[3],
                                 negz ? n.w[5] : n.w[4], negz ? n.w[4] : n.w[5], n.c0, n.c1, stk, cap, bestT, node, sp, qn);
            }
        }
        const bool walking = wave_any(node >= 0);
        if (walking && !wave_any_both(sp + qn + 3 > cap, qn > 0)) continue;
        for (int i = 0; wave_any(i < qn); ++i) {
            if (i < qn) {
                const int32_t l = stk.get(cap - 1 - i);
                leaf(l, load_tri(tris, l));
            }
        }
        qn = 0;
        if (!walking) break;
    }
    return node != kNodeOverflow;
}

template <class Stack>
DXV_HD bool trace_reference_h(Ray& r, const Node32* nodes, const Node64* wide, const TriPos* tris, const Stack& stk, int cap, Hit& best)
{
    best.t = kTMax; best.b1 = 0.0f; best.b2 = 0.0f; best.k = 0xffffffffu; best.leaf = -1;
    return walk_queued_hybrid<Stack, LeafReference>(r, nodes, wide, tris, stk, cap, best.t, LeafReference{r, best});
}

template <class Stack, bool STATS = false>
DXV_HD bool trace_reference_w(Ray& r, const Node64* nodes, const TriPos* tris, const Stack& stk, int cap, Hit& best,
                              TraceStats* st = nullptr)
{
    best.t = kTMax; best.b1 = 0.0f; best.b2 = 0.0f; best.k = 0xffffffffu; best.leaf = -1;
    return walk_queued_wide<Stack, LeafReference, STATS>(r, nodes, tris, stk, cap, best.t, LeafReference{r, best}, st);
}

template <class Stack>
DXV_HD bool trace_parity_q(const Ray& r, const Node32* nodes, const TriPos* tris, const Stack& stk, int cap, uint32_t& count)
{
    count = 0;
    const float unused = 0.0f;
    return walk_queued<true, Stack, LeafParity>(r, nodes, tris, stk, cap, unused, LeafParity{r, count});
}

DXV_HD uint32_t leaf_parity(const Ray& r, const TriPos* tris, int32_t leaf)
{
    const TriPos tp = load_tri(tris, leaf);
    float lo[3], hi[3];
    tri_box(tp.v0, tp.v1, tp.v2, lo, hi);                            // exact canonical box of the triangle
    if (!slab_parity(r, lo[1], lo[2], hi[0], hi[1], hi[2])) return 0u;
    float t, b1, b2;
    return tri_test<true>(r, tp.v0, tp.v1, tp.v2, t, b1, b2) ? 1u : 0u;
}

template <class Stack>
DXV_HD bool trace_parity(const Ray& r, const Node32* nodes, const TriPos* tris, const Stack& stk, int cap, uint32_t& count)
{
    count = 0;
    stk.put(0, -1);
    int sp = 1;
    bool ok = true;
    int32_t node = 0;
    while (node >= 0) {
        const NodePlanes n = load_node(nodes, node);
        const int32_t c0 = n.c0, c1 = n.c1;
        bool h0 = slab_parity(r, n.b[1], n.b[2], n.b[3], n.b[4], n.b[5]);
        bool h1 = slab_parity(r, n.b[7], n.b[8], n.b[9], n.b[10], n.b[11]);
        const bool l0 = h0 && c0 < 0, l1 = h1 && c1 < 0;
        if (l0 || l1) {
            count += leaf_parity(r, tris, l0 ? ~c0 : ~c1);
            if (l0 && l1) count += leaf_parity(r, tris, ~c1);
        }
        h0 = h0 && c0 >= 0;
        h1 = h1 && c1 >= 0;
        if (h0 && h1) {
            if (sp >= cap) { ok = false; break; }
            stk.put(sp++, c1);
        }
        if (h0 || h1) node = h0 ? c0 : c1;
        else node = stk.get(--sp);
    }
    return ok;
}

// ------------------------------------------------------------------------------------------
// One voxel, start to finish: raygenMain + closestHitMain + missMain
// (Content/Shaders/DXRVoxelizer.hlsl:58-85, :132-148).  Shared by the kernels and tests/hostcheck.
// ------------------------------------------------------------------------------------------
struct SceneView {
    const Node32* nodes;
    const TriPos* triPos;
    const TriNrm* triNrm;
    float rootLo[3], rootHi[3];
    const Node64* wide;     // wide copy of `nodes` (reference rule, WALK 2)
    const void* dmCells;    // direction-space lists (dxv_dirmap.h, WALK 4): DirCell[6 R R], DirEntry[], R
    const void* dmEntries;
    uint32_t dmR;
    uint32_t dmCoop;        // 1: a lone lane's long list is scanned by its whole wave (trace_reference_dm_from; option coop)
    const uint32_t* plCells;    // row lists of the parity rule (dirmap.hip): (begin, count) per texel of the plR x plR grid over (y, z); NULL: none
    const uint32_t* plEntries;  // triangle slots
    uint32_t plR;
};

template <class Stack, int ABL = 0>
DXV_HD void trace_reference_lists(Ray& r, const SceneView& sc, const Stack& stk, int cap, Hit& best, float& bestDet);      // dxv_dirmap.h

// closestHitMain / missMain for the closest hit a walk (or the lists: WALK 4, hit not yet divided) found
template <int WALK, int ABL = 0>
DXV_HD uint8_t shade_reference(const SceneView& sc, Ray& r, Hit& best, float bestDet, uint32_t* texel)
{
    if (best.k == 0xffffffffu) return 0;                                         // missMain
    if (ABL & 4) return 1;
    if (WALK == 4) {
        // most triangles answer the predicate for every ray that can hit them (normal_class): no normals, no barycentrics
        const uint32_t cls = (uint32_t)best.leaf >> kClassShift;
        best.leaf &= (int32_t)((1u << kClassShift) - 1u);
        // (with the texel image on only the "inside" class needs its normal: an "outside" hit writes nothing, hlsl:83)
        if (cls != 0u && (!texel || cls != kClassIn)) return cls == kClassIn ? 1 : 0;
        finish_hit(best, bestDet);
        finish_ray_reference(r);                                        // the direction again (not kept through the scan)
    }
    const TriNrm tn = sc.triNrm[best.leaf];
    float nx, ny, nz;
    const bool in = predicate(r, tn.n0, tn.n1, tn.n2, best.b1, best.b2, nx, ny, nz);
    if (in && texel) *texel = pack_texel(nx, ny, nz);
    return in ? 1 : 0;
}

// shade_reference<4> for a hit whose barycentrics wait in the LDS column (leaf_reference_deferred_lds): the same decisions, the same divisions
template <class Stack>
DXV_HD uint8_t shade_reference_lds(const SceneView& sc, Ray& r, int32_t bestLeaf, const Stack& stk, int hitAt, uint32_t* texel)
{
    if (bestLeaf == -1) return 0;                                                // missMain
    const uint32_t cls = (uint32_t)bestLeaf >> kClassShift;
    const int32_t leaf = bestLeaf & (int32_t)((1u << kClassShift) - 1u);
    if (cls != 0u && (!texel || cls != kClassIn)) return cls == kClassIn ? 1 : 0;
    const float det = __builtin_bit_cast(float, stk.get(hitAt + 2));
    const float b1 = __builtin_bit_cast(float, stk.get(hitAt)) / det, b2 = __builtin_bit_cast(float, stk.get(hitAt + 1)) / det;     // finish_hit
    finish_ray_reference(r);                                            // the direction again (not kept through the scan)
    const TriNrm tn = sc.triNrm[leaf];
    float nx, ny, nz;
    const bool in = predicate(r, tn.n0, tn.n1, tn.n2, b1, b2, nx, ny, nz);
    if (in && texel) *texel = pack_texel(nx, ny, nz);
    return in ? 1 : 0;
}

// shade_reference<4> for a hit of which only the tagged slot was kept (leaf_reference_min): the hit triangle is tested once more for its
// barycentrics where they are needed
DXV_HD uint8_t shade_reference_again(const SceneView& sc, Ray& r, int32_t bestLeaf)
{
    if (bestLeaf == -1) return 0;                                                // missMain
    const uint32_t cls = (uint32_t)bestLeaf >> kClassShift;
    if (cls != 0u) return cls == kClassIn ? 1 : 0;
    const int32_t leaf = bestLeaf & (int32_t)((1u << kClassShift) - 1u);
    finish_ray_reference(r);                                            // the direction again (not kept through the scan)
    ray_shear_finished(r);
    const TriPos tp = load_tri(sc.triPos, leaf);
    float t, V, W, det = 1.0f;
    (void)tri_test<false, true>(r, tp.v0, tp.v1, tp.v2, t, V, W, &det);         // (it hit before: the same operands)
    const TriNrm tn = sc.triNrm[leaf];
    float nx, ny, nz;
    return predicate(r, tn.n0, tn.n1, tn.n2, V / det, W / det, nx, ny, nz) ? 1 : 0;
}

// returns occupancy; *texel (optional) = the R10G10B10A2_UNORM value of hlsl:84 or 0; *overflow set
// when the traversal stack was too small.
// WALK: 0 = leaves tested as they are met, 1 = postponed-leaf walk, 2 = postponed-leaf walk over the
// wide nodes, 3 = wide nodes on wave-uniform visits only, 4 = direction-space lists.  All return the same voxel.
template <int WALK, class Stack, int ABL = 0>
DXV_HD uint8_t voxel_reference(const SceneView& sc, uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz,
                               const Stack& stk, int cap, uint32_t* texel, bool& overflow)
{
    if (texel) *texel = 0;
    Ray r;
    ray_origin(N, ix, iy, iz, r.ox, r.oy, r.oz);
    if (origin_leaves_root(r.ox, r.oy, r.oz, sc.rootLo, sc.rootHi)) return 0;   // provably missMain
    if (WALK != 4) finish_ray_reference(r);                                     // (the lists finish the ray when they first test a triangle)
    Hit best;
    float bestDet = 1.0f;
    if (WALK == 4) trace_reference_lists<Stack, ABL>(r, sc, stk, cap, best, bestDet);                           // no tree, no stack: cannot overflow
    const bool ok = WALK == 4 ? true
                  : WALK == 3 ? trace_reference_h(r, sc.nodes, sc.wide, sc.triPos, stk, cap, best)
                  : WALK == 2 ? trace_reference_w(r, sc.wide, sc.triPos, stk, cap, best)
                  : WALK == 1 ? trace_reference_q(r, sc.nodes, sc.triPos, stk, cap, best)
                              : trace_reference(r, sc.nodes, sc.triPos, stk, cap, best);
    if (!ok) { overflow = true; return 0; }
    return shade_reference<WALK, ABL>(sc, r, best, bestDet, texel);
}

template <bool QUEUED, class Stack>
DXV_HD uint8_t voxel_parity(const SceneView& sc, uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz,
                            const Stack& stk, int cap, bool& overflow)
{
    const Ray r = make_ray_parity(N, ix, iy, iz);
    // +X ray: nothing to count when the origin is outside the root's y/z extent or beyond its +x face
    if (!slab_parity(r, sc.rootLo[1], sc.rootLo[2], sc.rootHi[0], sc.rootHi[1], sc.rootHi[2])) return 0;
    uint32_t count;
    const bool ok = QUEUED ? trace_parity_q(r, sc.nodes, sc.triPos, stk, cap, count)
                           : trace_parity(r, sc.nodes, sc.triPos, stk, cap, count);
    if (!ok) { overflow = true; return 0; }
    return (uint8_t)(count & 1u);
}

// ------------------------------------------------------------------------------------------
// Parity mode, row form: one walk of the tree for a run of voxels of a few neighbouring grid rows
// (the kernel takes 2 x 2 rows: they share most of their path through the tree).
// `each(tri)` is called for every triangle whose box meets [ylo, yhi] x [zlo, zhi] in y/z and is
// not entirely left of oxMin; node tests use the outward-rounded boxes (supersets), and the
// caller applies the exact per-row test (parity_row_setup) to what it is handed.  The walk depends
// only on the five bounds: on the device it is wave-uniform (scalar node and triangle fetches, one
// LDS stack per wave).  `TriFetch(leaf)` returns a TriPos.
// ------------------------------------------------------------------------------------------
struct NodeHits { bool h0, h1; int32_t c0, c1; };

// the node test of the row walk on decoded planes (host replay; the kernel tests in the half domain)
DXV_HD NodeHits parity_rows_node(const NodePlanes& n, float ylo, float yhi, float zlo, float zhi, float oxMin)
{
    NodeHits r;
    r.h0 = n.b[1] <= yhi && ylo <= n.b[4] && n.b[2] <= zhi && zlo <= n.b[5] && n.b[3] >= oxMin;
    r.h1 = n.b[7] <= yhi && ylo <= n.b[10] && n.b[8] <= zhi && zlo <= n.b[11] && n.b[9] >= oxMin;
    r.c0 = n.c0; r.c1 = n.c1;
    return r;
}

// `visit(node)` returns which children of a node the rows may meet (NodeHits).
template <class Visit, class TriFetch, class StackT, class Each>
DXV_HD void walk_parity_rows(Visit&& visit, TriFetch&& triAt, StackT& stk, Each&& each)
{
    int sp = 0;
    int32_t node = 0;
    for (;;) {
        const NodeHits n = visit(node);
        if (n.h0 && n.c0 < 0) each(triAt(~n.c0));
        if (n.h1 && n.c1 < 0) each(triAt(~n.c1));
        const bool i0 = n.h0 && n.c0 >= 0, i1 = n.h1 && n.c1 >= 0;
        if (i0 && i1) { stk.push(sp, n.c1); node = n.c0; }
        else if (i0) node = n.c0;
        else if (i1) node = n.c1;
        else {
            if (sp == 0) break;
            node = stk.pop(sp);
        }
    }
}

// The same walk over the four-box nodes (Node64): half as many dependent fetches.  `visit(node)`
// returns the up to four children the rows may meet.
struct WideHits { bool h[4]; int32_t c[4]; };

// host replay of the node test on decoded planes (b = Node64::b, axis-major: [axis * 8 + side * 4 + child])
DXV_HD WideHits parity_rows_wide_node(const Node64& n, float ylo, float yhi, float zlo, float zhi, float oxMin)
{
    WideHits r;
    for (int k = 0; k < 4; ++k) {
        const float hix = half_bits_to_float(n.b[0 * 8 + 4 + k]);
        const float loy = half_bits_to_float(n.b[1 * 8 + k]), hiy = half_bits_to_float(n.b[1 * 8 + 4 + k]);
        const float loz = half_bits_to_float(n.b[2 * 8 + k]), hiz = half_bits_to_float(n.b[2 * 8 + 4 + k]);
        r.h[k] = loy <= yhi && ylo <= hiy && loz <= zhi && zlo <= hiz && hix >= oxMin;     // an unused slot is [+inf, -inf]: never met
        r.c[k] = n.c[k];
    }
    return r;
}

template <class Visit, class TriFetch, class StackT, class Each>
DXV_HD void walk_parity_rows_wide(Visit&& visit, TriFetch&& triAt, StackT& stk, Each&& each)
{
    int sp = 0;
    int32_t node = 0;
    for (;;) {
        const WideHits n = visit(node);
        int32_t next = -1;
        for (int k = 0; k < 4; ++k) {
            if (!n.h[k]) continue;
            if (n.c[k] < 0) each(triAt(~n.c[k]));
            else if (next < 0) next = n.c[k];
            else stk.push(sp, n.c[k]);
        }
        if (next < 0) {
            if (sp == 0) break;
            next = stk.pop(sp);
        }
        node = next;
    }
}

} // namespace dxv
