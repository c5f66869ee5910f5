// lbvh.hip -- on-device LBVH build: replaces the driver BLAS/TLAS build behind
// Voxelizer::buildAccelerationStructures (Content/Voxelizer.cpp:264-326).
//
//   K1  k_tri_keys     per triangle: gather 3 positions through the index buffer, map to the
//                      reference's normalised space, padded box, 30-bit Morton key | index
//   K2  radix sort     radix_sort.hip (three passes of 10-bit digits)
//   K2b k_tri_gather   per Morton slot: write the 48-B position and normal records
//   K3  hierarchy      Karras 2012, one thread per internal node, parent links (k_gather_and_hierarchy: K2b and K3 share a launch)
//   K4  k_refit_*      bottom-up box merge; a node stores the boxes of BOTH children
//   K5  32-B traversal copy of every node (outward-rounded half-float boxes): written by k_refit_ranges from the registers
//       that hold the exact boxes (k_compress_nodes for the sweep / atomic variants of K4)
//   K6  k_widen_from32  64-B wide traversal copy: up to four boxes per node (reference rule), re-arranged from the
//       32-B copies of the node and its children; made when a wide walk first needs it
//
// All kernels are HBM-streaming integer/float work: one thread per element, 16-B accesses where
// the layout allows, no LDS needed outside the sort.
#include "dxv_device.h"
#include "dxv_math.h"

namespace dxv {

constexpr int kThreads = 256;
static inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kThreads - 1) / kThreads); }

struct Bound4 { float c[4]; };

__device__ __forceinline__ void gather_tri(const float* __restrict__ vb, const uint32_t* __restrict__ ib, uint32_t k,
                                           const Bound4& bnd, F4& a, F4& b, F4& c, uint32_t idx[3])
{
    idx[0] = ib[3ull * k]; idx[1] = ib[3ull * k + 1]; idx[2] = ib[3ull * k + 2];
    a = normalise_pos(vb + 6ull * idx[0], bnd.c);
    b = normalise_pos(vb + 6ull * idx[1], bnd.c);
    c = normalise_pos(vb + 6ull * idx[2], bnd.c);
}

// Also samples the y and z extents of the triangle boxes (about 256 workgroups' worth, fixed point,
// so the sum depends neither on the order of the atomics nor on the run) into rootInfo[8..9] with
// the number of triangles sampled in rootInfo[10]: the mean triangle size in voxels decides how
// many grid rows share a walk in parity mode (traverse.hip, k_parity_rows).
__global__ __launch_bounds__(kThreads) void k_tri_keys(const float* __restrict__ vb, const uint32_t* __restrict__ ib,
                                                       uint32_t T, Bound4 bnd, uint64_t* __restrict__ keys,
                                                       uint32_t* __restrict__ rootInfo, uint32_t sampleStride)
{
    __shared__ unsigned long long part[kThreads / 64];
    const uint32_t k = blockIdx.x * kThreads + threadIdx.x;
    unsigned long long ext = 0;
    if (k < T) {
        F4 a, b, c;
        uint32_t idx[3];
        gather_tri(vb, ib, k, bnd, a, b, c, idx);
        float lo[3], hi[3];
        tri_box(a, b, c, lo, hi);
        keys[k] = morton_key(lo, hi, k);
        ext = (unsigned long long)(((hi[1] - lo[1]) + (hi[2] - lo[2])) * 1048576.0f);      // 2^-20 units
    }
    if (blockIdx.x % sampleStride) return;
    for (int off = 32; off; off >>= 1) ext += __shfl_down(ext, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ext;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sum = 0;
        for (int w = 0; w < kThreads / 64; ++w) sum += part[w];
        atomicAdd(reinterpret_cast<unsigned long long*>(rootInfo + 8), sum);
        const uint32_t first = blockIdx.x * kThreads;
        atomicAdd(rootInfo + 10, T - first < (uint32_t)kThreads ? T - first : (uint32_t)kThreads);
    }
}

__device__ __forceinline__ void tri_gather(uint32_t i, const float* __restrict__ vb, const uint32_t* __restrict__ ib,
                                           uint32_t T, const Bound4& bnd, const uint64_t* __restrict__ keys,
                                           TriPos* __restrict__ triPos, TriNrm* __restrict__ triNrm,
                                           uint32_t* __restrict__ rootInfo)
{
    if (i >= T) return;
    const uint32_t k = (uint32_t)(keys[i] & 0xffffffffull);
    F4 a, b, c;
    uint32_t idx[3];
    gather_tri(vb, ib, k, bnd, a, b, c, idx);
    // vertices that came from a device buffer (dxv_update_vertices_device) were never seen by the host: a NaN / Inf position
    // inside an otherwise finite mesh would not show in the root box.  rootInfo[11] counts such triangles (x - x is 0 for
    // every finite x); dxv_build / dxv_refit fail on a non-zero count.
    const float probe = ((a.x - a.x) + (a.y - a.y) + (a.z - a.z)) + ((b.x - b.x) + (b.y - b.y) + (b.z - b.z)) + ((c.x - c.x) + (c.y - c.y) + (c.z - c.z));
    if (!(probe == 0.0f)) atomicAdd(rootInfo + 11, 1u);
    a.w = __builtin_bit_cast(float, k);
    TriPos tp; tp.v0 = a; tp.v1 = b; tp.v2 = c;
    TriNrm tn;
    const float* n0 = vb + 6ull * idx[0] + 3;
    const float* n1 = vb + 6ull * idx[1] + 3;
    const float* n2 = vb + 6ull * idx[2] + 3;
    tn.n0 = F4{n0[0], n0[1], n0[2], 0.0f};
    tn.n1 = F4{n1[0], n1[1], n1[2], 0.0f};
    tn.n2 = F4{n2[0], n2[1], n2[2], 0.0f};
    triNrm[i] = tn;
    tp.v1.w = __builtin_bit_cast(float, normal_class(a, b, c, tn.n0, tn.n1, tn.n2) << kClassShift);   // spare word of the record
    triPos[i] = tp;
}
__global__ __launch_bounds__(kThreads) void k_tri_gather(const float* __restrict__ vb, const uint32_t* __restrict__ ib,
                                                         uint32_t T, Bound4 bnd, const uint64_t* __restrict__ keys,
                                                         TriPos* __restrict__ triPos, TriNrm* __restrict__ triNrm,
                                                         uint32_t* __restrict__ rootInfo)
{
    tri_gather(blockIdx.x * kThreads + threadIdx.x, vb, ib, T, bnd, keys, triPos, triNrm, rootInfo);
}

// parents[0 .. T-2]: internal nodes, parents[T-1 .. 2T-2]: leaves; word = (parent << 1) | side.
// splits != NULL (a build whose boxes come from the pyramid): the node's split position goes there -- 4 bytes from which
// k_refit_ranges, which knows the node's range, makes both links -- and the 64-byte node is not touched here at all: it is written
// once, whole, by k_refit_ranges, instead of 8 bytes now and the rest later with the line read back in between.
__device__ __forceinline__ void hierarchy_node(uint32_t i, const uint64_t* __restrict__ keys, uint32_t T,
                                               Node* __restrict__ nodes, uint32_t* __restrict__ parents,
                                               uint32_t* __restrict__ rangeEnd, uint32_t* __restrict__ splits)
{
    if (i >= T - 1) return;
    int32_t l, r;
    uint32_t other;
    karras_node(keys, (int64_t)T, (int64_t)i, l, r, other);
    if (splits) splits[i] = l >= 0 ? (uint32_t)l : (uint32_t)~l;
    else { nodes[i].c0 = l; nodes[i].c1 = r; }
    rangeEnd[i] = other;              // node i covers the leaves between i and rangeEnd[i] (pyramid refit)
    parents[l >= 0 ? (uint32_t)l : (T - 1) + (uint32_t)~l] = (i << 1);
    parents[r >= 0 ? (uint32_t)r : (T - 1) + (uint32_t)~r] = (i << 1) | 1u;
    if (i == 0) parents[0] = 0xffffffffu;
}
__global__ __launch_bounds__(kThreads) void k_hierarchy(const uint64_t* __restrict__ keys, uint32_t T,
                                                        Node* __restrict__ nodes, uint32_t* __restrict__ parents,
                                                        uint32_t* __restrict__ rangeEnd, uint32_t* __restrict__ splits)
{
    hierarchy_node(blockIdx.x * kThreads + threadIdx.x, keys, T, nodes, parents, rangeEnd, splits);
}
// K2b and K3 in one launch: both read the sorted keys and nothing of each other, each is a chain of dependent gathers (index ->
// vertices; key -> key -> key) that leaves the memory system mostly waiting -- even workgroups gather triangle records, odd ones
// make hierarchy nodes, and the two chains fill each other's gaps (37.8 + 42.1 us as two launches at 1 M triangles).
__global__ __launch_bounds__(kThreads) void k_gather_and_hierarchy(const float* __restrict__ vb, const uint32_t* __restrict__ ib,
                                                                   uint32_t T, Bound4 bnd, const uint64_t* __restrict__ keys,
                                                                   TriPos* __restrict__ triPos, TriNrm* __restrict__ triNrm,
                                                                   uint32_t* __restrict__ rootInfo, Node* __restrict__ nodes,
                                                                   uint32_t* __restrict__ parents, uint32_t* __restrict__ rangeEnd,
                                                                   uint32_t* __restrict__ splits)
{
    const uint32_t i = (blockIdx.x >> 1) * kThreads + threadIdx.x;
    if (blockIdx.x & 1u) hierarchy_node(i, keys, T, nodes, parents, rangeEnd, splits);
    else tri_gather(i, vb, ib, T, bnd, keys, triPos, triNrm, rootInfo);
}

__device__ __forceinline__ void store_child(Node* node, uint32_t side, const float lo[3], const float hi[3], uint32_t h)
{
    float* w = reinterpret_cast<float*>(node) + 6 * side; // words 0..5: child 0 box, 6..11: child 1 box
    w[0] = lo[0]; w[1] = lo[1]; w[2] = lo[2]; w[3] = hi[0]; w[4] = hi[1]; w[5] = hi[2];
    if (side) node->h1 = h; else node->h0 = h;
}

__device__ __forceinline__ void load_child(const Node* node, uint32_t side, float lo[3], float hi[3], uint32_t& h)
{
    const float* w = reinterpret_cast<const float*>(node) + 6 * side;
    lo[0] = w[0]; lo[1] = w[1]; lo[2] = w[2]; hi[0] = w[3]; hi[1] = w[4]; hi[2] = w[5];
    h = side ? node->h1 : node->h0;
}

// K4, one pass: each leaf thread climbs; at every node the first arrival stops, the second
// merges.  Inter-workgroup hand-off follows the agent-scope release/acquire recipe: plain
// stores -> release fence -> drained -> relaxed agent atomic; the consumer fences (acquire)
// before its plain loads.  Results are order independent (min/max).
__global__ __launch_bounds__(kThreads) void k_refit_atomic(const TriPos* __restrict__ triPos, uint32_t T,
                                                           Node* __restrict__ nodes, const uint32_t* __restrict__ parents,
                                                           uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= T) return;
    float lo[3], hi[3];
    {
        const TriPos tp = triPos[i];
        tri_box(tp.v0, tp.v1, tp.v2, lo, hi);
    }
    uint32_t h = 0;
    uint32_t word = parents[(T - 1) + i];
    for (;;) {
        const uint32_t p = word >> 1, side = word & 1u;
        store_child(&nodes[p], side, lo, hi, h);
        __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = __hip_atomic_fetch_add(&flags[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0) return;
        __threadfence();
        float slo[3], shi[3];
        uint32_t sh;
        load_child(&nodes[p], side ^ 1u, slo, shi, sh);
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = min_(lo[a], slo[a]); hi[a] = max_(hi[a], shi[a]); }
        h = (h > sh ? h : sh) + 1;
        if (p == 0) return;
        word = parents[p];
    }
}

// K4, level-synchronous: sweep number `epoch` (1, 2, ...) merges every node whose children were
// finished by an EARLIER sweep and stamps it with its own number; a stamp of this very sweep, racing
// in from another thread, reads as "not yet" (0 or >= epoch), so one array of stamps does for all
// sweeps and kernel boundaries give the visibility of the boxes.
__global__ __launch_bounds__(kThreads) void k_refit_sweep(const TriPos* __restrict__ triPos, uint32_t T,
                                                          Node* __restrict__ nodes, uint32_t* ready, uint32_t epoch)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= T - 1) return;
    if (ready[i]) return;
    const int32_t c[2] = {nodes[i].c0, nodes[i].c1};
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        if (c[side] < 0) continue;
        const uint32_t stamp = ready[c[side]];
        if (stamp == 0 || stamp >= epoch) return;
    }
#pragma unroll
    for (uint32_t side = 0; side < 2; ++side) {
        float lo[3], hi[3];
        uint32_t h = 0;
        if (c[side] < 0) {
            const TriPos tp = triPos[~c[side]];
            tri_box(tp.v0, tp.v1, tp.v2, lo, hi);
        } else {
            float lo1[3], hi1[3];
            uint32_t h0, h1;
            load_child(&nodes[c[side]], 0, lo, hi, h0);
            load_child(&nodes[c[side]], 1, lo1, hi1, h1);
#pragma unroll
            for (int a = 0; a < 3; ++a) { lo[a] = min_(lo[a], lo1[a]); hi[a] = max_(hi[a], hi1[a]); }
            h = (h0 > h1 ? h0 : h1) + 1;
        }
        store_child(&nodes[i], side, lo, hi, h);
    }
    ready[i] = epoch;
}

// T == 1: one node, the second child is a far-away dummy that no ray accepts.
__global__ void k_single_tri(const TriPos* __restrict__ triPos, Node* __restrict__ nodes)
{
    float lo[3], hi[3];
    const TriPos tp = triPos[0];
    tri_box(tp.v0, tp.v1, tp.v2, lo, hi);
    store_child(&nodes[0], 0, lo, hi, 0);
    const float far[3] = {1e30f, 1e30f, 1e30f};
    store_child(&nodes[0], 1, far, far, 0);
    nodes[0].c0 = ~0;
    nodes[0].c1 = ~0;
}

// K5: traversal copy of the nodes, boxes rounded outward to half floats (dxv_types.h Node32)
__global__ __launch_bounds__(kThreads) void k_compress_nodes(const Node* __restrict__ nodes, uint32_t n, Node32* __restrict__ out)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) out[i] = compress_node(nodes[i]);
}

// rootInfo: lo[3], hi[3] (float bits), height of the root, 1
__global__ void k_root_info(const Node* __restrict__ nodes, uint32_t* __restrict__ rootInfo,
                            const uint32_t* __restrict__ rootReady)
{
    float lo[3], hi[3], lo1[3], hi1[3];
    uint32_t h0, h1;
    load_child(&nodes[0], 0, lo, hi, h0);
    load_child(&nodes[0], 1, lo1, hi1, h1);
    for (int a = 0; a < 3; ++a) {
        rootInfo[a] = __builtin_bit_cast(uint32_t, min_(lo[a], lo1[a]));
        rootInfo[3 + a] = __builtin_bit_cast(uint32_t, max_(hi[a], hi1[a]));
    }
    rootInfo[6] = (h0 > h1 ? h0 : h1) + 1;
    rootInfo[7] = rootReady ? (*rootReady != 0u ? 1u : 0u) : 1u;      // sweep refit: the root has been stamped
}

static hipError_t refit_pyramid(const BuildBuffers& b, bool withHeights, hipStream_t s, uint32_t knownHeight = 0);
__global__ void k_widen_from32(const Node32* __restrict__ nodes32, uint32_t n, Node64* __restrict__ out);    // (below, with the pyramid refit)

// K4 + K5 + root info over an existing hierarchy (links and parent words in place).
static hipError_t refit_stage(const BuildBuffers& b, int refitMode, hipStream_t s, uint32_t knownHeight = 0)
{
    const uint32_t T = b.T;
    const uint32_t numNodes = T > 1 ? T - 1 : 1;
    hipError_t e;
    const uint32_t* rootReadyFlag = nullptr;
    if (T == 1) {
        k_single_tri<<<1, 1, 0, s>>>(b.triPos, b.nodes);
    } else {
        if ((e = hipMemsetAsync(b.flags, 0, sizeof(uint32_t) * (size_t)(T - 1), s)) != hipSuccess) return e;
        if (refitMode == 0) {
            k_refit_atomic<<<blocks_for(T), kThreads, 0, s>>>(b.triPos, T, b.nodes, b.parents, b.flags);
        } else {
            uint32_t* ready = b.flags;
            uint32_t rootReady = 0, epoch = 0;
            if (knownHeight) {
                // same hierarchy as before (dxv_refit): a node of height h is ready after h sweeps, so
                // exactly `height` sweeps finish the root; no host round trips, the stream stays async
                for (uint32_t it = 0; it < knownHeight; ++it)
                    k_refit_sweep<<<blocks_for(T - 1), kThreads, 0, s>>>(b.triPos, T, b.nodes, ready, ++epoch);
                rootReady = 1;      // verified on the device: k_root_info copies the root's stamp
            }
            // first build: tree height unknown (<= 62: distinct 62-bit keys); sweep in batches until the root is ready
            for (int batch = 0; batch < 16 && !rootReady; ++batch) {
                for (int it = 0; it < 8; ++it)
                    k_refit_sweep<<<blocks_for(T - 1), kThreads, 0, s>>>(b.triPos, T, b.nodes, ready, ++epoch);
                if ((e = hipMemcpyAsync(&rootReady, ready, sizeof(uint32_t), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
                if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
            }
            if (!rootReady) return hipErrorUnknown;
            rootReadyFlag = ready;
        }
    }
    k_compress_nodes<<<blocks_for(numNodes), kThreads, 0, s>>>(b.nodes, numNodes, b.nodes32);
    if (!b.deferCopies && b.nodes64) k_widen_from32<<<blocks_for(numNodes), kThreads, 0, s>>>(b.nodes32, numNodes, b.nodes64);
    k_root_info<<<1, 1, 0, s>>>(b.nodes, b.rootInfo, rootReadyFlag);
    return hipGetLastError();
}

hipError_t lbvh_build(const BuildBuffers& b, int refitMode, hipStream_t s, hipEvent_t ev[5])
{
    const uint32_t T = b.T;
    Bound4 bnd;
    for (int a = 0; a < 4; ++a) bnd.c[a] = b.bound[a];
    const uint32_t numNodes = T > 1 ? T - 1 : 1;
    hipError_t e;
    // boxes from the min/max pyramid: every node is written once and whole by k_refit_ranges, links included (made from the split
    // positions k_hierarchy leaves in b.flags -- the arrival counters of the other box merges, free here)
    const bool pyramid = b.pyramid && refitMode == 1 && T > 1;
    uint32_t* splits = pyramid ? b.flags : nullptr;
    // the other merges fill the nodes in piecemeal: poison them first (all-ones = NaN boxes: a box that was never merged cannot pass a slab test)
    if (!pyramid && (e = hipMemsetAsync(b.nodes, 0xff, sizeof(Node) * (size_t)numNodes, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(b.rootInfo, 0, 16 * sizeof(uint32_t), s)) != hipSuccess) return e;

    (void)hipEventRecord(ev[0], s);
    const uint32_t keyBlocks = blocks_for(T);
    // the Morton half of the keys, bits [32, 62), is sorted; the index half starts ascending and every pass is stable, so the whole
    // key ends up ordered.  The keys are written where the sort's passes (three of 10 bits) bring them back to b.keys.
    const int sortPlan = radix_sort_plan();                              // (one snapshot of the diagnostic override for both questions below)
    const bool odd = T > 1 && (radix_sort_passes(T, 30, sortPlan) & 1);
    uint64_t* unsorted = odd ? b.keysTmp : b.keys;
    k_tri_keys<<<keyBlocks, kThreads, 0, s>>>(b.vb, b.ib, T, bnd, unsorted, b.rootInfo, keyBlocks > 256u ? keyBlocks / 256u : 1u);
    (void)hipEventRecord(ev[1], s);
    if (T > 1) {
        uint64_t* sorted = nullptr;
        if ((e = radix_sort_keys_bits(unsorted, odd ? b.keys : b.keysTmp, T, b.hist, 32, 30, &sorted, s, sortPlan)) != hipSuccess) return e;
        if (sorted != b.keys) return hipErrorUnknown;
    }
    (void)hipEventRecord(ev[2], s);
    // (one launch for both up to 2 M triangles: -9 us at 1 M; at 10 M the two are bound by bytes, not by latency, and get in each
    // other's way: 1.34 ms together against 0.79 + 0.36 apart)
    if (T > 1 && T <= (2u << 20))
        k_gather_and_hierarchy<<<2u * blocks_for(T), kThreads, 0, s>>>(b.vb, b.ib, T, bnd, b.keys, b.triPos, b.triNrm, b.rootInfo, b.nodes, b.parents, b.flags2, splits);
    else {
        k_tri_gather<<<blocks_for(T), kThreads, 0, s>>>(b.vb, b.ib, T, bnd, b.keys, b.triPos, b.triNrm, b.rootInfo);
        if (T > 1) k_hierarchy<<<blocks_for(T - 1), kThreads, 0, s>>>(b.keys, T, b.nodes, b.parents, b.flags2, splits);
    }
    (void)hipEventRecord(ev[3], s);
    if (pyramid) e = refit_pyramid(b, true, s);
    else e = refit_stage(b, refitMode, s);
    if (e != hipSuccess) return e;
    (void)hipEventRecord(ev[4], s);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Refit of a known hierarchy without level sweeps.  Node i covers a contiguous run of leaves of
// the Morton order (rangeEnd from k_hierarchy), so each child box is a range union: a min/max
// pyramid over the leaf boxes (an implicit segment tree over P = 2^k >= T slots, heap order,
// 24 B per entry, leaves recomputed from triPos) is built in two launches -- ten levels per
// workgroup through LDS, the top by one workgroup -- and one launch over the nodes queries it,
// at most 2 log2(run) entries per child.  Unions of the same leaf boxes by min/max: the same
// bits as the bottom-up merge.  Heights are a property of the hierarchy: a refit keeps them, the
// first build (DEPTH = true) gets them from the same pyramid -- k_depths climbs the parent links
// once per leaf, which gives the depth of leaf j and, because node j is always an ancestor of
// leaf j, the depth of node j on the way; a seventh pyramid channel holds the deepest leaf of each
// run, and height(child) = deepest leaf of its run - depth(child).  No level-by-level pass, no
// host round trip.
// ---------------------------------------------------------------------------------------------
// depthLeaf[j] = internal nodes above leaf j; depthNode[i] = internal nodes above node i (root: 0)
__global__ __launch_bounds__(kThreads) void k_depths(const uint32_t* __restrict__ parents, uint32_t T,
                                                     uint32_t* __restrict__ depthLeaf, uint32_t* __restrict__ depthNode)
{
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= T) return;
    uint32_t p = parents[(T - 1) + j] >> 1, steps = 1, mine = 0;
    for (;;) {
        if (p == j) mine = steps;           // node j covers leaf j: met exactly once on the way up
        if (p == 0u) break;
        p = parents[p] >> 1;
        ++steps;
    }
    depthLeaf[j] = steps;
    if (j < T - 1) depthNode[j] = steps - mine;
}

struct Box6 { float lo[3], hi[3]; };
__device__ __forceinline__ void box_empty(Box6& b) { for (int a = 0; a < 3; ++a) { b.lo[a] = __builtin_inff(); b.hi[a] = -__builtin_inff(); } }
__device__ __forceinline__ void box_union(Box6& b, const Box6& o) { for (int a = 0; a < 3; ++a) { b.lo[a] = min_(b.lo[a], o.lo[a]); b.hi[a] = max_(b.hi[a], o.hi[a]); } }
__device__ __forceinline__ Box6 leaf_box(const TriPos* __restrict__ triPos, uint32_t T, uint32_t leaf)
{
    Box6 b;
    if (leaf < T) { const TriPos tp = triPos[leaf]; tri_box(tp.v0, tp.v1, tp.v2, b.lo, b.hi); }
    else box_empty(b);
    return b;
}

constexpr uint32_t kPyrLeaves = 1024;         // leaves per workgroup of k_pyramid_low = 10 levels
// pyrD (DEPTH only): the deepest leaf under each pyramid entry, same heap order
template <bool DEPTH>
__global__ __launch_bounds__(kThreads) void k_pyramid_low(const TriPos* __restrict__ triPos, uint32_t T, uint32_t P,
                                                          Box6* __restrict__ pyr, const uint32_t* __restrict__ depthLeaf,
                                                          uint32_t* __restrict__ pyrD)
{
    __shared__ Box6 lds[kPyrLeaves / 2];                         // level 1 .. : at most 512 entries live at a time
    __shared__ uint32_t ldsD[DEPTH ? kPyrLeaves / 2 : 1];
    const uint32_t j0 = blockIdx.x * kPyrLeaves;
    // level 1 from the leaves
    for (uint32_t e = threadIdx.x; e < kPyrLeaves / 2; e += kThreads) {
        Box6 b = leaf_box(triPos, T, j0 + 2 * e);
        box_union(b, leaf_box(triPos, T, j0 + 2 * e + 1));
        lds[e] = b;
        pyr[(P >> 1) + (j0 >> 1) + e] = b;
        if (DEPTH) {
            const uint32_t l = j0 + 2 * e, d0 = l < T ? depthLeaf[l] : 0u, d1 = l + 1 < T ? depthLeaf[l + 1] : 0u;
            ldsD[e] = pyrD[(P >> 1) + (j0 >> 1) + e] = d0 > d1 ? d0 : d1;
        }
    }
    __syncthreads();
    uint32_t n = kPyrLeaves / 4;
    for (uint32_t level = 2; level <= 10; ++level, n >>= 1) {
        Box6 b;
        uint32_t d = 0;
        const bool mine = threadIdx.x < n;
        if (mine) {
            b = lds[2 * threadIdx.x]; box_union(b, lds[2 * threadIdx.x + 1]);
            if (DEPTH) { const uint32_t d0 = ldsD[2 * threadIdx.x], d1 = ldsD[2 * threadIdx.x + 1]; d = d0 > d1 ? d0 : d1; }
        }
        __syncthreads();
        if (mine) {
            lds[threadIdx.x] = b; pyr[(P >> level) + (j0 >> level) + threadIdx.x] = b;
            if (DEPTH) ldsD[threadIdx.x] = pyrD[(P >> level) + (j0 >> level) + threadIdx.x] = d;
        }
        __syncthreads();
    }
}

template <bool DEPTH>
__global__ __launch_bounds__(1024) void k_pyramid_high(uint32_t P, Box6* pyr, uint32_t* pyrD)
{
    for (uint32_t level = 11; (P >> level) >= 1u; ++level) {
        const uint32_t n = P >> level;
        for (uint32_t e = threadIdx.x; e < n; e += 1024u) {
            Box6 b = pyr[(P >> (level - 1)) + 2 * e];
            box_union(b, pyr[(P >> (level - 1)) + 2 * e + 1]);
            pyr[n + e] = b;
            if (DEPTH) { const uint32_t d0 = pyrD[(P >> (level - 1)) + 2 * e], d1 = pyrD[(P >> (level - 1)) + 2 * e + 1]; pyrD[n + e] = d0 > d1 ? d0 : d1; }
        }
        __syncthreads();      // one workgroup: what it stored before the barrier is what it loads after
    }
}

// union over the leaves lo..hi; DEPTH: also their greatest depth
template <bool DEPTH>
__device__ __forceinline__ Box6 range_box(const TriPos* __restrict__ triPos, uint32_t T, uint32_t P, const Box6* __restrict__ pyr,
                                          uint32_t lo, uint32_t hi, const uint32_t* __restrict__ depthLeaf,
                                          const uint32_t* __restrict__ pyrD, uint32_t& deepest)
{
    Box6 b;
    box_empty(b);
    deepest = 0;
    auto take = [&](uint32_t e) {
        box_union(b, e >= P ? leaf_box(triPos, T, e - P) : pyr[e]);
        if (DEPTH) { const uint32_t d = e >= P ? (e - P < T ? depthLeaf[e - P] : 0u) : pyrD[e]; deepest = d > deepest ? d : deepest; }
    };
    uint32_t l = lo + P, r = hi + P + 1u;
    while (l < r) {
        if (l & 1u) { take(l); ++l; }
        if (r & 1u) { --r; take(r); }
        l >>= 1; r >>= 1;
    }
    return b;
}

// nodes32 != NULL: the half-float traversal copy of the node as well, from the registers that hold its exact boxes (the copy
// kernel would read the 64-byte node back: 640 MB at 10 M triangles)
template <bool DEPTH>
__global__ __launch_bounds__(kThreads) void k_refit_ranges(const TriPos* __restrict__ triPos, uint32_t T, uint32_t P,
                                                           const Box6* __restrict__ pyr, const uint32_t* __restrict__ rangeEnd,
                                                           Node* __restrict__ nodes, const uint32_t* __restrict__ depthLeaf,
                                                           const uint32_t* __restrict__ depthNode, const uint32_t* __restrict__ pyrD,
                                                           Node32* __restrict__ nodes32, uint32_t* __restrict__ rootInfo, uint32_t knownHeight,
                                                           const uint32_t* __restrict__ splits)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= T - 1) return;
    const uint32_t j = rangeEnd[i], lo = i < j ? i : j, hi = i < j ? j : i;
    // the build (splits): both links from the split position and the range (karras_node's last lines); a refit: the node's own
    int32_t c0, c1;
    uint32_t gamma;
    if (splits) {
        gamma = splits[i];
        c0 = lo == gamma ? ~(int32_t)gamma : (int32_t)gamma;
        c1 = hi == gamma + 1u ? ~(int32_t)(gamma + 1u) : (int32_t)(gamma + 1u);
    } else {
        c0 = nodes[i].c0; c1 = nodes[i].c1;
        gamma = c0 >= 0 ? (uint32_t)c0 : (uint32_t)~c0;
    }
    uint32_t deep0, deep1;
    const Box6 b0 = range_box<DEPTH>(triPos, T, P, pyr, lo, gamma, depthLeaf, pyrD, deep0);
    const Box6 b1 = range_box<DEPTH>(triPos, T, P, pyr, gamma + 1u, hi, depthLeaf, pyrD, deep1);
    float* w = reinterpret_cast<float*>(&nodes[i]);      // words 0..5: child 0 box, 6..11: child 1 box; a refit's links and heights stay
    w[0] = b0.lo[0]; w[1] = b0.lo[1]; w[2] = b0.lo[2]; w[3] = b0.hi[0]; w[4] = b0.hi[1]; w[5] = b0.hi[2];
    w[6] = b1.lo[0]; w[7] = b1.lo[1]; w[8] = b1.lo[2]; w[9] = b1.hi[0]; w[10] = b1.hi[1]; w[11] = b1.hi[2];
    if (splits) { nodes[i].c0 = c0; nodes[i].c1 = c1; }
    uint32_t h0 = 0, h1 = 0;
    if (DEPTH) {                                         // a child sits one level below node i; a leaf child has height 0
        const uint32_t below = depthNode[i] + 1u;
        nodes[i].h0 = h0 = deep0 - below;
        nodes[i].h1 = h1 = deep1 - below;
    }
    if (i == 0u && rootInfo) {                           // rootInfo (k_root_info's words) from the root's own thread: a launch less
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            rootInfo[a] = __builtin_bit_cast(uint32_t, min_(b0.lo[a], b1.lo[a]));
            rootInfo[3 + a] = __builtin_bit_cast(uint32_t, max_(b0.hi[a], b1.hi[a]));
        }
        rootInfo[6] = DEPTH ? (h0 > h1 ? h0 : h1) + 1u : knownHeight;
        rootInfo[7] = 1u;
    }
    if (nodes32) {
        Node n;                                          // (compress_node reads boxes and links only)
        n.lo0x = b0.lo[0]; n.lo0y = b0.lo[1]; n.lo0z = b0.lo[2]; n.hi0x = b0.hi[0]; n.hi0y = b0.hi[1]; n.hi0z = b0.hi[2];
        n.lo1x = b1.lo[0]; n.lo1y = b1.lo[1]; n.lo1z = b1.lo[2]; n.hi1x = b1.hi[0]; n.hi1y = b1.hi[1]; n.hi1z = b1.hi[2];
        n.c0 = c0; n.c1 = c1; n.h0 = n.h1 = 0;
        nodes32[i] = compress_node(n);
    }
}

// K6 from the half-float copy: the four-box node of binary node i is a re-arrangement of half planes that nodes32 already
// holds -- its own for a leaf child, the child's two boxes for an internal child -- so this pass reads 32-byte nodes (its own
// and up to two children's) instead of 64-byte ones and converts nothing.  Same words as widen_node (tests).
__global__ __launch_bounds__(kThreads) void k_widen_from32(const Node32* __restrict__ nodes32, uint32_t n, Node64* __restrict__ out)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const Node32 me = nodes32[i];
    Node64 w;
    int slot = 0;
    auto emit = [&](const Node32& src, int child, int32_t link) {        // Node32.b: [axis * 4 + {lo0, lo1, hi0, hi1}]
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            w.b[a * 8 + slot] = src.b[a * 4 + child];
            w.b[a * 8 + 4 + slot] = src.b[a * 4 + 2 + child];
        }
        w.c[slot++] = link;
    };
    const int32_t link[2] = {me.c0, me.c1};
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        if (link[side] >= 0) {
            const Node32 m = nodes32[link[side]];
            emit(m, 0, m.c0);
            emit(m, 1, m.c1);
        } else emit(me, side, link[side]);
    }
    for (; slot < 4; ++slot) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { w.b[a * 8 + slot] = 0x7c00u; w.b[a * 8 + 4 + slot] = 0xfc00u; }   // [+inf, -inf]
        w.c[slot] = kNoChild;
    }
    out[i] = w;
}

uint32_t pyramid_slots(uint32_t T)            // entries (24 B box + 4 B depth each) of the pyramid scratch for T leaves
{
    uint32_t P = kPyrLeaves;
    while (P < T) P <<= 1;
    return P;
}

// withHeights: first build -- subtree heights from the parent links as well (scratch: keysTmp, free after the sort)
// rootInfo of a refit that stops at the pyramid: the root's box is the pyramid's top entry (the union of all leaf boxes by the
// same min / max: the same bits as the union of the root's two child boxes), the height is the hierarchy's
__global__ void k_root_info_pyramid(const Box6* __restrict__ pyr, uint32_t height, uint32_t* __restrict__ rootInfo)
{
    const Box6 b = pyr[1];
    for (int a = 0; a < 3; ++a) {
        rootInfo[a] = __builtin_bit_cast(uint32_t, b.lo[a]);
        rootInfo[3 + a] = __builtin_bit_cast(uint32_t, b.hi[a]);
    }
    rootInfo[6] = height;
    rootInfo[7] = 1u;
}

static hipError_t refit_pyramid(const BuildBuffers& b, bool withHeights, hipStream_t s, uint32_t knownHeight)
{
    const uint32_t T = b.T, P = pyramid_slots(T);
    const uint32_t numNodes = T > 1 ? T - 1 : 1;
    Box6* pyr = reinterpret_cast<Box6*>(b.pyramid);
    if (withHeights) {
        uint32_t* depthLeaf = reinterpret_cast<uint32_t*>(b.keysTmp);
        uint32_t* depthNode = depthLeaf + T;
        uint32_t* pyrD = reinterpret_cast<uint32_t*>(pyr + P);
        k_depths<<<blocks_for(T), kThreads, 0, s>>>(b.parents, T, depthLeaf, depthNode);
        k_pyramid_low<true><<<P / kPyrLeaves, kThreads, 0, s>>>(b.triPos, T, P, pyr, depthLeaf, pyrD);
        if (P > kPyrLeaves) k_pyramid_high<true><<<1, 1024, 0, s>>>(P, pyr, pyrD);
        k_refit_ranges<true><<<blocks_for(T - 1), kThreads, 0, s>>>(b.triPos, T, P, pyr, b.flags2, b.nodes, depthLeaf, depthNode, pyrD, b.nodes32, b.rootInfo, 0u, b.flags);
    } else {
        k_pyramid_low<false><<<P / kPyrLeaves, kThreads, 0, s>>>(b.triPos, T, P, pyr, nullptr, nullptr);
        if (P > kPyrLeaves) k_pyramid_high<false><<<1, 1024, 0, s>>>(P, pyr, nullptr);
        if (b.deferBoxes && knownHeight) {                       // (lbvh_refit_boxes does the rest when a walk wants the tree)
            k_root_info_pyramid<<<1, 1, 0, s>>>(pyr, knownHeight, b.rootInfo);
            return hipGetLastError();
        }
        // (a refit keeps the hierarchy's height: the nodes' own height words are the build's.  knownHeight 0 -- a caller without
        // it -- reads them back through k_root_info)
        k_refit_ranges<false><<<blocks_for(T - 1), kThreads, 0, s>>>(b.triPos, T, P, pyr, b.flags2, b.nodes, nullptr, nullptr, nullptr, b.nodes32,
                                                                     knownHeight ? b.rootInfo : nullptr, knownHeight, nullptr);
    }
    // (the half-float copy came out of k_refit_ranges' registers; the four-box copy is a re-arrangement of it)
    if (!b.deferCopies && b.nodes64) k_widen_from32<<<blocks_for(numNodes), kThreads, 0, s>>>(b.nodes32, numNodes, b.nodes64);
    if (!withHeights && !knownHeight) k_root_info<<<1, 1, 0, s>>>(b.nodes, b.rootInfo, nullptr);
    return hipGetLastError();
}

// The four-box copy of the hierarchy (what the wide tree walks read) from the half-float one: what a build or refit with
// deferCopies left undone -- a scene whose launches go through the direction-space lists never needs it.
hipError_t lbvh_traversal_copies(const BuildBuffers& b, hipStream_t s)
{
    const uint32_t numNodes = b.T > 1 ? b.T - 1 : 1;
    if (b.nodes64) k_widen_from32<<<blocks_for(numNodes), kThreads, 0, s>>>(b.nodes32, numNodes, b.nodes64);
    return hipGetLastError();
}

// The node boxes a refit with deferBoxes did not write: the pyramid it built (b.pyramid, untouched since) and the triangle
// records are current, so this is the refit's own last launch, late -- the same words as an undeferred refit (tests).
hipError_t lbvh_refit_boxes(const BuildBuffers& b, hipStream_t s)
{
    const uint32_t T = b.T, P = pyramid_slots(T);
    if (T < 2 || !b.pyramid) return hipErrorInvalidValue;
    k_refit_ranges<false><<<blocks_for(T - 1), kThreads, 0, s>>>(b.triPos, T, P, reinterpret_cast<const Box6*>(b.pyramid), b.flags2, b.nodes,
                                                                 nullptr, nullptr, nullptr, b.nodes32, nullptr, 0u, nullptr);
    return lbvh_traversal_copies(b, s);
}

// Dynamic meshes (the reference API's ALLOW_UPDATE / PERFORM_UPDATE, XUSG/RayTracing/XUSGRayTracing.h:13-22,
// unused by the sample): vertices moved, topology and Morton order kept -> re-gather the triangle
// records and refit the boxes.  ev[0..1] bracket the work.
hipError_t lbvh_refit(const BuildBuffers& b, int refitMode, uint32_t treeHeight, hipStream_t s, hipEvent_t ev[2])
{
    Bound4 bnd;
    for (int a = 0; a < 4; ++a) bnd.c[a] = b.bound[a];
    hipError_t e;
    if ((e = hipMemsetAsync(b.rootInfo, 0, 8 * sizeof(uint32_t), s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(b.rootInfo + 11, 0, sizeof(uint32_t), s)) != hipSuccess) return e;      // (words 8..10: the build's triangle extent stays)
    (void)hipEventRecord(ev[0], s);
    k_tri_gather<<<blocks_for(b.T), kThreads, 0, s>>>(b.vb, b.ib, b.T, bnd, b.keys, b.triPos, b.triNrm, b.rootInfo);
    if (b.pyramid && refitMode != 0 && b.T > 1) e = refit_pyramid(b, false, s, treeHeight);
    else e = refit_stage(b, refitMode, s, treeHeight);
    if (e != hipSuccess) return e;
    (void)hipEventRecord(ev[1], s);
    return hipGetLastError();
}

} // namespace dxv
