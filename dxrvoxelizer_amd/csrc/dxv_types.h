// dxv_types.h -- device data layout of the scene (HBM resident) shared by the build kernels,
// the traversal kernels and the host-side API.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DXV_HD __host__ __device__ __forceinline__
#else
#define DXV_HD inline
#endif

namespace dxv {

struct alignas(16) F4 { float x, y, z, w; };

// Internal LBVH node, 64 B = four 16-B loads.  Aila/Laine-style: a node carries the boxes of
// BOTH children, so one fetch decides both descents.  Child link >= 0: internal node index;
// < 0: ~leaf, leaf = position in Morton order (index into TriPos/TriNrm).
struct alignas(64) Node {
    float lo0x, lo0y, lo0z, hi0x;   // q0
    float hi0y, hi0z, lo1x, lo1y;   // q1
    float lo1z, hi1x, hi1y, hi1z;   // q2
    int32_t c0, c1;                 // q3.x, q3.y
    uint32_t h0, h1;                // q3.z, q3.w: heights of the child subtrees (leaf = 0)
};
static_assert(sizeof(Node) == 64, "node is 64 B");

// Traversal copy of a node, 32 B = two 16-B loads: the twelve box planes as IEEE half floats
// rounded OUTWARD (lo down, hi up), so every stored box contains the exact one.  Supersets keep
// the slab test conservative (monotone under inclusion); the exact per-triangle box is
// re-derived from the triangle's vertices when a leaf is tested, so results are unchanged while
// a node visit moves half the bytes through the vector memory path (the measured limiter).
// Plane order: one 32-bit word per (axis, side) holding child 0 in its low and child 1 in its high
// half, so that picking the plane a ray ENTERS through (lo when the direction component is
// positive, hi when negative) is one select per axis for both children (dxv_trace.h node_step).
struct alignas(32) Node32 {
    uint16_t b[12];                 // x: lo0 lo1 hi0 hi1, y: lo0 lo1 hi0 hi1, z: lo0 lo1 hi0 hi1
    int32_t c0, c1;
};
static_assert(sizeof(Node32) == 32, "compressed node is 32 B");

// Wide traversal node (reference rule): binary node i with each INTERNAL child replaced by that
// child's two children -- up to four boxes per visit, half the dependent fetches per ray (the
// walk waits on those, not on arithmetic).  Same outward-rounded half planes, same axis-major
// order as Node32: b[axis * 8 + side * 4 + child], i.e. twelve words {child 0 | child 1},
// {child 2 | child 3} per (axis, side).  Links: >= 0 binary node index (whose wide node is entry
// i of the same array), < 0 ~leaf.  Unused slots carry an inverted box (lo = +inf, hi = -inf: no
// ray passes the slab test) and kNoChild.  Entries of binary nodes that were absorbed into their
// parent are never reached.
struct alignas(64) Node64 {
    uint16_t b[24];
    int32_t c[4];
};
static_assert(sizeof(Node64) == 64, "wide node is 64 B");
constexpr int32_t kNoChild = (int32_t)0x80000000;

// Leaf payload in Morton order.  Positions are pre-mapped to the reference's normalised space
// p' = (p - c) / w (Content/Voxelizer.cpp:304-306).  v0.w carries the triangle's index in the
// caller's index buffer (PrimitiveIndex(), hlsl:93) as raw bits.
struct alignas(16) TriPos { F4 v0, v1, v2; };
// Vertex normals of the same triangle (hlsl:102-107, :114-116), fetched once per ray at the end.
struct alignas(16) TriNrm { F4 n0, n1, n2; };

// Relocatable scene blob: [SceneHeader | nodes | nodes32 | nodes64 | triPos | triNrm | (list cells | list entries) | (row cells | row entries)],
// sections 256-B aligned.
struct SceneHeader {
    uint32_t magic;       // 'DXVS'
    uint32_t version;
    uint32_t numTris;
    uint32_t numVerts;
    uint32_t numNodes;    // max(T-1, 1)
    uint32_t treeHeight;  // height of the root (levels of internal nodes)
    float bound[4];
    float rootLo[3], rootHi[3];
    uint64_t offNodes, offTriPos, offTriNrm, totalBytes;
    uint64_t offNodes32;
    uint64_t offNodes64;  // wide nodes: present (numNodes entries) only when hasWide
    uint32_t hasWide;
    float triExtent;      // mean extent of a triangle's box along y and z, normalised units (parity row blocks)
    // Optional trailing sections of an EXPORTED blob: the direction-space lists of the reference rule (dxv_dirmap.h), so that
    // the ranks that import the scene need not build them again.  0 / 0 when the blob carries none.
    uint64_t offListCells, offListEntries;
    uint32_t listRes, listCount;
    // ... and the row lists of the parity rule (dirmap.hip): plRes x plRes cells of (begin, count), plCount triangle slots
    uint64_t offPlCells, offPlEntries;
    uint32_t plRes, plCount;
    uint32_t pad[14];
};
static_assert(sizeof(SceneHeader) % 16 == 0, "header alignment");

constexpr uint32_t kSceneMagic = 0x53565844u; // "DXVS"
constexpr uint32_t kSceneVersion = 8;

// canonical constants (hlsl:5, :76-77)
constexpr float kThreshold = 0.12f;
constexpr float kTMax = 10000.0f;
constexpr float kPad = 1.52587890625e-05f; // 2^-16, outward pad of every per-triangle box

} // namespace dxv
