// dirmap.hip -- device build of the direction-space lists (dxv_dirmap.h) from a scene's triangle
// records.  One-off per scene (and per refit), HBM-bound passes:
//   k_dm_records  one thread per triangle (per triangle and face for meshes of few triangles): footprint records of the faces
//                 that see it + number of texels each covers; rectangles of more than 32 texels are counted by a wave
//                 (k_dm_count_waves: one wave each)
//   k_dm_emit     one thread per (triangle, face): (texel | far radius | triangle) keys for its texels -- by size: up to 32 texels
//                 by that thread, up to 16,384 by a wave of k_dm_emit_waves, more (a triangle at the grid's centre: up to a whole
//                 face of the map) by the whole GPU, k_dm_emit_whole
//   radix sort    by texel, then far radius (stable: then triangle)
//   k_dm_cells    first / last entry of every texel
//   (k_dm_cells also writes the 16-byte entries in list order: each record cut to its texel, dm_local_entry)
// The total is capped (the caller then keeps the tree).
#include "dxv_device.h"
#include "dxv_dirmap.h"

namespace dxv {

namespace {
constexpr uint32_t kThreads = 256;
constexpr uint32_t kDmWholeFrom = 16384u;            // footprints of more texels than this are filled by the whole GPU (k_dm_emit_whole)
constexpr uint32_t kDmWholeMax = 256u;               // ... the first this many of them: the kernel walks their list in every workgroup
constexpr uint32_t kDmFewTriangles = 300000u;       // up to here the record and key kernels spread their work over more waves (below)

// slot of this lane's item in a list whose length sits in *count: one add per wave for all its lanes with `mine` set (an add per
// item on one word is ~3 ns each, serialised: 90 k items were 0.27 ms)
__device__ __forceinline__ uint32_t wave_append(uint32_t* count, bool mine)
{
    const unsigned long long m = __ballot(mine);
    if (!m) return 0u;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(count, (uint32_t)__builtin_popcountll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
    return base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
}

// PER_FACE = false: one thread per triangle, its six faces in turn -- the triangle is loaded once, the 64 triangles of a wave are
// neighbours in Morton order and mostly see the same one or two faces (the clip runs with nearly full waves), and a face that does
// not see the triangle costs a dozen comparisons and no record: its count of 0 is all k_dm_emit looks at.
// PER_FACE = true: one thread per (triangle, face) -- for meshes of few, large triangles (the bunny on the 512 map: 70 k triangles
// of 50 - 100 texels each): a wave of 64 triangles there is a chain of ~80 footprints counted one after the other by the whole
// wave while three quarters of the GPU have no wave at all (0.24 ms for 274 waves); six times the waves, a sixth of the chain.
template <bool PER_FACE>
__global__ __launch_bounds__(kThreads) void k_dm_records(const TriPos* __restrict__ triPos, uint32_t T, uint32_t R,
                                                         DirRecord* __restrict__ rec, uint32_t* __restrict__ counts,
                                                         unsigned long long* __restrict__ total, uint32_t* __restrict__ pairs,
                                                         uint32_t* __restrict__ wideList, uint32_t stride)
{
    // (stride > 1: every stride-th triangle only -- dirmap_count's estimate of a scene's entries on a map)
    const uint32_t t = blockIdx.x * kThreads + threadIdx.x, tri = (PER_FACE ? t / 6u : t) * stride;
    unsigned long long n = 0;
    uint32_t seen = 0;                                                  // faces of this triangle that get entries (PER_FACE: this face or none)
    TriPos tp{};
    if (tri < T) tp = triPos[tri];
#pragma unroll 1
    for (uint32_t face = PER_FACE ? t % 6u : 0u, last = PER_FACE ? face + 1u : 6u; face < last; ++face) {
        const uint32_t i = tri * 6u + face;
        DirFootprint f;
        uint32_t c = 0, i0 = 0, i1 = 0, j0 = 0, j1 = 0;
        DirTexelTest tt{};
        bool wide = false;
        if (tri < T && dm_footprint(tp, face, f)) {
            DirRecord e = dm_record(f);
            if (dm_rect(e, R, i0, i1, j0, j1)) {
                c = (i1 - i0 + 1u) * (j1 - j0 + 1u);
                dm_record_on_map(e, c);                                 // (by the rectangle's area, like every other threshold of the record)
                // texels wholly outside an edge of the projected triangle get no entry (dm_texel_outside: k_dm_emit skips the same ones)
                tt = dm_texel_test(e, c);
                wide = tt.on && c > 32u;                                // (counted by the whole wave below)
                if (tt.on && !wide) {
                    c = 0;
                    for (uint32_t j = j0; j <= j1; ++j)
                        for (uint32_t x = i0; x <= i1; ++x) c += dm_texel_outside(tt, R, x, j) ? 0u : 1u;
                }
            }
            rec[i] = e;
        }
        // rectangles of more than 32 texels with a texel test (a coarse mesh, a soup: 27 texels per triangle on average, up to 1,024)
        // are counted by a wave, 64 texels at a time (one thread walking a thousand texels while its neighbours wait was a third of a
        // 10 M-triangle list build) -- a wave of k_dm_count_waves each: this kernel's own waves counting their 64 x 6 rectangles one
        // after the other were the tail of a region of large triangles (dragon x9).  Their count here: 0; their pair is listed for
        // k_dm_emit all the same.
        if (!PER_FACE) {
            const uint32_t slot = wave_append(reinterpret_cast<uint32_t*>(total) + 6, wide);
            if (wide) { wideList[slot] = i; c = 0; }
        } else {
            // (a mesh of few triangles: its waves hold a dozen rectangles each, not 64 x 6, and count them themselves -- the second
            // kernel would cost such a build more than the chain does: bunny 0.75 against 0.66 ms, dragon 1.03 against 0.90)
            unsigned long long m = __ballot(wide);
            const uint32_t lane0 = threadIdx.x & 63u;
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1ull;
                auto bf = [src](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); };
                auto bu = [src](uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); };
                DirTexelTest t;
                t.on = true;
#pragma unroll
                for (int k = 0; k < 3; ++k) { t.nx[k] = bf(tt.nx[k]); t.ny[k] = bf(tt.ny[k]); t.c[k] = bf(tt.c[k]); }
                const uint32_t a0 = bu(i0), a1 = bu(i1), b0 = bu(j0), b1 = bu(j1), wdt = a1 - a0 + 1u, cnt = wdt * (b1 - b0 + 1u);
                const uint32_t stepX = 64u % wdt, stepJ = 64u / wdt;
                uint32_t x = a0 + lane0 % wdt, j = b0 + lane0 / wdt, kept = 0;
                for (uint32_t first = 0; first < cnt; first += 64u) {
                    kept += (uint32_t)__builtin_popcountll(__ballot(first + lane0 < cnt && !dm_texel_outside(t, R, x, j)));
                    x += stepX; j += stepJ;
                    if (x > a1) { x -= wdt; ++j; }
                }
                if ((int)lane0 == src) c = kept;
            }
        }
        if (tri < T) {
            counts[i] = c;
            n += c;
            seen |= (c || wide) ? 1u << face : 0u;
        }
    }
    // The (triangle, face) pairs that get entries, as a compact list for k_dm_emit: five pairs in six have none, and a thread per
    // pair left k_dm_emit's waves with a sixth of their lanes in the loop over texels.  One add per workgroup; the order of the
    // list is whatever the adds make it, the keys land at their pair's offset all the same.
    const uint32_t mine = (uint32_t)__builtin_popcount(seen), lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t incl = mine;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if ((int)lane >= off) incl += o; }
    for (int off = 32; off; off >>= 1) n += __shfl_down(n, off);
    __shared__ unsigned long long part[kThreads / 64];
    __shared__ uint32_t pairPart[kThreads / 64], pairBase;
    if (lane == 0u) part[w] = n;
    if (lane == 63u) pairPart[w] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        uint32_t np = 0;
        for (uint32_t k = 0; k < kThreads / 64; ++k) { s += part[k]; np += pairPart[k]; }
        if (s) atomicAdd(total, s);
        pairBase = np ? atomicAdd(reinterpret_cast<uint32_t*>(total + 1), np) : 0u;
    }
    __syncthreads();
    uint32_t at = pairBase + incl - mine;
    for (uint32_t k = 0; k < w; ++k) at += pairPart[k];
    for (uint32_t face = 0; face < 6u; ++face)
        if (seen & (1u << face)) pairs[at++] = tri * 6u + face;
}

// counts of the rectangles k_dm_records listed: a fixed number of waves, each takes every (number of waves)-th
__global__ __launch_bounds__(kThreads) void k_dm_count_waves(const DirRecord* __restrict__ rec, uint32_t R, const uint32_t* __restrict__ wideList,
                                                             uint32_t* __restrict__ counts, unsigned long long* __restrict__ total)
{
    const uint32_t n = reinterpret_cast<const uint32_t*>(total)[6], lane = threadIdx.x & 63u, waves = gridDim.x * (kThreads / 64u);
    unsigned long long sum = 0;
    for (uint32_t p = (blockIdx.x * kThreads + threadIdx.x) >> 6; p < n; p += waves) {
        const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)wideList[p]);
        const DirRecord rc = rec[k];
        uint32_t a0, a1, b0, b1;
        (void)dm_rect(rc, R, a0, a1, b0, b1);
        const uint32_t w = a1 - a0 + 1u, cnt = w * (b1 - b0 + 1u);
        const DirTexelTest tt = dm_texel_test(rc, cnt);
        const uint32_t stepX = 64u % w, stepJ = 64u / w;
        uint32_t x = a0 + lane % w, j = b0 + lane / w, kept = 0;
        for (uint32_t first = 0; first < cnt; first += 64u) {
            kept += (uint32_t)__builtin_popcountll(__ballot(first + lane < cnt && !dm_texel_outside(tt, R, x, j)));
            x += stepX; j += stepJ;
            if (x > a1) { x -= w; ++j; }
        }
        if (lane == 0u) counts[k] = kept;
        sum += kept;
    }
    // (one add per workgroup: 16 k waves adding to the one 64-bit word were 0.2 ms of a 0.21 ms kernel)
    __shared__ unsigned long long part[kThreads / 64];
    if (lane == 0u) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0u) {
        unsigned long long s4 = 0;
        for (uint32_t q = 0; q < kThreads / 64u; ++q) s4 += part[q];
        if (s4) atomicAdd(total, s4);
    }
}

// exclusive scan of counts[0 .. n) in three launches: per-block sums, scan of the sums by one block, add
constexpr uint32_t kScanBlock = 1024;
__global__ __launch_bounds__(256) void k_scan_sums(const uint32_t* __restrict__ counts, uint32_t n, uint32_t* __restrict__ sums)
{
    const uint32_t base = blockIdx.x * kScanBlock;
    uint32_t s = 0;
    for (uint32_t k = threadIdx.x; k < kScanBlock; k += 256u)
        if (base + k < n) s += counts[base + k];
    for (int off = 32; off; off >>= 1) s += __shfl_down(s, off);
    __shared__ uint32_t part[4];
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(1024) void k_scan_top(uint32_t* sums, uint32_t nb)
{
    // one workgroup, sequential chunks of 1024 with a running carry
    __shared__ uint32_t buf[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024u) {
        const uint32_t idx = base + threadIdx.x;
        const uint32_t v = idx < nb ? sums[idx] : 0u;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024u; off <<= 1) {
            const uint32_t add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0u;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (idx < nb) sums[idx] = carry + buf[threadIdx.x] - v;      // exclusive
        __syncthreads();
        if (threadIdx.x == 1023u) carry += buf[1023];
        __syncthreads();
    }
}

// offsets[i] = sums[block] + exclusive scan inside the block (serial per thread chunk of 4, then across the block)
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* __restrict__ counts, uint32_t n, const uint32_t* __restrict__ sums,
                                                    uint32_t* __restrict__ offsets)
{
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * 4u;
    uint32_t c[4], s = 0;
    for (int k = 0; k < 4; ++k) { c[k] = base + k < n ? counts[base + k] : 0u; s += c[k]; }
    __shared__ uint32_t buf[256];
    buf[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 256u; off <<= 1) {
        const uint32_t add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0u;
        __syncthreads();
        buf[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = sums[blockIdx.x] + buf[threadIdx.x] - s;
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) offsets[base + k] = run;
        run += c[k];
    }
}

// Keys of one (triangle, face) footprint of more than 32 texels, written by a whole wave, 64 texels at a time (one thread walking a
// thousand texels, each with its own radial range, was the build's tail).  rc, the rectangle and `out` are wave-uniform.
__device__ __forceinline__ void emit_footprint(const DirRecord& rc, uint32_t R, const DirKeyLayout& lay, uint32_t k, uint32_t a0, uint32_t a1,
                                               uint32_t b0, uint32_t b1, uint64_t* __restrict__ out, uint32_t lane)
{
    const uint32_t w = a1 - a0 + 1u, n = w * (b1 - b0 + 1u), tri = k / 6u, face = k % 6u;
    const DirTexelTest tt = dm_texel_test(rc, n);
    uint32_t base = 0;                                              // texels with an entry so far (wave-uniform)
    // (the lane's texel moves on by 64 per round: column and row by the remainder and quotient of 64 / w -- one division per
    // footprint instead of two per texel)
    const uint32_t stepX = 64u % w, stepJ = 64u / w;
    uint32_t x = a0 + lane % w, j = b0 + lane / w;
    for (uint32_t first = 0; first < n; first += 64u) {
        const bool keep = first + lane < n && !dm_texel_outside(tt, R, x, j);
        const unsigned long long km = __ballot(keep);
        if (keep) {
            uint32_t r0h, r1h;
            dm_local_radial(rc, R, x, j, r0h, r1h);
            out[base + (uint32_t)__builtin_popcountll(km & ((1ull << lane) - 1ull))] = dm_key(lay, (face * R + j) * R + x, (uint16_t)r1h, tri);
        }
        base += (uint32_t)__builtin_popcountll(km);
        x += stepX; j += stepJ;
        if (x > a1) { x -= w; ++j; }
    }
}

// one thread per (triangle, face) pair that gets entries (k_dm_records' list; workgroups behind its end leave at once): footprints
// of up to 32 texels are written here, by their thread; larger ones go on two lists --
//   wave list  (33 .. wholeFrom texels): k_dm_emit_waves gives each a wave of its own.  Written here by the pair's wave one after
//              the other they were a chain of up to 64 x 3 rounds in the waves of a region of large triangles (the dragon around
//              the centre of dragon x9: 0.7 ms of a 1.9 ms build) while the rest of the GPU had long finished;
//   whole list (more: a triangle close to the grid's centre, up to a whole face of the map): every texel gets an entry with the
//              footprint's own radial range (no texel test, no local range: dm_texel_test, dm_record_on_map), a plain fill that
//              k_dm_emit_whole spreads over the GPU.
// Lists: (pair, first key) x 2 words from the front / pairs from the back of `lists` (the counts, which nobody reads after the scan).
__global__ __launch_bounds__(kThreads) void k_dm_emit(const DirRecord* __restrict__ rec, const uint32_t* __restrict__ pairs,
                                                      const unsigned long long* __restrict__ total,
                                                      const uint32_t* __restrict__ offsets, uint32_t R, uint64_t* __restrict__ keys,
                                                      uint32_t* __restrict__ lists, uint32_t listWords, uint32_t* __restrict__ listCounts, uint32_t wholeFrom)
{
    const uint32_t t = blockIdx.x * kThreads + threadIdx.x, numPairs = *reinterpret_cast<const uint32_t*>(total + 1);
    if (blockIdx.x * kThreads >= numPairs) return;
    const uint32_t i = t < numPairs ? pairs[t] : 0u;
    const DirKeyLayout lay = dm_key_layout(R);
    uint32_t i0 = 0, i1 = 0, j0 = 0, j1 = 0;
    DirRecord rc{};
    if (t < numPairs) rc = rec[i];
    const bool valid = t < numPairs && dm_rect(rc, R, i0, i1, j0, j1);
    const uint32_t area = valid ? (i1 - i0 + 1u) * (j1 - j0 + 1u) : 0u;
    const bool toWhole = area > wholeFrom, toWaves = area > 32u && !toWhole;
    {
        // (every workgroup of k_dm_emit_whole walks the whole list: it stays short -- what does not fit gets a wave like the others)
        const uint32_t back = wave_append(listCounts + 1, toWhole);
        const bool whole = toWhole && back < kDmWholeMax;
        if (whole) lists[listWords - 1u - back] = i;
        const bool wave = toWaves || (toWhole && !whole);
        const uint32_t slot = wave_append(listCounts, wave);
        if (wave) lists[slot] = i;
    }
    if (!valid || area > 32u) return;
    const uint32_t tri = i / 6u, face = i % 6u;
    uint64_t* out = keys + offsets[i];
    const DirTexelTest tt = dm_texel_test(rc, area);
    for (uint32_t j = j0; j <= j1; ++j)
        for (uint32_t x = i0; x <= i1; ++x) {
            if (dm_texel_outside(tt, R, x, j)) continue;            // (not counted either: k_dm_records)
            uint32_t r0h, r1h;                                      // (the entry's own far radius: the record cut to this texel)
            dm_local_radial(rc, R, x, j, r0h, r1h);
            *out++ = dm_key(lay, (face * R + j) * R + x, (uint16_t)r1h, tri);
        }
}

// the wave list: a fixed number of waves, each takes every (number of waves)-th footprint
__global__ __launch_bounds__(kThreads) void k_dm_emit_waves(const DirRecord* __restrict__ rec, const uint32_t* __restrict__ lists,
                                                            const uint32_t* __restrict__ listCounts, const uint32_t* __restrict__ offsets,
                                                            uint32_t R, uint64_t* __restrict__ keys)
{
    const uint32_t n = *listCounts, lane = threadIdx.x & 63u, waves = gridDim.x * (kThreads / 64u);
    const DirKeyLayout lay = dm_key_layout(R);
    for (uint32_t p = (blockIdx.x * kThreads + threadIdx.x) >> 6; p < n; p += waves) {
        const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)lists[p]);
        const DirRecord rc = rec[k];
        uint32_t a0, a1, b0, b1;
        (void)dm_rect(rc, R, a0, a1, b0, b1);
        emit_footprint(rc, R, lay, k, a0, a1, b0, b1, keys + offsets[k], lane);
    }
}

// the keys of the rectangles k_dm_emit put on its list: texel idx of a rectangle -> key idx of its run, so any thread can write any
// of them; every workgroup walks the (short) list and takes its stride of each rectangle
__global__ __launch_bounds__(kThreads) void k_dm_emit_whole(const DirRecord* __restrict__ rec, const uint32_t* __restrict__ lists, uint32_t listWords,
                                                            const uint32_t* __restrict__ listCounts, const uint32_t* __restrict__ offsets,
                                                            uint32_t R, uint64_t* __restrict__ keys)
{
    const uint32_t n = listCounts[1] < kDmWholeMax ? listCounts[1] : kDmWholeMax;
    const DirKeyLayout lay = dm_key_layout(R);
    for (uint32_t h = 0; h < n; ++h) {
        const uint32_t i = lists[listWords - 1u - h], tri = i / 6u, face = i % 6u;
        const DirRecord rc = rec[i];
        uint32_t a0, a1, b0, b1;
        (void)dm_rect(rc, R, a0, a1, b0, b1);
        const uint32_t w = a1 - a0 + 1u, cnt = w * (b1 - b0 + 1u);
        uint64_t* out = keys + offsets[i];
        for (uint32_t idx = blockIdx.x * kThreads + threadIdx.x; idx < cnt; idx += gridDim.x * kThreads) {
            const uint32_t x = a0 + idx % w, j = b0 + idx / w;
            uint32_t r0h, r1h;
            dm_local_radial(rc, R, x, j, r0h, r1h);                     // (the record's own range: its flag for a local one is not set)
            out[idx] = dm_key(lay, (face * R + j) * R + x, (uint16_t)r1h, tri);
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_dm_cells(const uint64_t* __restrict__ keys, uint32_t n, const DirRecord* __restrict__ rec,
                                                       uint32_t R, DirCell* __restrict__ cells, DirEntry* __restrict__ entries)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const DirKeyLayout lay = dm_key_layout(R);
    const uint64_t key = keys[i];
    const uint32_t cell = dm_key_cell(lay, key), tri = dm_key_tri(lay, key);
    uint32_t* words = reinterpret_cast<uint32_t*>(cells + cell);      // [0] begin, [1] count | r1max << 16, [2] thick | q1 << 16, [3] q2 | q3 << 16
    if (i == 0u || dm_key_cell(lay, keys[i - 1u]) != cell) words[0] = i;
    const DirRecord rc = rec[(size_t)tri * 6u + cell / (R * R)];
    // (count and far radius: k_dm_close, once every texel's begin is in place; thickest entry: k_dm_stops / k_dm_stops_long)
    const uint32_t inFace = cell % (R * R);
    entries[i] = dm_local_entry(rc, R, inFace % R, inFace / R, tri);    // the record cut to this texel
}
// stop codes (dm_stop_code): the list is walked from the far end, carrying the earliest start seen so far.  Short lists
// (all of a surface mesh's): one thread per texel; long ones (deep scenes: hundreds of entries): one wave per texel, 64
// entries per step, the running minimum by a prefix scan across the lanes (lane 0 = the entry nearest the far end).
constexpr uint32_t kStopsShort = 32u;
// (eight lanes per texel: lane j of a group takes the j-th entry from the far end of a chunk of eight -- the entries of a group lie
// behind one another in memory, the running minimum is three shuffle steps; one thread walking its texel's list alone, twice, was
// 0.15 ms of the 1.1 ms a 1 M-triangle list build takes)
// texels with a long list go on k_dm_stops_long's work list (at most n / 33 of them): one thread per texel, one atomic per
// workgroup of 1024 texels (one per texel on one address: 0.1 ms at 50 k long texels; one per wave of eight texels: 0.5 ms on a soup)
// ... in the same launch (one thread per texel, after k_dm_close): the far radii the start search of a texel looks at first
// (dm_search_hints)
__global__ __launch_bounds__(1024) void k_dm_hints_and_long_cells(DirCell* __restrict__ cells, uint32_t ncells, const DirEntry* __restrict__ entries,
                                                                  uint32_t* __restrict__ longCells, uint32_t* __restrict__ longCount)
{
    __shared__ uint32_t waveCount[16], base;
    const uint32_t c = blockIdx.x * 1024u + threadIdx.x, lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    DirCell cell{};
    if (c < ncells) cell = cells[c];
    if (cell.count > 8u) {
        const DirSearchHints h = dm_search_hints(cell.count);
        auto r1 = [&](uint32_t k) { return (uint16_t)((entries[cell.begin + k].rr >> 16) & 0x7fffu); };
        cells[c].q2 = r1(h.m2);
        if (h.has1) cells[c].q1 = r1(h.m1);
        if (h.has3) cells[c].q3 = r1(h.m3);
    }
    const bool isLong = cell.count > kStopsShort;
    const unsigned long long m = __ballot(isLong);
    if (lane == 0u) waveCount[w] = (uint32_t)__builtin_popcountll(m);
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t n = 0;
        for (int k = 0; k < 16; ++k) n += waveCount[k];
        base = n ? atomicAdd(longCount, n) : 0u;
    }
    __syncthreads();
    if (!isLong) return;
    uint32_t at = base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
    for (uint32_t k = 0; k < w; ++k) at += waveCount[k];
    longCells[at] = c;
}
__global__ __launch_bounds__(kThreads) void k_dm_stops(DirCell* __restrict__ cells, uint32_t ncells, DirEntry* __restrict__ entries)
{
    const uint32_t t = blockIdx.x * kThreads + threadIdx.x, c = t >> 3, j = t & 7u;
    DirCell cell{};
    if (c < ncells) cell = cells[c];
    const bool isLong = cell.count > kStopsShort;                       // (k_dm_stops_long's: k_dm_long_cells has listed them)
    if (c >= ncells || isLong || cell.count == 0u) return;              // (the eight lanes of a group leave together)
    auto far = [](uint32_t rr) { return half_bits_to_float((rr >> 16) & 0x7fffu); };            // dm_entry_r1 / dm_entry_r0 of the radial word
    auto near = [](uint32_t rr) { return half_bits_to_float(0x7fffu - (rr & 0x7fffu)); };
    // radial extent of the thickest entry of the texel: the unit of the stop codes (halfs convert and subtract exactly)
    uint32_t thick = 0;
    for (uint32_t base = 0; base < cell.count; base += 8u)
        if (base + j < cell.count) {
            const uint32_t rr = entries[cell.begin + base + j].rr;
            const uint32_t th = half_up(far(rr) - near(rr));
            if (th > thick) thick = th;
        }
    for (int off = 1; off < 8; off <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)thick, off, 8); if (o > thick) thick = o; }
    if (j == 0u) cells[c].thick = (uint16_t)thick;
    const float step = dm_stop_step(half_bits_to_float((uint16_t)thick));
    // from the far end: lane j of a chunk holds the earliest start among the chunk's entries 0 .. j (three shuffle steps), then
    // among everything behind the chunk (carry)
    float carry = 3.0e38f;
    for (uint32_t base = 0; base < cell.count; base += 8u) {
        const bool valid = base + j < cell.count;
        const uint32_t k = valid ? cell.begin + cell.count - 1u - base - j : cell.begin;
        const uint32_t rr = entries[k].rr, tri = entries[k].tri;
        float v = valid ? near(rr) : 3.0e38f;
        for (int off = 1; off < 8; off <<= 1) {
            const float o = __shfl_up(v, off, 8);
            if ((int)j >= off && o < v) v = o;
        }
        const float sofar = v < carry ? v : carry;
        if (valid) entries[k].tri = (tri & kDmTriMask) | (dm_stop_code(far(rr), sofar, step) << kDmTriBits);
        const float last = __shfl(v, 7, 8);
        if (last < carry) carry = last;
    }
}
// (the texels with long lists were listed by k_dm_stops: longCount[0] of them in longCells)
__global__ __launch_bounds__(64) void k_dm_stops_long(DirCell* __restrict__ cells, const uint32_t* __restrict__ longCells,
                                                      const uint32_t* __restrict__ longCount, DirEntry* __restrict__ entries)
{
    const uint32_t lane = threadIdx.x, nLong = *longCount;
    for (uint32_t which = blockIdx.x; which < nLong; which += gridDim.x) {
    DirCell cell = cells[longCells[which]];
    uint32_t thick = 0;                                                 // thickest entry first (the unit of the stop codes)
    for (uint32_t base = 0; base < cell.count; base += 64u)
        if (base + lane < cell.count) {
            const DirEntry e = entries[cell.begin + base + lane];
            const uint32_t th = half_up(dm_entry_r1(e) - dm_entry_r0(e));
            if (th > thick) thick = th;
        }
    for (int off = 32; off; off >>= 1) { const uint32_t o = __shfl_xor(thick, off); if (o > thick) thick = o; }
    if (lane == 0u) cells[longCells[which]].thick = (uint16_t)thick;
    cell.thick = (uint16_t)thick;
    const float step = dm_stop_step(half_bits_to_float(cell.thick));
    float carry = 3.0e38f;
    for (uint32_t base = 0; base < cell.count; base += 64u) {
        const bool valid = base + lane < cell.count;
        const uint32_t k = valid ? cell.begin + cell.count - 1u - base - lane : cell.begin;
        const DirEntry e = entries[k];
        float v = valid ? dm_entry_r0(e) : 3.0e38f;
        for (int off = 1; off < 64; off <<= 1) {
            const float o = __shfl_up(v, off);
            if ((int)lane >= off && o < v) v = o;
        }
        const float s = v < carry ? v : carry;
        if (valid) entries[k].tri = (e.tri & kDmTriMask) | (dm_stop_code(dm_entry_r1(e), s, step) << kDmTriBits);
        const float last = __shfl(v, 63);
        if (last < carry) carry = last;
    }
    }
}
// count and far radius of every texel: the thread of a texel's LAST key (the lists are sorted by far radius) reads the begin
// its first key wrote in k_dm_cells; lists too long for the 16-bit count field are reported through `longest`
__global__ __launch_bounds__(kThreads) void k_dm_close(const uint64_t* __restrict__ keys, uint32_t n, uint32_t R, DirCell* __restrict__ cells,
                                                       const DirEntry* __restrict__ entries, uint32_t* __restrict__ longest)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const DirKeyLayout lay = dm_key_layout(R);
    const uint32_t cell = dm_key_cell(lay, keys[i]);
    if (i + 1u != n && dm_key_cell(lay, keys[i + 1u]) == cell) return;
    uint32_t* words = reinterpret_cast<uint32_t*>(cells + cell);
    const uint32_t count = i + 1u - words[0];
    words[1] = (count < 0xffffu ? count : 0xffffu) | (((entries[i].rr >> 16) & 0x7fffu) << 16);
    if (count > 0xffffu) atomicMax(longest, count);                    // (only what does not fit is reported: one address, 300 k texels)
}

// Max-mips of the texels' far radii and of their entry counts (dm_mip_max, dm_box_max_count, dxv_dirmap.h), and the count
// levels' "long list" words.  Two launches:
//   k_dm_mip_tiles  one workgroup per TILE x TILE texels of a face (TILE = min(R, 32)): level 0 of both mips from the cells (read
//                   once), levels 1 .. log2(TILE) through LDS; for the levels that get a "long list" word, the tile's sum of counts
//                   and number of non-empty cells into the scratch behind the mips
//   k_dm_mip_top    two workgroups, one per mip: the levels above, each from the one below (a few hundred words at R = 256); the
//                   counts' workgroup then makes the "long list" words (below)
// (until round 5 five launches -- tiles and top per mip, then the words, one workgroup per level walking up to 98 k cells: 75 us of a
// 0.94 ms build at 1 M triangles)
__device__ __forceinline__ void wg_sum2(unsigned long long& s, uint32_t& c, unsigned long long* sumLds, uint32_t* numLds, uint32_t waves)
{
    for (int off = 32; off; off >>= 1) { s += __shfl_down(s, off); c += __shfl_down(c, off); }
    __syncthreads();                                                    // (the arrays may still be read from the last call)
    if ((threadIdx.x & 63u) == 0u) { sumLds[threadIdx.x >> 6] = s; numLds[threadIdx.x >> 6] = c; }
    __syncthreads();
    s = 0; c = 0;
    for (uint32_t k = 0; k < waves; ++k) { s += sumLds[k]; c += numLds[k]; }
}
__global__ __launch_bounds__(256) void k_dm_mip_tiles(const DirCell* __restrict__ cells, uint32_t R, uint32_t tile, uint16_t* __restrict__ mip)
{
    __shared__ uint16_t lds[2][2][32 * 32];                             // [mip][ping-pong]
    __shared__ unsigned long long sumLds[4];
    __shared__ uint32_t numLds[4];
    uint16_t* counts = mip + dm_mip_words(R);
    uint32_t* partials = reinterpret_cast<uint32_t*>(mip + dm_mip_partials_at(R));
    const uint32_t tilesPerSide = R / tile, face = blockIdx.x / (tilesPerSide * tilesPerSide), in = blockIdx.x % (tilesPerSide * tilesPerSide);
    const uint32_t ti0 = (in % tilesPerSide) * tile, tj0 = (in / tilesPerSide) * tile;
    for (uint32_t k = threadIdx.x; k < tile * tile; k += 256u) {
        const uint32_t i = ti0 + k % tile, j = tj0 + k / tile;
        const DirCell c = cells[(face * R + j) * R + i];
        lds[0][0][k] = mip[(face * R + j) * R + i] = (uint16_t)dm_mip_key(c);
        lds[1][0][k] = counts[(face * R + j) * R + i] = (uint16_t)dm_mip_count_key(c);
    }
    __syncthreads();
    uint32_t side = tile, cur = 0, off = 0, r = R;
    for (uint32_t l = 1; side > 1u; ++l) {
        off += 6u * r * r; r >>= 1;
        const uint32_t half = side >> 1;
        unsigned long long sum = 0;
        uint32_t num = 0;
        for (uint32_t k = threadIdx.x; k < half * half; k += 256u) {
            const uint32_t x = k % half, y = k / half;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const uint16_t* a = lds[which][cur] + (2u * y) * side + 2u * x;
                uint16_t m = a[0];
                if (a[1] > m) m = a[1];
                if (a[side] > m) m = a[side];
                if (a[side + 1u] > m) m = a[side + 1u];
                lds[which][cur ^ 1u][y * half + x] = m;
                (which ? counts : mip)[off + (face * r + (tj0 >> l) + y) * r + (ti0 >> l) + x] = m;
                if (which) { sum += m; num += m ? 1u : 0u; }
            }
        }
        if (l >= kDmHeavyLevelMin) {                                     // (uniform: every thread of the workgroup takes part)
            wg_sum2(sum, num, sumLds, numLds, 4u);
            if (threadIdx.x == 0u) {
                uint32_t* p = partials + 2u * ((size_t)(l - kDmHeavyLevelMin) * gridDim.x + blockIdx.x);
                p[0] = (uint32_t)sum; p[1] = num;                       // (at most 64 cells of at most 65,535 entries)
            }
        }
        __syncthreads();
        side = half; cur ^= 1u;
    }
}
// "A long list", per level of the count mip: one and a half times the mean of the level's non-empty cells (at least 8).  A brick's
// rays look into a patch of texels whose size depends on the grid (4 voxels of 2 / N against texels of 2 / R); the launch picks the
// level whose cells are about that patch (dm_heavy_level) and calls a brick heavy when the longest list it can look into is
// longer than that level's word -- a scene of 13 entries per direction and one of 8 draw the line in different places.
// (levels kDmHeavyLevelMin and up: the two levels below would cost more than the rest of the mip build and no grid of up to 2048^3
// asks for them, dm_heavy_level)
__global__ __launch_bounds__(1024) void k_dm_mip_top(uint32_t R, uint32_t fromLevel, uint16_t* mipBase, uint32_t tiles)
{
    __shared__ unsigned long long sumLds[16];
    __shared__ uint32_t numLds[16];
    const uint32_t levels = dm_mip_levels(R);
    uint16_t* mip = mipBase + (blockIdx.x ? dm_mip_words(R) : 0u);      // workgroup 0: far radii, 1: counts
    for (uint32_t l = fromLevel + 1u; l < levels; ++l) {
        const uint32_t r = R >> l, rp = r << 1;
        const uint16_t* below = mip + dm_mip_offset(R, l - 1u);
        uint16_t* out = mip + dm_mip_offset(R, l);
        for (uint32_t k = threadIdx.x; k < 6u * r * r; k += 1024u) {
            const uint32_t face = k / (r * r), y = (k % (r * r)) / r, x = k % r;
            const uint16_t* a = below + (face * rp + 2u * y) * rp + 2u * x;
            uint16_t m = a[0];
            if (a[1] > m) m = a[1];
            if (a[rp] > m) m = a[rp];
            if (a[rp + 1u] > m) m = a[rp + 1u];
            out[k] = m;
        }
        __threadfence();
        __syncthreads();
    }
    if (blockIdx.x == 0u) return;
    const uint32_t* partials = reinterpret_cast<const uint32_t*>(mipBase + dm_mip_partials_at(R));
    uint16_t* thr = mipBase + 2u * dm_mip_words(R);
    for (uint32_t l = kDmHeavyLevelMin; l < levels; ++l) {
        unsigned long long s = 0;
        uint32_t c = 0;
        if (l <= fromLevel) {                                           // made inside the tiles: their sums
            const uint32_t* p = partials + 2u * (size_t)(l - kDmHeavyLevelMin) * tiles;
            for (uint32_t k = threadIdx.x; k < tiles; k += 1024u) { s += p[2u * k]; c += p[2u * k + 1u]; }
        } else {                                                        // made above: the level's own cells (a few hundred)
            const uint32_t r = R >> l, n = 6u * r * r;
            const uint16_t* lc = mip + dm_mip_offset(R, l);
            for (uint32_t k = threadIdx.x; k < n; k += 1024u) { const uint32_t v = lc[k]; s += v; c += v ? 1u : 0u; }
        }
        wg_sum2(s, c, sumLds, numLds, 16u);
        if (threadIdx.x == 0u) {
            const unsigned long long t = c ? (3ull * s + 2ull * c - 1ull) / (2ull * c) : 8ull;
            thr[l] = (uint16_t)(t < 8ull ? 8ull : t > 65535ull ? 65535ull : t);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row lists of the parity rule.  All its rays are +X lines: a row of voxels (fixed y, z) is one point of the (y, z)
// plane, and the triangles its rays can cross are those whose padded box covers that point (parity_row_setup's first
// test).  A grid of R x R texels over the plane lists per texel the triangles whose box reaches it: a row reads one
// 8-byte cell and then its candidates one after the other, instead of walking the tree to them (k_parity_rows waited on
// that chain of ~17 dependent node fetches per row).  A triangle is in a texel's list at most once, every candidate still
// takes the exact per-row test, and the order inside a list cannot matter to a count: the lists are filled through atomic
// cursors, without a sort.  Texels are a monotone function of the coordinate (dm_texel), the same on both sides.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_pl_total(const TriPos* __restrict__ triPos, uint32_t T, uint32_t R, unsigned long long* __restrict__ total)
{
    // grid-stride: a few thousand waves, one pair of atomics each (one per wave of a 1 M-triangle launch on two addresses
    // was 0.3 ms of contention)
    unsigned long long n = 0, m = 0;                                   // entries; the largest rectangle of one triangle (a thread's loop in the fill)
    for (uint32_t t = blockIdx.x * kThreads + threadIdx.x; t < T; t += gridDim.x * kThreads) {
        uint32_t j0, j1, k0, k1;
        pl_rect(triPos[t], R, j0, j1, k0, k1);
        const unsigned long long r = (unsigned long long)(j1 - j0 + 1u) * (k1 - k0 + 1u);
        n += r;
        if (r > m) m = r;
    }
    for (int off = 32; off; off >>= 1) { n += __shfl_down(n, off); const unsigned long long o = __shfl_down(m, off); if (o > m) m = o; }
    if ((threadIdx.x & 63u) == 0u && n) { atomicAdd(total, n); atomicMax(total + 1, m); }
}
// FILL = false: counts[texel] += 1 per covered texel; FILL = true: entries[begin[texel] + cursor[texel]++] = triangle
template <bool FILL>
__global__ __launch_bounds__(kThreads) void k_pl_scatter(const TriPos* __restrict__ triPos, uint32_t T, uint32_t R, uint32_t* __restrict__ counts,
                                                         const uint32_t* __restrict__ begin, uint32_t* __restrict__ entries)
{
    const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
    if (t >= T) return;
    uint32_t j0, j1, k0, k1;
    pl_rect(triPos[t], R, j0, j1, k0, k1);
    for (uint32_t k = k0; k <= k1; ++k) {
#pragma unroll 4
        for (uint32_t j = j0; j <= j1; ++j) {                           // (independent atomics: several in flight)
            const uint32_t c = k * R + j, slot = atomicAdd(counts + c, 1u);
            if (FILL) entries[begin[c] + slot] = t;
        }
    }
}
__global__ __launch_bounds__(kThreads) void k_pl_cells(const uint32_t* __restrict__ begin, const uint32_t* __restrict__ counts, uint32_t n, uint32_t* __restrict__ cells)
{
    const uint32_t c = blockIdx.x * kThreads + threadIdx.x;
    if (c < n) { cells[2u * c] = begin[c]; cells[2u * c + 1u] = counts[c]; }
}
} // namespace

// scratch bytes of a build that emits `entries` keys for T triangles (records, counts/offsets, block sums, keys x 2, histogram)
size_t dirmap_scratch_bytes(uint32_t T, uint64_t entries)
{
    const size_t n6 = 6 * (size_t)T, nb = (n6 + kScanBlock - 1) / kScanBlock;
    return n6 * sizeof(DirRecord) + 2 * n6 * sizeof(uint32_t) + (nb + 1) * sizeof(uint32_t) + 256 +
           2 * (size_t)entries * sizeof(uint64_t) + sizeof(uint32_t) * (size_t)radix_sort_hist_words((uint32_t)entries) + 1024;
}

// Pass 1: records, per-(triangle, face) counts and the total.  rec: 6T entries, counts: 6T words, total: one 64-bit word.
// pairs: 6T words (the (triangle, face) pairs that get entries; their number lands in the word behind the total)
// wideList: 6T words of scratch (the offsets of pass 2, not made yet)
// stride > 1: every stride-th triangle only (an estimate: total x stride; what it leaves in the arrays is of no use to pass 2)
hipError_t dirmap_count(const TriPos* triPos, uint32_t T, uint32_t R, DirRecord* rec, uint32_t* counts, uint32_t* pairs, uint32_t* wideList,
                        unsigned long long* total, hipStream_t s, uint32_t stride)
{
    hipError_t e = hipMemsetAsync(total, 0, 4 * sizeof(unsigned long long), s);      // (entries, pairs; three list lengths)
    if (e != hipSuccess) return e;
    const uint32_t Ts = (T + stride - 1u) / stride;
    if (T <= kDmFewTriangles) k_dm_records<true><<<(6u * Ts + kThreads - 1) / kThreads, kThreads, 0, s>>>(triPos, T, R, rec, counts, total, pairs, wideList, stride);
    else k_dm_records<false><<<(Ts + kThreads - 1) / kThreads, kThreads, 0, s>>>(triPos, T, R, rec, counts, total, pairs, wideList, stride);
    if (T > kDmFewTriangles) k_dm_count_waves<<<2048, kThreads, 0, s>>>(rec, R, wideList, counts, total);
    return hipGetLastError();
}

// Pass 2: lists.  offsets: 6T words, sums: ceil(6T / 1024) words, keys / keysTmp: n each, hist: radix_sort_hist_words(n),
// cells: 6 R R, entries: n (n = the total of pass 1).
hipError_t dirmap_fill(uint32_t T, uint32_t R, const DirRecord* rec, const uint32_t* counts, const uint32_t* pairs, const unsigned long long* total,
                       uint32_t* offsets, uint32_t* sums, uint64_t* keys, uint64_t* keysTmp, uint32_t* hist, uint32_t n, DirCell* cells, DirEntry* entries, uint32_t* longestOut,
                       hipStream_t s)
{
    const uint32_t n6 = 6u * T, nb = (n6 + kScanBlock - 1) / kScanBlock;
    hipError_t e;
    k_scan_sums<<<nb, 256, 0, s>>>(counts, n6, sums);
    k_scan_top<<<1, 1024, 0, s>>>(sums, nb);
    k_scan_apply<<<nb, 256, 0, s>>>(counts, n6, sums, offsets);
    if ((e = hipMemsetAsync(cells, 0, sizeof(DirCell) * 6 * (size_t)R * R, s)) != hipSuccess) return e;
    if (n == 0) return hipGetLastError();
    // (the two lists of large footprints: in the counts, which nobody reads after the scan above; their lengths: two words behind
    // the totals that dirmap_count has zeroed)
    uint32_t* lists = const_cast<uint32_t*>(counts);
    uint32_t* listCounts = reinterpret_cast<uint32_t*>(const_cast<unsigned long long*>(total)) + 4;
    k_dm_emit<<<(n6 + kThreads - 1) / kThreads, kThreads, 0, s>>>(rec, pairs, total, offsets, R, keys, lists, n6, listCounts, kDmWholeFrom);
    k_dm_emit_waves<<<4096, kThreads, 0, s>>>(rec, lists, listCounts, offsets, R, keys);     // (1,024 or 16,384 workgroups: the same times)
    k_dm_emit_whole<<<1024, kThreads, 0, s>>>(rec, lists, n6, listCounts, offsets, R, keys);
    // sort by (texel, far radius): the bits above the triangle field (keys of one texel and radius are emitted in triangle order
    // and the sort is stable)
    const DirKeyLayout lay = dm_key_layout(R);
    const int numBits = (int)(lay.cellBits + 16u), loBit = 64 - numBits;
    uint64_t* sorted = keys;
    if (n > 1 && (e = radix_sort_keys_bits(keys, keysTmp, n, hist, loBit, numBits, &sorted, s)) != hipSuccess) return e;
    k_dm_cells<<<(n + kThreads - 1) / kThreads, kThreads, 0, s>>>(sorted, n, rec, R, cells, entries);
    // the texel words count entries in 16 bits: the caller reads `*longest` (sums[0] is free by now) when it synchronises
    // and keeps the tree walk for a scene with a longer list
    // (sums[1]: number of texels with a long list; their indices go where the sort's other buffer was)
    if ((e = hipMemsetAsync(sums, 0, 2 * sizeof(uint32_t), s)) != hipSuccess) return e;
    uint32_t* longCells = reinterpret_cast<uint32_t*>(sorted == keys ? keysTmp : keys);
    k_dm_close<<<(n + kThreads - 1) / kThreads, kThreads, 0, s>>>(sorted, n, R, cells, entries, sums);
    const uint32_t ncells = 6u * R * R;
    k_dm_hints_and_long_cells<<<(ncells + 1023u) / 1024u, 1024, 0, s>>>(cells, ncells, entries, longCells, sums + 1);
    k_dm_stops<<<(8u * ncells + kThreads - 1) / kThreads, kThreads, 0, s>>>(cells, ncells, entries);
    k_dm_stops_long<<<4096, 64, 0, s>>>(cells, longCells, sums + 1, entries);
    if ((e = hipMemcpyAsync(longestOut, sums, sizeof(uint32_t), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The far-radius map WITHOUT lists: what the brick test of a tree walk reads (k_voxelize: a brick none of whose rays can reach a
// triangle is zeroed and left).  A radial ray from p hits a triangle at p + t d = (|p| + t) d: a point of the triangle in the ray's
// own direction, farther out than its start -- so a ray that starts beyond the farthest triangle point of its texel of direction
// space is a miss, which is the very test the lists' max-mip answers (dm_box_may_be_live) with "farthest entry" for "farthest
// point".  Per texel: the maximum, over the (triangle, face) footprints whose bounding rectangle reaches the texel, of the
// footprint's far radius as a half rounded up -- dm_footprint / dm_record / dm_rect, the list build's own conservative functions, on
// a coarse map (a superset of every texel's lists' far radius: no outline test, no per-texel radial cut).  One thread per triangle.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_dm_far(const TriPos* __restrict__ triPos, uint32_t T, uint32_t R, uint32_t* __restrict__ far32)
{
    const uint32_t tri = blockIdx.x * kThreads + threadIdx.x;
    if (tri >= T) return;
    const TriPos tp = triPos[tri];
#pragma unroll 1
    for (uint32_t face = 0; face < 6u; ++face) {
        DirFootprint f;
        if (!dm_footprint(tp, face, f)) continue;
        const DirRecord e = dm_record(f);
        uint32_t i0, i1, j0, j1;
        if (!dm_rect(e, R, i0, i1, j0, j1)) continue;
        const uint32_t r1 = e.rr >> 16;
        for (uint32_t j = j0; j <= j1; ++j)
            for (uint32_t i = i0; i <= i1; ++i) {
                uint32_t* w = far32 + (face * R + j) * R + i;
                if (*w < r1) atomicMax(w, r1);                           // (most texels have their maximum after a few triangles)
            }
    }
}
// ... as cells the mip build reads: one "entry" where any footprint reaches, its far radius
__global__ __launch_bounds__(256) void k_dm_far_cells(const uint32_t* __restrict__ far32, uint32_t n, DirCell* __restrict__ cells)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    DirCell c{};
    c.count = far32[k] ? 1u : 0u;
    c.r1max = (uint16_t)far32[k];
    cells[k] = c;
}
hipError_t dirmap_far(const TriPos* triPos, uint32_t T, uint32_t R, uint32_t* far32, DirCell* cells, uint16_t* mip, hipStream_t s)
{
    const uint32_t n = 6u * R * R;
    hipError_t e = hipMemsetAsync(far32, 0, sizeof(uint32_t) * n, s);
    if (e != hipSuccess) return e;
    k_dm_far<<<(T + kThreads - 1u) / kThreads, kThreads, 0, s>>>(triPos, T, R, far32);
    k_dm_far_cells<<<(n + 255u) / 256u, 256, 0, s>>>(far32, n, cells);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return dirmap_mip(cells, R, mip, s);
}

// mip: dm_mip_buffer_words(R) 16-bit words (far radii, then entry counts, then the count levels' "long list" words, then scratch)
hipError_t dirmap_mip(const DirCell* cells, uint32_t R, uint16_t* mip, hipStream_t s)
{
    const uint32_t tile = R < 32u ? R : 32u, tiles = 6u * (R / tile) * (R / tile);
    uint32_t tileLevel = 0;
    while ((1u << tileLevel) < tile) ++tileLevel;
    k_dm_mip_tiles<<<tiles, 256, 0, s>>>(cells, R, tile, mip);
    k_dm_mip_top<<<2, 1024, 0, s>>>(R, tileLevel, mip, tiles);
    return hipGetLastError();
}

__global__ __launch_bounds__(kThreads) void k_dm_validate(const DirCell* __restrict__ cells, uint32_t ncells, const DirEntry* __restrict__ entries,
                                                          uint32_t n, uint32_t T, uint32_t* __restrict__ out)
{
    uint32_t badCells = 0, badTris = 0;
    for (uint32_t c = blockIdx.x * kThreads + threadIdx.x; c < ncells; c += gridDim.x * kThreads) {
        const DirCell cell = cells[c];
        if (cell.count && ((uint64_t)cell.begin + cell.count > (uint64_t)n)) ++badCells;
    }
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
        if (dm_entry_tri(entries[i]) >= T) ++badTris;
    if (badCells) atomicAdd(out, badCells);
    if (badTris) atomicAdd(out + 1, badTris);
}
hipError_t dirmap_validate(const DirCell* cells, uint32_t R, const DirEntry* entries, uint32_t n, uint32_t T, uint32_t* out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    k_dm_validate<<<1024, kThreads, 0, s>>>(cells, 6u * R * R, entries, n, T, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(kThreads) void k_pl_validate(const uint32_t* __restrict__ cells, uint32_t ncells, const uint32_t* __restrict__ entries,
                                                          uint32_t n, uint32_t T, uint32_t* __restrict__ out)
{
    uint32_t badCells = 0, badTris = 0;
    for (uint32_t c = blockIdx.x * kThreads + threadIdx.x; c < ncells; c += gridDim.x * kThreads)
        if (cells[2u * c + 1u] && ((uint64_t)cells[2u * c] + cells[2u * c + 1u] > (uint64_t)n)) ++badCells;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
        if (entries[i] >= T) ++badTris;
    if (badCells) atomicAdd(out, badCells);
    if (badTris) atomicAdd(out + 1, badTris);
}
hipError_t parity_lists_validate(const uint32_t* cells, uint32_t R, const uint32_t* entries, uint32_t n, uint32_t T, uint32_t* out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    k_pl_validate<<<1024, kThreads, 0, s>>>(cells, R * R, entries, n, T, out);
    return hipGetLastError();
}

// Row lists of the parity rule (above).  parity_lists_total: total[0] = entries the lists would have, total[1] = texels of the
// largest single rectangle (one thread of the fill walks it); parity_lists_fill: cells = 2 words
// (begin, count) per texel of the R x R grid, entries = `total` triangle slots (+ a few spare words behind them).
// counts / offsets: R R words each, sums: ceil(R R / 1024) + 1 words of scratch.
hipError_t parity_lists_total(const TriPos* triPos, uint32_t T, uint32_t R, unsigned long long* total, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(total, 0, 2 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    const uint32_t blocks = (T + kThreads - 1) / kThreads;
    k_pl_total<<<blocks < 512u ? blocks : 512u, kThreads, 0, s>>>(triPos, T, R, total);
    return hipGetLastError();
}
hipError_t parity_lists_fill(const TriPos* triPos, uint32_t T, uint32_t R, uint32_t* counts, uint32_t* offsets, uint32_t* sums, uint32_t* cells,
                             uint32_t* entries, hipStream_t s)
{
    const uint32_t n = R * R, nb = (n + kScanBlock - 1) / kScanBlock, blocks = (T + kThreads - 1) / kThreads;
    hipError_t e;
    if ((e = hipMemsetAsync(counts, 0, sizeof(uint32_t) * n, s)) != hipSuccess) return e;
    k_pl_scatter<false><<<blocks, kThreads, 0, s>>>(triPos, T, R, counts, nullptr, nullptr);
    k_scan_sums<<<nb, 256, 0, s>>>(counts, n, sums);
    k_scan_top<<<1, 1024, 0, s>>>(sums, nb);
    k_scan_apply<<<nb, 256, 0, s>>>(counts, n, sums, offsets);
    if ((e = hipMemsetAsync(counts, 0, sizeof(uint32_t) * n, s)) != hipSuccess) return e;
    k_pl_scatter<true><<<blocks, kThreads, 0, s>>>(triPos, T, R, counts, offsets, entries);
    k_pl_cells<<<(n + kThreads - 1) / kThreads, kThreads, 0, s>>>(offsets, counts, n, cells);
    return hipGetLastError();
}

} // namespace dxv
