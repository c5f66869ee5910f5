// radix_sort.hip -- stable LSD radix sort of 64-bit keys on gfx950 (the build's morton << 32 | index keys, the lists' texel | far radius |
// triangle keys).
//
// A field of the key, bits [loBit, loBit + numBits), is sorted in passes of equal digits of 8 - 11 bits, their width by the number
// of keys (below): 30 bits of Morton code of a million triangles in three passes of 10, the lists' 37 bits of 6.4 M keys in five
// of 8, of 150 M keys in four of 10.  Each pass is
//   histogram (per tile, 2^bits bins)  ->  exclusive scan of every digit's row over the tiles  ->  stable scatter.
// Three shapes (measured, tools/micro/sort_check.hip): up to 2 M keys tiles of 2048 keys (4 waves x 8 keys per lane) and the fewest
// passes -- such a sort is bound by its launches and by the chain of dependent steps inside a workgroup, not by bytes; up to 16 M keys
// tiles of 4096 keys and 8-bit digits (a tile then writes 16 keys = 128 B per digit; wider digits cost more in scattered writes than
// the pass they save); beyond, tiles of 16384 keys (16 waves) and 10-bit digits -- again 16 keys per digit and tile, and the
// histogram stays n / 16 words.
// The scatter keeps its keys in registers (a wave owns 64 x ITEMS consecutive keys) and needs three barriers per TILE: each
// wave ranks its own keys against a wave-private count row in the LDS (wave64 ballot match inside an item, the row carries the count
// from item to item), the rows are then turned into global positions (digit start + this tile's offset + the earlier waves'
// counts), and every key is written.  (Until round 5: one item of all waves at a time, three barriers per ITEM -- 48 per tile.)
// HBM traffic per pass: 8 B read (histogram) + 8 B read + 8 B write (scatter) per key.
#include "dxv_device.h"
#include <atomic>

namespace dxv {

constexpr uint32_t kSortSmall = 2u << 20;           // up to here: tiles of 4 waves x 8 keys per lane, digits of up to 11 bits (fewest passes)
constexpr uint32_t kSortMedium = 16u << 20;         // up to here: tiles of 4 waves x 16 keys per lane, digits of 8 bits; beyond: 16 waves, 10 bits
// TW waves per tile, ITEMS keys per lane: a wave owns 64 x ITEMS consecutive keys
template <int TW, int ITEMS> struct SortShape {
    static constexpr int waves = TW, items = ITEMS, threads = 64 * TW, waveKeys = 64 * ITEMS, tile = waveKeys * TW;
    static constexpr int maxBits = TW == 4 ? 11 : 10, maxBins = 1 << maxBits;      // TW x maxBins count words: 32 / 64 KB of LDS
};
using SortSmall = SortShape<4, 8>;
using SortMedium = SortShape<4, 16>;
using SortLarge = SortShape<16, 16>;

// diagnostic override of the plan (dxv_set_option "sortbits", process-wide; results do not depend on it): 0 = automatic,
// 8..11 = digits of at most that many bits; +16 / +32 / +48: the medium / large / small shape where the scratch allows it
// (atomic, and a build snapshots it ONCE -- radix_sort_plan -- for every question it asks about its sort: a set_option from another
// thread or context between lbvh_build's radix_sort_passes and its radix_sort_keys_bits used to change the parity of the passes)
static std::atomic<int> g_sortPlanWord{0};
void radix_sort_set_plan(int v) { g_sortPlanWord.store(v, std::memory_order_relaxed); }
int radix_sort_plan() { return g_sortPlanWord.load(std::memory_order_relaxed); }

struct SortPlan { int shape, bits, passes; };          // shape: 0 small, 1 medium, 2 large
static SortPlan sort_plan(uint32_t n, int numBits, int plan)
{
    const int g_sortPlan = plan < 0 ? radix_sort_plan() : plan;
    SortPlan pl;
    pl.shape = n <= kSortSmall ? 0 : n <= kSortMedium ? 1 : 2;
    if ((g_sortPlan & 48) == 16 && n <= kSortMedium) pl.shape = 1;
    if ((g_sortPlan & 48) == 32) pl.shape = 2;
    if ((g_sortPlan & 48) == 48 && n <= kSortSmall) pl.shape = 0;
    int maxBits = pl.shape == 0 ? 11 : pl.shape == 1 ? 8 : 10;
    const int asked = g_sortPlan & 15;
    if (asked >= 8 && asked <= (pl.shape == 2 ? 10 : 11)) maxBits = asked;
    pl.passes = (numBits + maxBits - 1) / maxBits;
    if (pl.passes < 1) pl.passes = 1;
    pl.bits = (numBits + pl.passes - 1) / pl.passes;
    if (pl.bits < 8) pl.bits = 8;
    return pl;
}

// exclusive prefix of one value per thread over the workgroup (wave scan + one LDS word per wave); two barriers
template <int TW>
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t v, uint32_t* wsum)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= (uint32_t)off) inc += t;
    }
    if (lane == 63u) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < TW; ++w) base += (uint32_t)w < wave ? wsum[w] : 0u;
    __syncthreads();
    return base + inc - v;
}

template <class S>
__global__ __launch_bounds__(S::threads) void k_sort_hist(const uint64_t* __restrict__ keys, uint32_t n, int shift, int bits,
                                                                       uint32_t* __restrict__ hist, uint32_t numTiles)
{
    __shared__ uint32_t bin[S::maxBins];
    const uint32_t bins = 1u << bits, mask = bins - 1u, tile = blockIdx.x;
    for (uint32_t b = threadIdx.x; b < bins; b += S::threads) bin[b] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)tile * S::tile;
#pragma unroll 4
    for (int j = 0; j < S::items; ++j) {
        const uint64_t i = base + (uint64_t)j * S::threads + threadIdx.x;
        if (i < n) atomicAdd(&bin[(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < bins; b += S::threads) hist[(uint64_t)b * numTiles + tile] = bin[b];     // digit-major
}

// Exclusive scan of every row of the digit-major histogram hist[bins][numTiles] over the tiles, one workgroup per digit, in place;
// the row's total -> totals[digit].  The 2^bits totals are scanned by every workgroup of the scatter kernel for itself (less than
// the launch of a one-workgroup kernel in between).
__global__ __launch_bounds__(256) void k_sort_scan_rows(uint32_t* __restrict__ hist, uint32_t numTiles, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t roundEnd;
    uint32_t* row = hist + (uint64_t)blockIdx.x * numTiles;
    const uint32_t tid = threadIdx.x;
    uint32_t carry = 0;
    // four consecutive tiles per thread: 1024 per round
    for (uint32_t base = 0; base < numTiles; base += 1024u) {
        const uint32_t i = base + 4u * tid;
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = i + q < numTiles ? row[i + q] : 0u; sum += v[q]; }
        uint32_t ex = carry + wg_exclusive_scan<4>(sum, wsum);
#pragma unroll
        for (int q = 0; q < 4; ++q) { if (i + q < numTiles) row[i + q] = ex; ex += v[q]; }
        if (tid == 255u) roundEnd = ex;          // the round's total: the last thread's running sum
        __syncthreads();
        carry = roundEnd;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = carry;
}

template <class S>
__global__ __launch_bounds__(S::threads) void k_sort_scatter(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n,
                                                                          int shift, int bits, const uint32_t* __restrict__ offs,
                                                                          const uint32_t* __restrict__ totals, uint32_t numTiles)
{
    constexpr int TW = S::waves, kSortItems = S::items;
    __shared__ uint32_t cnt[TW * S::maxBins];      // row of wave w: cnt + w * bins
    __shared__ uint32_t wsum[TW];
    const uint32_t bins = 1u << bits, mask = bins - 1u, tile = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t* mine = cnt + wave * bins;
    for (uint32_t b = lane; b < bins; b += 64u) mine[b] = 0;      // (a wave clears and counts in its own row: no barrier in between)
    const uint64_t base = (uint64_t)tile * S::tile + (uint64_t)wave * S::waveKeys;
    // (the words of the second step -- this thread's digits' totals and this tile's offsets -- are asked for first: their latency
    // passes behind the ranking)
    constexpr int per = S::maxBins / S::threads;              // consecutive digits per thread (8 / 1 at full width)
    const uint32_t b0 = tid * per;
    uint32_t t[per], off[per], sum = 0;
#pragma unroll
    for (int q = 0; q < per; ++q) {
        t[q] = b0 + q < bins ? totals[b0 + q] : 0u;
        off[q] = b0 + q < bins ? offs[(uint64_t)(b0 + q) * numTiles + tile] : 0u;
        sum += t[q];
    }
    uint64_t key[kSortItems];
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = base + (uint64_t)j * 64u + lane;
        key[j] = i < n ? in[i] : ~0ull;        // (padding sits behind every valid key of the last tile: it never disturbs a valid key's rank)
    }
    // rank of every key among the keys of its digit in this wave's run: count of the earlier items from the row, ballot match
    // inside the item
    uint32_t rk[kSortItems / 2];
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const uint32_t digit = (uint32_t)(key[j] >> shift) & mask;
        uint64_t peers = ~0ull;
#pragma unroll
        for (int b = 0; b < 11; ++b) {
            if (b < bits) {
                const bool bit = (digit >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
        }
        const uint32_t before = mine[digit];
        const uint32_t inItem = __popcll(peers & ((1ull << lane) - 1ull));
        __builtin_amdgcn_wave_barrier();                     // (every lane has read the count before the digit's first lane moves it on)
        if (inItem == 0) mine[digit] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        const uint32_t r = before + inItem;                   // < 1024
        if (j & 1) rk[j >> 1] |= r << 16; else rk[j >> 1] = r;
    }
    __syncthreads();
    // rows -> global positions: where the digit starts in the output (scan of the totals) + what the tiles in front of this one
    // hold of it + what the waves in front of this one hold of it
    {
        uint32_t start = wg_exclusive_scan<TW>(sum, wsum);
#pragma unroll
        for (int q = 0; q < per; ++q) {
            const uint32_t b = b0 + q;
            if (b < bins) {
                uint32_t run = start + off[q];
#pragma unroll
                for (int w = 0; w < TW; ++w) { const uint32_t c = cnt[w * bins + b]; cnt[w * bins + b] = run; run += c; }
            }
            start += t[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = base + (uint64_t)j * 64u + lane;
        const uint32_t digit = (uint32_t)(key[j] >> shift) & mask;
        const uint32_t r = (j & 1) ? rk[j >> 1] >> 16 : rk[j >> 1] & 0xffffu;
        if (i < n) out[mine[digit] + r] = key[j];
    }
}

template <class S>
static void sort_pass(const uint64_t* src, uint64_t* dst, uint32_t n, int shift, int bits, uint32_t* hist, hipStream_t s)
{
    const uint32_t numTiles = (n + S::tile - 1) / S::tile, bins = 1u << bits;
    uint32_t* totals = hist + (size_t)bins * numTiles;          // the row totals behind the histogram
    k_sort_hist<S><<<numTiles, S::threads, 0, s>>>(src, n, shift, bits, hist, numTiles);
    k_sort_scan_rows<<<bins, 256, 0, s>>>(hist, numTiles, totals);
    k_sort_scatter<S><<<numTiles, S::threads, 0, s>>>(src, dst, n, shift, bits, hist, totals, numTiles);
}

// Stable sort by key bits [loBit, loBit + numBits) (the passes may take in a few bits above the field: the callers' keys hold
// zeros or more of the same order there); tmp is a same-size ping-pong buffer, hist holds radix_sort_hist_words(n) words;
// *result = keys or tmp, whichever holds the sorted keys.
hipError_t radix_sort_keys_bits(uint64_t* keys, uint64_t* tmp, uint32_t n, uint32_t* hist, int loBit, int numBits, uint64_t** result,
                                hipStream_t s, int plan)
{
    const SortPlan pl = sort_plan(n, numBits, plan);
    uint64_t* src = keys;
    uint64_t* dst = tmp;
    for (int pass = 0; pass < pl.passes; ++pass) {
        const int shift = loBit + pl.bits * pass;
        if (shift >= 64) break;
        if (pl.shape == 0) sort_pass<SortSmall>(src, dst, n, shift, pl.bits, hist, s);
        else if (pl.shape == 1) sort_pass<SortMedium>(src, dst, n, shift, pl.bits, hist, s);
        else sort_pass<SortLarge>(src, dst, n, shift, pl.bits, hist, s);
        uint64_t* t = src; src = dst; dst = t;
    }
    if (result) *result = src;
    return hipGetLastError();
}

int radix_sort_passes(uint32_t n, int numBits, int plan) { return sort_plan(n, numBits, plan).passes; }

// words of the histogram scratch of a sort of n keys, whatever its plan: bins x tiles + bins of the shape with the most of them
uint32_t radix_sort_hist_words(uint32_t n)
{
    if (n <= kSortSmall) return 2048u * ((n + 2047u) / 2048u) + 2048u;
    if (n <= kSortMedium) return 2048u * ((n + 4095u) / 4096u) + 2048u;
    return 1024u * (uint32_t)(((uint64_t)n + 16383ull) / 16384ull) + 1024u;
}

} // namespace dxv
