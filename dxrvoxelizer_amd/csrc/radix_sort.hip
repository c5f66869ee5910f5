// radix_sort.hip -- stable LSD radix sort of 64-bit keys on gfx950.
//
// The LBVH build sorts (morton30 << 32 | index): only the Morton bits need sorting -- the index half starts ascending and
// every pass is stable -- and 30 bits are three passes of 10-bit digits (round 3; rounds 1-2: four passes of 8 bits over
// bits [32, 64)).  The list build (dirmap.hip) sorts the bits above its triangle field: 33 - 37 bits, four passes of 9 or 10
// bits instead of five of 8.  Each pass is
//   histogram (per 4096-key tile, 2^D bins)  ->  two-level exclusive scan over (digit, tile)  ->
//   stable scatter (wave64 ballot match for the in-wave rank, LDS for the cross-wave rank).
// HBM traffic per pass: 8 B read (histogram) + 8 B read + 8 B write (scatter) per key.
#include "dxv_device.h"

namespace dxv {

constexpr int kSortThreads = 256;
constexpr int kSortItems = 16;
constexpr int kSortTile = kSortThreads * kSortItems; // 4096 keys per workgroup
constexpr int kWaves = kSortThreads / 64;
constexpr int kSortMaxBits = 10;                     // digit widths 8, 9, 10 are compiled

template <int D>
__global__ __launch_bounds__(kSortThreads) void k_sort_hist(const uint64_t* __restrict__ keys, uint32_t n,
                                                            int shift, uint32_t* __restrict__ hist, uint32_t numTiles)
{
    constexpr uint32_t BINS = 1u << D;
    __shared__ uint32_t bins[BINS];
    const uint32_t tile = blockIdx.x;
    for (uint32_t b = threadIdx.x; b < BINS; b += kSortThreads) bins[b] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)tile * kSortTile;
#pragma unroll 4
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = base + (uint64_t)j * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&bins[(uint32_t)(keys[i] >> shift) & (BINS - 1u)], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < BINS; b += kSortThreads) hist[(uint64_t)b * numTiles + tile] = bins[b]; // digit-major
}

// Two-level exclusive scan of the digit-major histogram hist[2^D][numTiles]:
//   k_sort_scan_rows   one workgroup per digit: in-place exclusive scan of its row (coalesced),
//                      row total -> totals[digit]
//   k_sort_scan_digits one workgroup: exclusive scan of the 2^D totals -> digitBase[2^D]
// The scatter kernel adds digitBase[d] + hist[d][tile].  (A single-workgroup scan of the whole
// 256 x numTiles array took 92 us per pass at 1 M keys, 80 % of the sort.)
__global__ __launch_bounds__(256) void k_sort_scan_rows(uint32_t* __restrict__ hist, uint32_t numTiles, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t part[256];
    __shared__ uint32_t carry;
    uint32_t* row = hist + (uint64_t)blockIdx.x * numTiles;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < numTiles; base += 256) {
        const uint32_t i = base + tid;
        const uint32_t v = i < numTiles ? row[i] : 0u;
        part[tid] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 256; off <<= 1) {
            const uint32_t a = tid >= off ? part[tid - off] : 0u;
            __syncthreads();
            part[tid] += a;
            __syncthreads();
        }
        if (i < numTiles) row[i] = carry + part[tid] - v;
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = carry;
}

// exclusive scan of the 2^D row totals (one workgroup of 256 threads, 2^D / 256 consecutive totals per thread)
template <int D>
__global__ __launch_bounds__(256) void k_sort_scan_digits(const uint32_t* __restrict__ totals, uint32_t* __restrict__ digitBase)
{
    constexpr uint32_t PER = (1u << D) / 256u;
    __shared__ uint32_t part[256];
    const uint32_t tid = threadIdx.x;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) { v[k] = totals[tid * PER + k]; sum += v[k]; }
    part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        const uint32_t a = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += a;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) { digitBase[tid * PER + k] = run; run += v[k]; }
}

template <int D>
__global__ __launch_bounds__(kSortThreads) void k_sort_scatter(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                               uint32_t n, int shift, const uint32_t* __restrict__ offs,
                                                               const uint32_t* __restrict__ digitBase, uint32_t numTiles)
{
    constexpr uint32_t BINS = 1u << D;
    __shared__ uint32_t run[BINS];            // keys of each digit already placed by earlier items
    __shared__ uint32_t wcnt[kWaves][BINS];   // per-wave digit counts of the current item
    const uint32_t tile = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t b = tid; b < BINS; b += kSortThreads) {
        run[b] = digitBase[b] + offs[(uint64_t)b * numTiles + tile];     // global start of (digit = b, this tile)
#pragma unroll
        for (int w = 0; w < kWaves; ++w) wcnt[w][b] = 0;
    }
    __syncthreads();
    const uint64_t base = (uint64_t)tile * kSortTile;
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = base + (uint64_t)j * kSortThreads + tid;
        const bool valid = i < n;
        const uint64_t key = valid ? in[i] : ~0ull;
        const uint32_t digit = (uint32_t)(key >> shift) & (BINS - 1u);
        // lanes of this wave holding the same digit (all 64 lanes take part; padding lanes sit
        // behind every valid key of the tile, so they never disturb a valid key's rank)
        uint64_t peers = ~0ull;
#pragma unroll
        for (int b = 0; b < D; ++b) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t rankInWave = __popcll(peers & ((1ull << lane) - 1ull));
        if (rankInWave == 0) wcnt[wave][digit] = __popcll(peers);
        __syncthreads();
        uint32_t before = run[digit];
#pragma unroll
        for (int w = 0; w < kWaves; ++w) before += (uint32_t)w < wave ? wcnt[w][digit] : 0u;
        if (valid) out[before + rankInWave] = key;
        __syncthreads();
        // Only the digits this item touched carry counts (the other slots of the rows are zero and stay zero): of the lanes that
        // lead a group of peers, the one in the lowest wave holding the digit adds the waves' counts to the digit's running
        // start; then every leader clears its own wave's slot.
        if (rankInWave == 0) {
            uint32_t tot = 0;
            bool first = true;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                const uint32_t c = wcnt[w][digit];
                tot += c;
                if ((uint32_t)w < wave && c) first = false;
            }
            if (first) run[digit] += tot;
        }
        __syncthreads();
        if (rankInWave == 0) wcnt[wave][digit] = 0;
    }
}

template <int D>
static void sort_pass(const uint64_t* src, uint64_t* dst, uint32_t n, uint32_t numTiles, uint32_t* hist, int shift, hipStream_t s)
{
    uint32_t* totals = hist + ((size_t)numTiles << D);              // 2^D row totals + 2^D digit bases behind the histogram
    uint32_t* digitBase = totals + (1u << D);
    k_sort_hist<D><<<numTiles, kSortThreads, 0, s>>>(src, n, shift, hist, numTiles);
    k_sort_scan_rows<<<1u << D, 256, 0, s>>>(hist, numTiles, totals);
    k_sort_scan_digits<D><<<1, 256, 0, s>>>(totals, digitBase);
    k_sort_scatter<D><<<numTiles, kSortThreads, 0, s>>>(src, dst, n, shift, hist, digitBase, numTiles);
}

// keys -> sorted keys; tmp is a same-size ping-pong buffer; hist holds radix_sort_hist_words(n) words.
// Stable sort by key bits [loBit, loBit + bits): ceil(bits / 10) passes of equal digit width (8, 9 or 10 bits; the last digit
// may reach beyond loBit + bits -- into bits the caller does not mind being sorted by, or past bit 63, where it reads zeros);
// *result = keys or tmp, whichever holds the sorted keys.
hipError_t radix_sort_keys_bits(uint64_t* keys, uint64_t* tmp, uint32_t n, uint32_t* hist, int loBit, int bits, uint64_t** result,
                                hipStream_t s)
{
    const uint32_t numTiles = (n + kSortTile - 1) / kSortTile;
    int passes = (bits + kSortMaxBits - 1) / kSortMaxBits;
    if (passes < 1) passes = 1;
    int D = (bits + passes - 1) / passes;
    if (D < 8) D = 8;
    uint64_t* src = keys;
    uint64_t* dst = tmp;
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = loBit + D * pass;
        if (shift >= 64) break;
        if (D == 8) sort_pass<8>(src, dst, n, numTiles, hist, shift, s);
        else if (D == 9) sort_pass<9>(src, dst, n, numTiles, hist, shift, s);
        else sort_pass<10>(src, dst, n, numTiles, hist, shift, s);
        uint64_t* t = src; src = dst; dst = t;
    }
    if (result) *result = src;
    return hipGetLastError();
}

// the build's sort: the 30 Morton bits at [32, 62), three passes of 10; the result is in *result (keys or tmp)
hipError_t radix_sort_keys(uint64_t* keys, uint64_t* tmp, uint32_t n, uint32_t* hist, uint64_t** result, hipStream_t s)
{
    return radix_sort_keys_bits(keys, tmp, n, hist, 32, 30, result, s);
}

uint32_t radix_sort_hist_words(uint32_t n) { return (((n + kSortTile - 1) / kSortTile) << kSortMaxBits) + (2u << kSortMaxBits); }

} // namespace dxv
