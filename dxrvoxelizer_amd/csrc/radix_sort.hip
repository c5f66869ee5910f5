// radix_sort.hip -- stable LSD radix sort of 64-bit (morton << 32 | index) keys on gfx950.
//
// Only the Morton half needs sorting: the index half starts ascending and every pass is stable,
// so P = 4 passes of 8-bit digits over bits [32, 64) order the full 64-bit key.  Each pass is
//   histogram (per 4096-key tile, 256 bins)  ->  two-level exclusive scan over (digit, tile)  ->
//   stable scatter (wave64 ballot match for the in-wave rank, LDS for the cross-wave rank).
// HBM traffic per pass: 8 B read (histogram) + 8 B read + 8 B write (scatter) per key.
#include "dxv_device.h"

namespace dxv {

constexpr int kSortThreads = 256;
constexpr int kSortItems = 16;
constexpr int kSortTile = kSortThreads * kSortItems; // 4096 keys per workgroup
constexpr int kWaves = kSortThreads / 64;

__global__ __launch_bounds__(kSortThreads) void k_sort_hist(const uint64_t* __restrict__ keys, uint32_t n,
                                                            int shift, uint32_t* __restrict__ hist, uint32_t numTiles)
{
    __shared__ uint32_t bins[256];
    const uint32_t tile = blockIdx.x;
    bins[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)tile * kSortTile;
#pragma unroll 4
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = base + (uint64_t)j * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&bins[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(uint64_t)threadIdx.x * numTiles + tile] = bins[threadIdx.x]; // digit-major
}

// Two-level exclusive scan of the digit-major histogram hist[256][numTiles]:
//   k_sort_scan_rows   one workgroup per digit: in-place exclusive scan of its row (coalesced),
//                      row total -> totals[digit]
//   the 256 totals     scanned by every workgroup of the scatter kernel for itself (eight steps in LDS: less than the
//                      launch of a one-workgroup kernel in between, 5 us + its gap per pass)
// The scatter kernel adds base[d] + hist[d][tile].  (A single-workgroup scan of the whole
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

__global__ __launch_bounds__(kSortThreads) void k_sort_scatter(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                               uint32_t n, int shift, const uint32_t* __restrict__ offs,
                                                               const uint32_t* __restrict__ totals, uint32_t numTiles)
{
    __shared__ uint32_t run[256];            // keys of each digit already placed by earlier items
    __shared__ uint32_t wcnt[kWaves][256];   // per-wave digit counts of the current item
    const uint32_t tile = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // where digit `tid` starts in the output: exclusive scan of the 256 digit totals (run[] as the scan's scratch)
    const uint32_t mine = totals[tid];
    run[tid] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        const uint32_t a = tid >= off ? run[tid - off] : 0u;
        __syncthreads();
        run[tid] += a;
        __syncthreads();
    }
    const uint32_t digitStart = run[tid] - mine;
    __syncthreads();
    run[tid] = digitStart + offs[(uint64_t)tid * numTiles + tile];     // global start of (digit = tid, this tile)
#pragma unroll
    for (int w = 0; w < kWaves; ++w) wcnt[w][tid] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)tile * kSortTile;
    for (int j = 0; j < kSortItems; ++j) {
        const uint64_t i = base + (uint64_t)j * kSortThreads + tid;
        const bool valid = i < n;
        const uint64_t key = valid ? in[i] : ~0ull;
        const uint32_t digit = (uint32_t)(key >> shift) & 255u;
        // lanes of this wave holding the same digit (all 64 lanes take part; padding lanes sit
        // behind every valid key of the tile, so they never disturb a valid key's rank)
        uint64_t peers = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
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
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { tot += wcnt[w][tid]; wcnt[w][tid] = 0; }
        run[tid] += tot;
        __syncthreads();
    }
}

// keys -> sorted keys; tmp is a same-size ping-pong buffer; hist holds 256 * numTiles words.
// After the 4 passes the result is back in `keys`.
// Stable sort by key bits [loBit, loBit + 8 * passes); *result = keys or tmp, whichever holds the sorted keys.
hipError_t radix_sort_keys_bits(uint64_t* keys, uint64_t* tmp, uint32_t n, uint32_t* hist, int loBit, int passes, uint64_t** result,
                                hipStream_t s)
{
    const uint32_t numTiles = (n + kSortTile - 1) / kSortTile;
    uint64_t* src = keys;
    uint64_t* dst = tmp;
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = loBit + 8 * pass;
        uint32_t* totals = hist + 256u * (size_t)numTiles;      // 256 row totals behind the histogram
        k_sort_hist<<<numTiles, kSortThreads, 0, s>>>(src, n, shift, hist, numTiles);
        k_sort_scan_rows<<<256, 256, 0, s>>>(hist, numTiles, totals);
        k_sort_scatter<<<numTiles, kSortThreads, 0, s>>>(src, dst, n, shift, hist, totals, numTiles);
        uint64_t* t = src; src = dst; dst = t;
    }
    if (result) *result = src;
    return hipGetLastError();
}

// the build's sort: the Morton half, bits [32, 64); the result is in `keys`
hipError_t radix_sort_keys(uint64_t* keys, uint64_t* tmp, uint32_t n, uint32_t* hist, hipStream_t s)
{
    return radix_sort_keys_bits(keys, tmp, n, hist, 32, 4, nullptr, s);
}

uint32_t radix_sort_hist_words(uint32_t n) { return 256u * ((n + kSortTile - 1) / kSortTile) + 512u; }

} // namespace dxv
