// l1_roof.hip -- the vector-L1 roof of a gather kernel on gfx950, measured: how many 128-byte line accesses per clock and CU does the
// texture-addresser / L1 pipe sustain when every lane of a 16-byte-per-lane load names its own line?  (The lists kernel of the
// voxelizer makes 23 such loads per brick at ~30 lines each: DESIGN.md section 4.2.)
//
//   hipcc --offload-arch=gfx950 -O3 -o l1_roof l1_roof.hip && ./l1_roof          (prints one JSON line per pattern)
//   rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE -- ./l1_roof   (the counters' view)
//
// Launch: 7,168 single-wave workgroups with 5.7 KB of LDS each = 28 per CU = 7 per SIMD, the occupancy of the lists kernel; every
// wave issues `iters` rounds of four independent 16-byte loads (like a scan round) and folds them into one word.
// Patterns (what the 64 lanes of one load name):
//   lines64_l1   64 different lines of a 16 KB table (stays in the L1)
//   lines32_l1   32 different lines, two lanes each (a brick's ~30 texels)
//   lines64_l2   64 different lines of a 2 MB table (L1 misses, L2 hits: where the lists live)
//   lines32_l2   32 different lines of the 2 MB table
//   coalesced    64 consecutive 16-byte slots = 8 lines (the streaming case, for scale)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LINES, bool L2>
__global__ __launch_bounds__(64) void k_gather(const uint4* __restrict__ table, uint32_t* __restrict__ out, int iters)
{
    __shared__ uint32_t pad[1456];                          // 5,824 B: 28 workgroups per CU
    const uint32_t lane = threadIdx.x;
    // lines of 128 B = 8 slots of 16 B; table sizes: 16 KB = 128 lines, 2 MB = 16,384 lines
    constexpr uint32_t kLines = L2 ? 16384u : 128u;
    const uint32_t who = LINES == 64 ? lane : LINES == 32 ? lane >> 1 : 0u;
    uint32_t line = (who * 2654435761u + blockIdx.x * 40503u) >> 7;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t slot;
            if (LINES == 0) slot = (lane + 64u * (4u * (uint32_t)i + k) + blockIdx.x * 64u) & (kLines * 8u - 1u);      // coalesced
            else slot = (((line + 37u * k) & (kLines - 1u)) << 3) | (lane & 7u);
            v[k] = table[slot];
        }
        acc += (v[0].x ^ v[1].y) + (v[2].z ^ v[3].w);
        line = line * 5u + 1u + (L2 ? acc & 1u : 0u);       // (another set of lines every round; L2: data dependent, like a list scan)
    }
    pad[lane] = acc;
    out[blockIdx.x * 64u + lane] = pad[lane ^ 1u];
}

template <int LINES, bool L2>
static void run(const char* name, const uint4* table, uint32_t* out, int iters, int cus, double clockGHz)
{
    const int blocks = cus * 28;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k_gather<LINES, L2><<<blocks, 64>>>(table, out, iters / 8);               // warm
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        k_gather<LINES, L2><<<blocks, 64>>>(table, out, iters);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double loads = (double)blocks * iters * 4.0, lines = loads * (LINES == 0 ? 8.0 : (double)LINES);
    const double ns = best * 1e6;
    printf("{\"pattern\": \"%s\", \"lines_per_load\": %d, \"waves_per_simd\": 7, \"ms\": %.4f, \"loads_per_ns_per_cu\": %.5f, "
           "\"line_accesses_per_ns_per_cu\": %.4f, \"line_accesses_per_clk_per_cu_at_%.2f_GHz\": %.4f, \"bytes_returned_per_clk_per_cu\": %.1f}\n",
           name, LINES == 0 ? 8 : LINES, best, loads / ns / cus, lines / ns / cus, clockGHz, lines / ns / cus / clockGHz, loads * 1024.0 / ns / cus / clockGHz);
    fflush(stdout);
}

// what a workgroup that finds nothing to do costs the dispatcher, and what a dependent launch costs the stream: the two numbers
// behind "a launch that does not know its size cannot simply launch the worst case" and "one kernel less in front of a short launch"
__global__ __launch_bounds__(64) void k_empty(const uint32_t* __restrict__ limit, uint32_t* __restrict__ out)
{
    if (blockIdx.x >= *limit) return;
    out[blockIdx.x] = blockIdx.x;
}
static void dispatch_costs(uint32_t* out)
{
    uint32_t* limit;
    CHECK(hipMalloc(&limit, 4)); CHECK(hipMemset(limit, 0, 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (uint32_t n : {1u << 16, 1u << 18, 1u << 21}) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(a));
            k_empty<<<n, 64>>>(limit, out);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("{\"pattern\": \"empty_workgroups\", \"workgroups\": %u, \"ms\": %.4f, \"workgroups_per_us\": %.1f}\n", n, best, n / (best * 1e3));
    }
    for (int chain : {1, 2, 4, 8}) {
        float best = 1e30f;
        for (int rep = 0; rep < 7; ++rep) {
            CHECK(hipEventRecord(a));
            for (int k = 0; k < chain; ++k) k_empty<<<1, 64>>>(limit, out);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("{\"pattern\": \"dependent_launches\", \"kernels\": %d, \"us\": %.2f}\n", chain, best * 1e3);
    }
    for (size_t bytes : {(size_t)16 << 20, (size_t)128 << 20}) {
        uint8_t* buf; CHECK(hipMalloc(&buf, bytes));
        float best = 1e30f;
        for (int rep = 0; rep < 7; ++rep) {
            CHECK(hipEventRecord(a));
            CHECK(hipMemsetAsync(buf, 0, bytes, 0));
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("{\"pattern\": \"memset\", \"MiB\": %zu, \"us\": %.2f, \"GB_s\": %.0f}\n", bytes >> 20, best * 1e3, bytes / (best * 1e6));
        CHECK(hipFree(buf));
    }
    fflush(stdout);
}

int main(int argc, char** argv)
{
    int dev = 0, cus = 256, mhz = 2400;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CHECK(hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, dev));   // kHz
    const double ghz = argc > 1 ? atof(argv[1]) : mhz / 1e6;
    uint4* table; uint32_t* out;
    CHECK(hipMalloc(&table, 2u << 20)); CHECK(hipMalloc(&out, (size_t)(1u << 21) * 4 + (size_t)cus * 28 * 64 * 4));
    CHECK(hipMemset(table, 1, 2u << 20));
    const int iters = 2048;
    run<64, false>("lines64_l1", table, out, iters, cus, ghz);
    run<32, false>("lines32_l1", table, out, iters, cus, ghz);
    run<64, true>("lines64_l2", table, out, iters, cus, ghz);
    run<32, true>("lines32_l2", table, out, iters, cus, ghz);
    run<0, false>("coalesced", table, out, iters, cus, ghz);
    dispatch_costs(out);
    CHECK(hipDeviceSynchronize());
    return 0;
}
