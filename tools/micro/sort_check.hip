// sort_check.hip -- the builds' radix sort (dxrvoxelizer_amd/csrc/radix_sort.hip, compiled into this program) against
// std::stable_sort on the same field, and its time, for a list of sizes and plans.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../dxrvoxelizer_amd/csrc sort_check.hip -o sort_check
//   ./sort_check            -> one JSON line per (n, field, plan)
#include "../../dxrvoxelizer_amd/csrc/radix_sort.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace dxv;

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char** argv)
{
    const bool timing = argc > 1 && !strncmp(argv[1], "time", 4);
    struct Case { uint32_t n; int loBit, numBits; };
    std::vector<Case> cases = {{1, 32, 30}, {63, 32, 30}, {4097, 32, 30}, {70000, 32, 30}, {1000000, 32, 30}, {6403636, 27, 37}, {5000000, 29, 35}};
    if (timing) cases = {{1000000, 32, 30}, {10000000, 32, 30}, {6403636, 27, 37}, {150000000, 29, 35}};
    std::vector<int> plans = {0, 8, 9, 10, 11, 16 + 10, 32 + 10, 32 + 8, 16 + 8, 48 + 11};
    if (argc > 2) { plans.clear(); for (int a = 2; a < argc; ++a) plans.push_back(atoi(argv[a])); if (!strcmp(argv[1], "time1")) cases.resize(1); else if (!strcmp(argv[1], "time6")) cases = {cases[2]}; }
    int bad = 0;
    for (const Case& c : cases) {
        std::vector<uint64_t> h(c.n);
        const uint64_t fieldMask = (c.numBits >= 64 ? ~0ull : ((1ull << c.numBits) - 1ull)) << c.loBit;
        for (uint32_t i = 0; i < c.n; ++i) {
            // clustered like Morton codes of a surface: a random field from a small set of high parts + the index below
            const uint64_t f = ((rng() % 4096ull) * 0x100000ull + (rng() & 0xfffffull)) * 0x9e3779b1ull;
            h[i] = ((f << c.loBit) & fieldMask) | (uint64_t)i;
        }
        std::vector<uint64_t> want;
        if (!timing) {
            want = h;
            std::stable_sort(want.begin(), want.end(), [&](uint64_t a, uint64_t b) { return (a & fieldMask) < (b & fieldMask); });
        }
        uint64_t *dk, *dt; uint32_t* dh;
        hipMalloc(&dk, 8ull * c.n); hipMalloc(&dt, 8ull * c.n); hipMalloc(&dh, 4ull * radix_sort_hist_words(c.n));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int plan : plans) {
            radix_sort_set_plan(plan);
            float best = 1e30f;
            uint64_t* res = nullptr;
            for (int rep = 0; rep < (timing ? 5 : 1); ++rep) {
                hipMemcpy(dk, h.data(), 8ull * c.n, hipMemcpyHostToDevice);
                hipEventRecord(e0, 0);
                hipError_t e = radix_sort_keys_bits(dk, dt, c.n, dh, c.loBit, c.numBits, &res, 0);
                hipEventRecord(e1, 0);
                hipError_t e2 = hipDeviceSynchronize();
                if (e != hipSuccess || e2 != hipSuccess) { printf("{\"error\": \"%s\"}\n", hipGetErrorString(e != hipSuccess ? e : e2)); return 2; }
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            long long diff = -1;
            if (!timing) {
                std::vector<uint64_t> got(c.n);
                hipMemcpy(got.data(), res, 8ull * c.n, hipMemcpyDeviceToHost);
                diff = 0; long long first = -1;
                for (uint32_t i = 0; i < c.n; ++i) if (got[i] != want[i]) { if (first < 0) first = i; ++diff; }
                if (diff) { ++bad; printf("{\"n\": %u, \"plan\": %d, \"first_diff\": %lld}\n", c.n, plan, first); }
            }
            printf("{\"n\": %u, \"loBit\": %d, \"numBits\": %d, \"plan\": %d, \"passes\": %d, \"ms\": %.4f, \"differences\": %lld}\n", c.n, c.loBit, c.numBits, plan,
                   radix_sort_passes(c.n, c.numBits), best, diff);
            fflush(stdout);
        }
        hipFree(dk); hipFree(dt); hipFree(dh);
    }
    return bad ? 1 : 0;
}
