// Microbenchmark: how many vector-L1 line accesses does ONE 16-byte-per-lane load instruction cost, by address pattern?
//   hipcc --offload-arch=gfx950 -O3 -o ta_merge ta_merge.hip ; rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD ...
// Patterns (lane -> 16-B slot index, all inside a 64 KB table so that everything hits the L1 after the first touch):
//   0 coalesced (lane)            1 all lanes one address            2 quads share an address (lane / 4, scattered lines)
//   3 16 scattered addresses, lanes of a quad the same   4 16 scattered addresses, lane % 16 (members far apart)
//   5 64 scattered lines          6 pairs: lane / 2       7 same 64-B line, 4 different 16-B slots per quad, 16 lines (= coalesced)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int P>
__global__ __launch_bounds__(64) void k_pattern(const uint4* __restrict__ table, uint4* __restrict__ out, int iters)
{
    const uint32_t lane = threadIdx.x;
    uint32_t slot;
    const uint32_t scatter = (lane * 2654435761u) >> 20;          // 12 bits
    if (P == 0) slot = lane;
    else if (P == 1) slot = 5;
    else if (P == 2) slot = (((lane / 4) * 2654435761u) >> 20);
    else if (P == 3) slot = (((lane / 4) * 2654435761u) >> 20);
    else if (P == 4) slot = (((lane % 16) * 2654435761u) >> 20);
    else if (P == 5) slot = scatter;
    else if (P == 6) slot = (((lane / 2) * 2654435761u) >> 20);
    else slot = lane;
    uint4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        const uint4 v = table[(slot + (uint32_t)i * 64u) & 4095u];
        acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w;
    }
    out[blockIdx.x * 64 + lane] = acc;
}
int main()
{
    uint4 *table, *out;
    hipMalloc(&table, 4096 * 16); hipMalloc(&out, 1024 * 64 * 16);
    hipMemset(table, 1, 4096 * 16);
    const int iters = 256, blocks = 1024;
    k_pattern<0><<<blocks, 64>>>(table, out, iters);
    k_pattern<1><<<blocks, 64>>>(table, out, iters);
    k_pattern<2><<<blocks, 64>>>(table, out, iters);
    k_pattern<4><<<blocks, 64>>>(table, out, iters);
    k_pattern<5><<<blocks, 64>>>(table, out, iters);
    k_pattern<6><<<blocks, 64>>>(table, out, iters);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
