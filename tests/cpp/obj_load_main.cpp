// obj_load_main.cpp -- drives dxv_obj_load from plain C++ (sanitizer harness, tools/sanitize_cpu.sh).
// usage: obj_load_main file.obj [repeats]   prints counts, AABB and an FNV-1a hash of VB and IB.
#include "../../include/dxv.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

static uint64_t fnv1a(const void* p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s file.obj [repeats]\n", argv[0]); return 2; }
    const int repeats = argc > 2 ? atoi(argv[2]) : 1;
    for (int r = 0; r < repeats; ++r) {
        float* vb = nullptr; uint32_t* ib = nullptr; uint32_t nv = 0, ni = 0; float aabb[6] = {0};
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = dxv_obj_load(argv[1], &vb, &nv, &ib, &ni, aabb);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rc) { fprintf(stderr, "dxv_obj_load failed: %d\n", rc); return 1; }
        printf("verts %u indices %u aabb %g %g %g %g %g %g vb %016llx ib %016llx  %.1f ms\n", nv, ni, aabb[0], aabb[1], aabb[2],
               aabb[3], aabb[4], aabb[5], (unsigned long long)fnv1a(vb, (size_t)nv * 24),
               (unsigned long long)fnv1a(ib, (size_t)ni * 4), ms);
        dxv_free(vb); dxv_free(ib);
    }
    return 0;
}
