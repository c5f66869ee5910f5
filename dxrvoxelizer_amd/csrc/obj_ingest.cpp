// obj_ingest.cpp -- mesh ingest for the voxelizer: dxv_obj_load / dxv_free (include/dxv.h).
//
// Produces exactly what XUSG::ObjLoader::Import(file, needNorm=true, needAABB=true, forDX=true,
// swapYZ=false) hands to Voxelizer::Init (XUSG/Optional/XUSGObjLoader.cpp:18-40,
// Content/Voxelizer.cpp:46-57): an interleaved {pos, nrm} vertex buffer with z negated, an index
// buffer whose whole array is reversed, per-vertex normals (split per distinct vn, or recomputed
// from unweighted face normals when the file has none) and the AABB.
//
// The file is mmap'ed and cut at line starts into spans of about 1 MiB that worker threads take
// from a shared counter (the calling thread works too, so a slow thread start costs nothing);
// every span is parsed on its own and the results are spliced in file order, so the output does
// not depend on the number of threads.  Numbers: a decimal whose digit string is <= 2^24 with
// |exponent| <= 10 is formed with ONE correctly rounded float operation on exact operands; up to
// 15 digits and |exponent| <= 22 with one correctly rounded double operation followed by the
// double->float rounding, unless that double lies next to a float rounding midpoint; both give
// the value strtof/fscanf("%f") returns, and every other spelling goes through strtof.
// The stage whose float result depends on the order of accumulation (recomputeNormals,
// XUSGObjLoader.cpp:337-384) is parallel over VERTEX ranges, each vertex still receiving its
// face normals in triangle order.  Texture coordinates are skipped (the reference never stores
// them, XUSGObjLoader.cpp:113,133).
#include "../../include/dxv.h"

#include <algorithm>
#include <atomic>
#include <new>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr long long kNoVn = INT64_MIN;
constexpr int32_t kNoVn32 = INT32_MIN;
constexpr long long kIndexLimit = 0x7ffffff0ll;    // vertices, normals and corners each stay below this

struct Corner { long long v, vn; };                // as parsed: 1-based, negative = relative to the end
struct Corner32 { int32_t v, vn; };                // as stored per span (out-of-range values saturate)

struct Span {                                      // one piece of the file, and what was found there
    const char* begin;
    const char* end;
    std::vector<float> pos;                        // file order, z negated
    std::vector<float> vn;                         // file order, z negated
    std::vector<Corner32> corners;                 // 3 per triangle after fan triangulation
};

inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }
inline bool is_digit(char c) { return (unsigned)(c - '0') < 10u; }

inline const char* skip_blanks(const char* p, const char* e) { while (p < e && is_blank(*p)) ++p; return p; }

// Decimal integer at p (optional sign).  Returns false when no digit is there.
inline bool parse_int(const char*& p, const char* e, long long& out)
{
    const char* q = p;
    bool neg = false;
    if (q < e && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
    if (q >= e || !is_digit(*q)) return false;
    unsigned long long m = 0;
    constexpr unsigned long long kSat = 1ull << 40;       // beyond every index or exponent this file accepts: saturate, never wrap
    while (q < e && is_digit(*q)) { if (m < kSat) m = m * 10u + (unsigned)(*q - '0'); ++q; }
    if (m > kSat) m = kSat;
    out = neg ? -(long long)m : (long long)m;
    p = q;
    return true;
}

// "v", "v/vt", "v//vn", "v/vt/vn".  Returns false at end of line / non-numeric token.
inline bool next_corner(const char*& p, const char* e, Corner& c)
{
    p = skip_blanks(p, e);
    if (!parse_int(p, e, c.v)) return false;
    c.vn = kNoVn;
    if (p < e && *p == '/') {
        ++p;
        long long vt;
        (void)parse_int(p, e, vt);
        if (p < e && *p == '/') {
            ++p;
            long long n;
            if (parse_int(p, e, n)) c.vn = n;
        }
    }
    return true;
}

const float kPow10[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f}; // all exact in binary32

const double kPow10d[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                            1e17, 1e18, 1e19, 1e20, 1e21, 1e22};                     // all exact in binary64

float parse_float_slow(const char*& p, const char* e)
{
    char buf[128];
    size_t n = 0;
    const char* q = p;
    while (q < e && !is_blank(*q) && n + 1 < sizeof(buf)) buf[n++] = *q++;
    buf[n] = 0;
    char* endp;
    const float v = strtof(buf, &endp);
    p += endp - buf;
    return v;
}

// One number of a "v"/"vn" line; a missing number reads as 0 like an unassigned fscanf target
// of the zero-initialised vectors would (XUSGObjLoader.cpp:194-216).
inline float parse_float(const char*& p, const char* e)
{
    p = skip_blanks(p, e);
    const char* q = p;
    bool neg = false;
    if (q < e && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
    uint64_t m = 0;
    int digits = 0, exp10 = 0;
    bool any = false;
    while (q < e && is_digit(*q)) { any = true; if (m || *q != '0') { m = m * 10u + (unsigned)(*q - '0'); ++digits; } ++q; if (digits > 15) break; }
    if (digits <= 15 && q < e && *q == '.') {
        ++q;
        while (q < e && is_digit(*q)) { any = true; if (m || *q != '0') { m = m * 10u + (unsigned)(*q - '0'); ++digits; } --exp10; ++q; if (digits > 15) break; }
    }
    if (!any || digits > 15) return parse_float_slow(p, e);
    if (q < e && (*q == 'e' || *q == 'E')) {
        const char* r = q + 1;
        long long x;
        if (!parse_int(r, e, x) || x > 100 || x < -100) return parse_float_slow(p, e);
        exp10 += (int)x;
        q = r;
    }
    if (q < e && !is_blank(*q)) return parse_float_slow(p, e);    // hex floats, "inf", trailing junk
    float v;
    if (m == 0) v = 0.0f;
    else if (m <= (1u << 24) && exp10 >= -10 && exp10 <= 10)
        v = exp10 < 0 ? (float)m / kPow10[-exp10] : (float)m * kPow10[exp10];
    else if (exp10 >= -22 && exp10 <= 22) {
        // m < 10^15 < 2^53 and 10^|exp10| are exact doubles: d is the correctly rounded double.
        // Rounding d again to binary32 is the correctly rounded float unless d sits within one
        // double ulp of a binary32 rounding midpoint -- those spellings go to strtof.
        const double d = exp10 < 0 ? (double)m / kPow10d[-exp10] : (double)m * kPow10d[exp10];
        uint64_t bits;
        memcpy(&bits, &d, 8);
        const uint32_t low = (uint32_t)(bits & 0x1fffffffu);
        if (d < 1.1754943508222875e-38 || d >= 1.7014118346046923e38 || (low >= 0x0fffffffu && low <= 0x10000001u))
            return parse_float_slow(p, e);
        v = (float)d;
    } else return parse_float_slow(p, e);
    p = q;
    return neg ? -v : v;
}

inline Corner32 narrow(const Corner& c)
{
    Corner32 r;
    r.v = (int32_t)(c.v > kIndexLimit ? kIndexLimit : (c.v < -kIndexLimit ? -kIndexLimit : c.v));
    r.vn = c.vn == kNoVn ? kNoVn32 : (int32_t)(c.vn > kIndexLimit ? kIndexLimit : (c.vn < -kIndexLimit ? -kIndexLimit : c.vn));
    return r;
}

void parse_span(Span& s)
{
    const size_t bytes = (size_t)(s.end - s.begin);
    const char* line = s.begin;
    while (line < s.end) {
        const char* eol = static_cast<const char*>(memchr(line, '\n', (size_t)(s.end - line)));
        if (!eol) eol = s.end;
        const char* p = skip_blanks(line, eol);
        if (eol - p >= 2) {
            if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
                const char* q = p + 1;
                const float x = parse_float(q, eol), y = parse_float(q, eol), z = parse_float(q, eol);
                if (s.pos.empty()) s.pos.reserve(bytes / 24 * 3);                  // typical line lengths;
                s.pos.push_back(x); s.pos.push_back(y); s.pos.push_back(-z);        // XUSGObjLoader.cpp:198
            } else if (p[0] == 'v' && p[1] == 'n' && eol - p >= 3 && (p[2] == ' ' || p[2] == '\t')) {
                const char* q = p + 2;
                const float x = parse_float(q, eol), y = parse_float(q, eol), z = parse_float(q, eol);
                if (s.vn.empty()) s.vn.reserve(bytes / 24 * 3);                    // the vectors grow past them
                s.vn.push_back(x); s.vn.push_back(y); s.vn.push_back(-z);          // :213
            } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
                const char* q = p + 1;
                Corner first, prev, cur;
                if (next_corner(q, eol, first) && next_corner(q, eol, prev)) {
                    if (s.corners.empty()) s.corners.reserve(bytes / 16 * 3);
                    const Corner32 f32 = narrow(first);
                    Corner32 p32 = narrow(prev);
                    while (next_corner(q, eol, cur)) {                                // fan: :263-297
                        const Corner32 c32 = narrow(cur);
                        s.corners.push_back(f32); s.corners.push_back(p32); s.corners.push_back(c32);
                        p32 = c32;
                    }
                }
            }
        }
        line = eol + 1;
    }
}

int worker_count(size_t bytes)
{
    if (const char* env = getenv("DXV_OBJ_THREADS")) {
        const int n = atoi(env);
        if (n >= 1) return n > 256 ? 256 : n;
    }
    int n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    // a container's CPU quota is usually below the affinity mask (cgroup v2: "<quota> <period>")
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
            const int q = (int)((quota + period - 1) / period);
            if (q < n) n = q;
        }
        fclose(f);
    }
    const size_t bySize = bytes / (1u << 20) + 1;       // below 1 MiB per thread a thread is not worth starting
    if ((size_t)n > bySize) n = (int)bySize;
    if (n > 32) n = 32;
    return n < 1 ? 1 : n;
}

// f(i) for every i in [0, items): the items are handed out through one counter to `workers`
// threads, the caller being one of them.
template <class F> void parallel_items(int workers, size_t items, F f)
{
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};                      // an exception in a worker (bad_alloc) must not reach std::terminate
    auto run = [&] {
        try {
            for (size_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < items;) f(i);
        } catch (...) { failed.store(true); next.store(items); }
    };
    std::vector<std::thread> th;
    const size_t extra = items < (size_t)workers ? (items ? items - 1 : 0) : (size_t)workers - 1;
    th.reserve(extra);
    try {
        for (size_t i = 0; i < extra; ++i) th.emplace_back(run);
    } catch (...) {}                                      // thread limit reached: the threads that started (and the caller) do the work
    run();
    for (auto& t : th) t.join();
    if (failed.load()) throw std::bad_alloc();
}

struct Mapping {
    const char* data = nullptr;
    size_t size = 0;
    ~Mapping() { if (data && size) munmap(const_cast<char*>(data), size); }
};

struct Owned {                                      // malloc'ed output array, released unless handed over
    void* p = nullptr;
    ~Owned() { free(p); }
    void* release() { void* r = p; p = nullptr; return r; }
};

} // namespace

extern "C" {

void dxv_free(void* p) { free(p); }

static int obj_load_impl(const char* path, float** vbOut, uint32_t* numVerts, uint32_t** ibOut, uint32_t* numIndices, float aabb[6]);

// C entry point: never throws (include/dxv_voxelizer.hpp promises bool returns like the reference's Import);
// 4 = out of memory or any other C++ exception.
int dxv_obj_load(const char* path, float** vbOut, uint32_t* numVerts, uint32_t** ibOut, uint32_t* numIndices,
                 float aabb[6])
{
    try {
        return obj_load_impl(path, vbOut, numVerts, ibOut, numIndices, aabb);
    } catch (...) {
        if (vbOut) *vbOut = nullptr;
        if (ibOut) *ibOut = nullptr;
        if (numVerts) *numVerts = 0;
        if (numIndices) *numIndices = 0;
        return 4;
    }
}

static int obj_load_impl(const char* path, float** vbOut, uint32_t* numVerts, uint32_t** ibOut, uint32_t* numIndices, float aabb[6])
{
    if (!path || !vbOut || !numVerts || !ibOut || !numIndices) return 1;
    *vbOut = nullptr; *ibOut = nullptr; *numVerts = 0; *numIndices = 0;

    Mapping map;
    {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) return 1;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return 1; }
        map.size = (size_t)st.st_size;
        if (map.size) {
            void* m = mmap(nullptr, map.size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); map.size = 0; return 1; }
            map.data = static_cast<const char*>(m);
        }
        close(fd);
    }
    if (!map.size) return 2;

    const int W = worker_count(map.size);
    const size_t S = W == 1 ? 1 : std::min<size_t>((size_t)W * 8, map.size / (512u << 10) + 1);   // spans
    std::vector<Span> spans(S);
    {
        const char* const b = map.data;
        const char* const e = map.data + map.size;
        const char* at = b;
        for (size_t i = 0; i < S; ++i) {
            spans[i].begin = at;
            const char* c = i + 1 == S ? e : b + map.size / S * (i + 1);
            if (c < at) c = at;
            while (c < e && c > b && c[-1] != '\n') ++c;             // first line start at or after the target
            spans[i].end = at = c;
        }
    }
    parallel_items(W, S, [&](size_t i) { parse_span(spans[i]); });

    // splice the spans in file order
    std::vector<size_t> posAt(S + 1, 0), vnAt(S + 1, 0), corAt(S + 1, 0);
    for (size_t i = 0; i < S; ++i) {
        posAt[i + 1] = posAt[i] + spans[i].pos.size() / 3;
        vnAt[i + 1] = vnAt[i] + spans[i].vn.size() / 3;
        corAt[i + 1] = corAt[i] + spans[i].corners.size();
    }
    const size_t V0s = posAt[S], NNs = vnAt[S], nIdx = corAt[S];
    if (!V0s || !nIdx || nIdx > (size_t)kIndexLimit || V0s > (size_t)kIndexLimit || NNs > (size_t)kIndexLimit) return 2;
    const uint32_t V0 = (uint32_t)V0s, NN = (uint32_t)NNs;

    Owned vbMem, ibMem;
    size_t vbCap = (size_t)V0 + (NN ? (size_t)V0 / 4 + 16 : 0);                    // room for split vertices
    vbMem.p = malloc(vbCap * 6 * sizeof(float));
    ibMem.p = malloc(nIdx * sizeof(uint32_t));
    if (!vbMem.p || !ibMem.p) return 4;
    float* vb = static_cast<float*>(vbMem.p);
    uint32_t* ib = static_cast<uint32_t*>(ibMem.p);
    std::vector<float> vn((size_t)NN * 3);
    std::vector<uint32_t> nib(NN ? nIdx : 0);
    std::atomic<int> bad{0};
    parallel_items(W, S, [&](size_t i) {
        Span& s = spans[i];
        float* d = vb + posAt[i] * 6;
        for (size_t k = 0, n = s.pos.size() / 3; k < n; ++k) {
            d[k * 6] = s.pos[k * 3]; d[k * 6 + 1] = s.pos[k * 3 + 1]; d[k * 6 + 2] = s.pos[k * 3 + 2];
            d[k * 6 + 3] = d[k * 6 + 4] = d[k * 6 + 5] = 0.0f;
        }
        if (!s.vn.empty()) memcpy(&vn[vnAt[i] * 3], s.vn.data(), s.vn.size() * sizeof(float));
        const size_t c0 = corAt[i];
        for (size_t k = 0; k < s.corners.size(); ++k) {
            const long long v = s.corners[k].v;
            const long long r = v < 0 ? v + (long long)V0 : v - 1;                // :243
            if (r < 0 || r >= (long long)V0) { bad.store(1); return; }
            ib[c0 + k] = (uint32_t)r;
            if (NN) {
                const long long n = s.corners[k].vn;
                const long long rn = n == kNoVn32 ? 0 : (n < 0 ? n + (long long)NN : n - 1);
                if (rn < 0 || rn >= (long long)NN) { bad.store(1); return; }
                nib[c0 + k] = (uint32_t)rn;
            }
        }
        std::vector<float>().swap(s.pos);
        std::vector<float>().swap(s.vn);
        std::vector<Corner32>().swap(s.corners);
    });
    if (bad.load()) return 3;

    uint32_t V = V0;
    if (NN) {
        // one normal per vertex; a vertex met again with another vn is duplicated (:300-335).
        // The duplicates are numbered in corner order, so this walk stays sequential.
        std::vector<uint32_t> owner(V0, UINT32_MAX);
        for (size_t i = 0; i < nIdx; ++i) {
            const uint32_t ni = nib[i];
            uint32_t vi = ib[i];
            if (owner[vi] == ni) continue;
            if (owner[vi] != UINT32_MAX) {
                if (V >= (uint32_t)kIndexLimit) return 2;
                if (V == vbCap) {
                    vbCap += vbCap / 2 + 16;
                    void* g = realloc(vbMem.p, vbCap * 6 * sizeof(float));
                    if (!g) return 4;
                    vbMem.p = g;
                    vb = static_cast<float*>(g);
                }
                memcpy(vb + (size_t)V * 6, vb + (size_t)vi * 6, 24);
                ib[i] = vi = V++;
            } else owner[vi] = ni;
            const float x = vn[(size_t)ni * 3], y = vn[(size_t)ni * 3 + 1], z = vn[(size_t)ni * 3 + 2];
            const float l = sqrtf(x * x + y * y + z * z);
            float* d = vb + (size_t)vi * 6 + 3;
            d[0] = x / l; d[1] = y / l; d[2] = z / l;
        }
    }

    // forDX: the WHOLE index array is reversed (:227): winding and triangle order flip
    std::reverse(ib, ib + nIdx);

    const size_t T = nIdx / 3;
    const size_t R = std::min<size_t>((size_t)W, 8);                              // vertex ranges
    if (!NN) {
        // recomputeNormals (:337-384): n = normalize(cross(v1-v0, v2-v1)) added unweighted to the
        // three corners, then one normalise per vertex.  Parallel over vertex ranges: a range
        // walks all triangles in order and serves the corners it owns, so every vertex adds its
        // faces in triangle order whatever the number of ranges.
        parallel_items(W, R, [&](size_t w) {
            const uint32_t lo = (uint32_t)((uint64_t)V * w / R), hi = (uint32_t)((uint64_t)V * (w + 1) / R);
            const uint32_t width = hi - lo;
            if (!width) return;
            for (size_t t = 0; t < T; ++t) {
                const uint32_t i0 = ib[t * 3], i1 = ib[t * 3 + 1], i2 = ib[t * 3 + 2];
                const bool m0 = i0 - lo < width, m1 = i1 - lo < width, m2 = i2 - lo < width;
                if (!(m0 | m1 | m2)) continue;
                const float* a = vb + (size_t)i0 * 6;
                const float* b = vb + (size_t)i1 * 6;
                const float* c = vb + (size_t)i2 * 6;
                const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
                const float e2[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
                float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                const float l = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                n[0] /= l; n[1] /= l; n[2] /= l;
                const uint32_t idx[3] = {i0, i1, i2};
                const bool mine[3] = {m0, m1, m2};
                for (int k = 0; k < 3; ++k)
                    if (mine[k]) {
                        float* d = vb + (size_t)idx[k] * 6 + 3;
                        d[0] += n[0]; d[1] += n[1]; d[2] += n[2];
                    }
            }
            for (uint32_t i = lo; i < hi; ++i) {
                float* d = vb + (size_t)i * 6 + 3;
                const float l = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                d[0] /= l; d[1] /= l; d[2] /= l;
            }
        });
    }

    if (aabb) {                                                                   // :386-416
        std::vector<float> part(R * 6);
        parallel_items(W, R, [&](size_t w) {
            const uint32_t lo = (uint32_t)((uint64_t)V * w / R), hi = (uint32_t)((uint64_t)V * (w + 1) / R);
            float* bx = &part[w * 6];
            const uint32_t first = lo < hi ? lo : 0;
            for (int a = 0; a < 3; ++a) bx[a] = bx[3 + a] = vb[(size_t)first * 6 + a];
            for (uint32_t i = lo; i < hi; ++i)
                for (int a = 0; a < 3; ++a) {
                    const float x = vb[(size_t)i * 6 + a];
                    if (x < bx[a]) bx[a] = x;
                    else if (x > bx[3 + a]) bx[3 + a] = x;
                }
        });
        for (int a = 0; a < 6; ++a) aabb[a] = part[a];
        for (size_t w = 1; w < R; ++w)
            for (int a = 0; a < 3; ++a) {
                if (part[w * 6 + a] < aabb[a]) aabb[a] = part[w * 6 + a];
                if (part[w * 6 + 3 + a] > aabb[3 + a]) aabb[3 + a] = part[w * 6 + 3 + a];
            }
    }

    *vbOut = static_cast<float*>(vbMem.release()); *numVerts = V;
    *ibOut = static_cast<uint32_t*>(ibMem.release()); *numIndices = (uint32_t)nIdx;
    return 0;
}

} // extern "C"
