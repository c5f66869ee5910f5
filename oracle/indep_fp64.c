/* indep_fp64.c -- TEST INFRASTRUCTURE ONLY: an independent anchor for the oracle (oracle/dxv_oracle.c).
 *
 * A second, deliberately different statement of Content/Shaders/DXRVoxelizer.hlsl:44-53, :58-85, :132-140:
 * double precision throughout, Moeller-Trumbore ray/triangle intersection (not the watertight shear test),
 * every triangle tested for every ray (no box, no hierarchy, no candidacy rule, no tn <= t), closest hit by
 * minimum t.  It shares no code with the oracle or the product; tests/test_oracle_anchor.py compares the two
 * on the reference's assets and requires every differing voxel to be explained by one of the cases DXR itself
 * leaves implementation-defined (equal-t ties on shared edges, |dot - 0.12| at rounding level, hits within
 * rounding of an edge or of the ray origin).
 *
 * Build: oracle/Makefile target libindep64.so.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define API __attribute__((visibility("default")))

typedef struct {
    uint32_t T;
    double *ax, *ay, *az, *e1x, *e1y, *e1z, *e2x, *e2y, *e2z;   /* vertex 0 and the two edges, normalised space */
    double* nrm;                                                /* T x 9 vertex normals */
} scene64;

/* vb: V x {pos.xyz, nrm.xyz} float32; ib: 3T indices; bound = {centre.xyz, half extent} as the application computes
 * it (Content/Voxelizer.cpp:52-57); p' = (p - c) / w (:304-306). */
API scene64* i64_create(const float* vb, const uint32_t* ib, uint32_t T, const float bound[4])
{
    scene64* s = (scene64*)calloc(1, sizeof(scene64));
    s->T = T;
    double** arr[9] = {&s->ax, &s->ay, &s->az, &s->e1x, &s->e1y, &s->e1z, &s->e2x, &s->e2y, &s->e2z};
    for (int i = 0; i < 9; ++i) *arr[i] = (double*)malloc(sizeof(double) * T);
    s->nrm = (double*)malloc(sizeof(double) * 9 * (size_t)T);
    const double c[3] = {bound[0], bound[1], bound[2]}, w = bound[3];
    for (uint32_t k = 0; k < T; ++k) {
        double p[3][3];
        for (int v = 0; v < 3; ++v) {
            const float* src = vb + 6 * (size_t)ib[3 * (size_t)k + v];
            for (int a = 0; a < 3; ++a) { p[v][a] = ((double)src[a] - c[a]) / w; s->nrm[9 * (size_t)k + 3 * v + a] = src[3 + a]; }
        }
        s->ax[k] = p[0][0]; s->ay[k] = p[0][1]; s->az[k] = p[0][2];
        s->e1x[k] = p[1][0] - p[0][0]; s->e1y[k] = p[1][1] - p[0][1]; s->e1z[k] = p[1][2] - p[0][2];
        s->e2x[k] = p[2][0] - p[0][0]; s->e2y[k] = p[2][1] - p[0][1]; s->e2z[k] = p[2][2] - p[0][2];
    }
    return s;
}

API void i64_destroy(scene64* s)
{
    if (!s) return;
    free(s->ax); free(s->ay); free(s->az); free(s->e1x); free(s->e1y); free(s->e1z); free(s->e2x); free(s->e2y); free(s->e2z);
    free(s->nrm); free(s);
}

static void ray_of(uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz, double o[3], double d[3])
{
    o[0] = (ix + 0.5) / N * 2.0 - 1.0;                 /* hlsl:46 */
    o[1] = -((iy + 0.5) / N * 2.0 - 1.0);              /* hlsl:49 */
    o[2] = (iz + 0.5) / N * 2.0 - 1.0;
    const double l = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
    for (int a = 0; a < 3; ++a) d[a] = o[a] / l;       /* hlsl:52 */
}

/* Moeller-Trumbore for the ray's supporting LINE: t and barycentrics (b1, b2 = weights of vertex 1, 2) wherever the
 * line meets the triangle's plane; returns 0 only for a ray parallel to the plane. */
static inline int plane_hit(const scene64* s, uint32_t k, const double o[3], const double d[3], double* t, double* b1, double* b2)
{
    const double e1[3] = {s->e1x[k], s->e1y[k], s->e1z[k]}, e2[3] = {s->e2x[k], s->e2y[k], s->e2z[k]};
    const double px = d[1] * e2[2] - d[2] * e2[1], py = d[2] * e2[0] - d[0] * e2[2], pz = d[0] * e2[1] - d[1] * e2[0];
    const double det = e1[0] * px + e1[1] * py + e1[2] * pz;
    if (det == 0.0) return 0;
    const double inv = 1.0 / det;
    const double tx = o[0] - s->ax[k], ty = o[1] - s->ay[k], tz = o[2] - s->az[k];
    const double u = (tx * px + ty * py + tz * pz) * inv;
    const double qx = ty * e1[2] - tz * e1[1], qy = tz * e1[0] - tx * e1[2], qz = tx * e1[1] - ty * e1[0];
    const double v = (d[0] * qx + d[1] * qy + d[2] * qz) * inv;
    *t = (e2[0] * qx + e2[1] * qy + e2[2] * qz) * inv;
    *b1 = u; *b2 = v;
    return 1;
}

static double dot_normal(const scene64* s, uint32_t k, double b1, double b2, const double d[3])
{
    const double* n = s->nrm + 9 * (size_t)k;
    double v[3];
    for (int a = 0; a < 3; ++a) v[a] = n[a] + b1 * (n[3 + a] - n[a]) + b2 * (n[6 + a] - n[a]);   /* hlsl:114-116 */
    const double l = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    return (v[0] * d[0] + v[1] * d[1] + v[2] * d[2]) / l;                                    /* hlsl:137-138 */
}

/* Whole grid.  occ[N^3] (hlsl:64-67 order), and per voxel: t of the closest hit (inf = miss), its triangle, the
 * predicate value dot(normalize(n), dir) and the second-closest t (inf = none). */
API void i64_voxelize(const scene64* s, uint32_t N, uint8_t* occ, double* tbest, uint32_t* kbest, double* dotbest, double* tsecond)
{
    const int64_t rows = (int64_t)N * N;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t row = 0; row < rows; ++row) {
        const uint32_t iz = (uint32_t)(row / N), iy = (uint32_t)(row % N);
        for (uint32_t ix = 0; ix < N; ++ix) {
            double o[3], d[3];
            ray_of(N, ix, iy, iz, o, d);
            double t1 = INFINITY, t2 = INFINITY, B1 = 0, B2 = 0;
            uint32_t k1 = UINT32_MAX;
            for (uint32_t k = 0; k < s->T; ++k) {
                double t, b1, b2;
                if (!plane_hit(s, k, o, d, &t, &b1, &b2)) continue;
                if (b1 < 0.0 || b2 < 0.0 || b1 + b2 > 1.0 || !(t > 0.0 && t < 10000.0)) continue;   /* hlsl:76-77, both faces */
                if (t < t1) { t2 = t1; t1 = t; k1 = k; B1 = b1; B2 = b2; }
                else if (t < t2) t2 = t;
            }
            const size_t id = (size_t)row * N + ix;
            double dn = 0.0;
            if (k1 != UINT32_MAX) dn = dot_normal(s, k1, B1, B2, d);
            occ[id] = (k1 != UINT32_MAX && dn > 0.12) ? 1 : 0;                                  /* hlsl:5, :138 */
            tbest[id] = t1; kbest[id] = k1; dotbest[id] = dn; tsecond[id] = t2;
        }
    }
}

/* One (voxel, triangle) pair seen through the independent arithmetic: out = {t, b0, b1, b2, dot, parallel?}.  b0 = 1 - b1 - b2;
 * "inside" is min(b0, b1, b2) >= 0; the caller classifies how far a disputed hit is from an edge / from the origin. */
API void i64_probe(const scene64* s, uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz, uint32_t k, double out[6])
{
    double o[3], d[3], t = 0, b1 = 0, b2 = 0;
    ray_of(N, ix, iy, iz, o, d);
    const int ok = plane_hit(s, k, o, d, &t, &b1, &b2);
    out[0] = t; out[1] = 1.0 - b1 - b2; out[2] = b1; out[3] = b2;
    out[4] = ok ? dot_normal(s, k, b1, b2, d) : 0.0;
    out[5] = ok ? 0.0 : 1.0;
}

/* distance scale of triangle k (longest edge), to turn barycentric margins into lengths */
API double i64_tri_size(const scene64* s, uint32_t k)
{
    const double a = sqrt(s->e1x[k] * s->e1x[k] + s->e1y[k] * s->e1y[k] + s->e1z[k] * s->e1z[k]);
    const double b = sqrt(s->e2x[k] * s->e2x[k] + s->e2y[k] * s->e2y[k] + s->e2z[k] * s->e2z[k]);
    const double cx = s->e2x[k] - s->e1x[k], cy = s->e2y[k] - s->e1y[k], cz = s->e2z[k] - s->e1z[k];
    const double c = sqrt(cx * cx + cy * cy + cz * cz);
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

/* The same for a list of voxels (ids in hlsl:64-67 order): what i64_voxelize computes, for samples of grids that are too
 * large to trace whole with every triangle tested for every ray (1 M triangles at 512^3). */
API void i64_voxels(const scene64* s, uint32_t N, uint32_t n, const uint64_t* ids, uint8_t* occ, double* tbest, uint32_t* kbest,
                    double* dotbest, double* tsecond)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t j = 0; j < (int64_t)n; ++j) {
        const uint64_t id = ids[j];
        const uint32_t ix = (uint32_t)(id % N), iy = (uint32_t)((id / N) % N), iz = (uint32_t)(id / ((uint64_t)N * N));
        double o[3], d[3];
        ray_of(N, ix, iy, iz, o, d);
        double t1 = INFINITY, t2 = INFINITY, B1 = 0, B2 = 0;
        uint32_t k1 = UINT32_MAX;
        for (uint32_t k = 0; k < s->T; ++k) {
            double t, b1, b2;
            if (!plane_hit(s, k, o, d, &t, &b1, &b2)) continue;
            if (b1 < 0.0 || b2 < 0.0 || b1 + b2 > 1.0 || !(t > 0.0 && t < 10000.0)) continue;
            if (t < t1) { t2 = t1; t1 = t; k1 = k; B1 = b1; B2 = b2; }
            else if (t < t2) t2 = t;
        }
        double dn = 0.0;
        if (k1 != UINT32_MAX) dn = dot_normal(s, k1, B1, B2, d);
        occ[j] = (k1 != UINT32_MAX && dn > 0.12) ? 1 : 0;
        tbest[j] = t1; kbest[j] = k1; dotbest[j] = dn; tsecond[j] = t2;
    }
}

/* The second occupancy rule (north_star's "axis-aligned ray, hit count"; no reference counterpart): occ = parity of the
 * number of triangles the +X ray from the voxel centre crosses.  Independent statement: float64, every triangle against
 * every grid row's line (all voxels of a row share it), a crossing counts when the point lies STRICTLY inside the triangle
 * and strictly right of the voxel centre -- no fill rule, no box, no hierarchy.  Where a line passes within `eps` of an
 * edge (or a voxel centre within eps of a crossing) the count is a matter of the fill rule: those voxels are flagged
 * (near = 1) instead of judged.  rows: nrows pairs (iy, iz); occ / near: nrows x N. */
API void i64_parity_rows(const scene64* s, uint32_t N, uint32_t nrows, const uint32_t* rows, double eps, uint8_t* occ, uint8_t* near)
{
#pragma omp parallel
    {
        double* xs = (double*)malloc(sizeof(double) * 4096);
        uint8_t* fl = (uint8_t*)malloc(4096);
        size_t cap = 4096;
#pragma omp for schedule(dynamic, 1)
        for (int64_t r = 0; r < (int64_t)nrows; ++r) {
            const uint32_t iy = rows[2 * r], iz = rows[2 * r + 1];
            double o[3], d[3] = {1.0, 0.0, 0.0}, dummy[3];
            ray_of(N, 0, iy, iz, o, dummy);
            o[0] = -4.0;                                       /* far left of the scene: t = x + 4 */
            size_t m = 0;
            for (uint32_t k = 0; k < s->T; ++k) {
                double t, b1, b2;
                if (!plane_hit(s, k, o, d, &t, &b1, &b2)) {
                    /* line parallel to the plane: it may run inside the triangle's plane -- flag the whole row's stretch if the
                     * line lies within eps of the plane and overlaps the triangle's extent in y and z */
                    const double nx = s->e1y[k] * s->e2z[k] - s->e1z[k] * s->e2y[k], ny = s->e1z[k] * s->e2x[k] - s->e1x[k] * s->e2z[k],
                                 nz = s->e1x[k] * s->e2y[k] - s->e1y[k] * s->e2x[k];
                    const double nn = sqrt(nx * nx + ny * ny + nz * nz);
                    if (nn > 0.0) {
                        const double dist = fabs((o[1] - s->ay[k]) * ny + (o[2] - s->az[k]) * nz + (o[0] - s->ax[k]) * nx) / nn;
                        if (dist <= eps) {
                            if (m == cap) { cap *= 2; xs = (double*)realloc(xs, sizeof(double) * cap); fl = (uint8_t*)realloc(fl, cap); }
                            xs[m] = INFINITY; fl[m] = 2; ++m;    /* (flags every voxel of the row: rare, conservative) */
                        }
                    }
                    continue;
                }
                const double b0 = 1.0 - b1 - b2;
                /* distances of the crossing point to the three edges: barycentric x altitude = b x (2 area / |edge|) */
                const double cx = s->e1y[k] * s->e2z[k] - s->e1z[k] * s->e2y[k], cy = s->e1z[k] * s->e2x[k] - s->e1x[k] * s->e2z[k],
                             cz = s->e1x[k] * s->e2y[k] - s->e1y[k] * s->e2x[k];
                const double area2 = sqrt(cx * cx + cy * cy + cz * cz);
                const double l1 = sqrt(s->e1x[k] * s->e1x[k] + s->e1y[k] * s->e1y[k] + s->e1z[k] * s->e1z[k]);
                const double l2 = sqrt(s->e2x[k] * s->e2x[k] + s->e2y[k] * s->e2y[k] + s->e2z[k] * s->e2z[k]);
                const double ex = s->e2x[k] - s->e1x[k], ey = s->e2y[k] - s->e1y[k], ez = s->e2z[k] - s->e1z[k];
                const double l0 = sqrt(ex * ex + ey * ey + ez * ez);
                /* b0 is the weight of vertex 0: distance to the edge opposite vertex 0 (length l0); b1: edge v0-v2 (l2); b2: edge v0-v1 (l1) */
                const double d0 = l0 > 0.0 ? b0 * area2 / l0 : 0.0, d1 = l2 > 0.0 ? b1 * area2 / l2 : 0.0, d2 = l1 > 0.0 ? b2 * area2 / l1 : 0.0;
                double dm = d0 < d1 ? d0 : d1;
                dm = dm < d2 ? dm : d2;
                if (dm < -eps) continue;                       /* clearly outside */
                if (m == cap) { cap *= 2; xs = (double*)realloc(xs, sizeof(double) * cap); fl = (uint8_t*)realloc(fl, cap); }
                xs[m] = t - 4.0;
                /* 0: clearly inside; within eps of an edge (either side): 1 when the triangle faces -X, 3 when it faces +X */
                fl[m] = dm <= eps ? (cx < 0.0 ? 1 : 3) : 0;
                ++m;
            }
            /* Near-edge crossings in pairs: a line through the shared edge of exactly two triangles crosses the surface once
             * when both face the same way along X (one crossing: keep one of them as a clear crossing) and not at all, or
             * twice, when they face opposite ways (a fold seen edge-on: parity unchanged, drop both).  Anything else near an
             * edge -- a mesh boundary, a vertex with its fan of triangles -- stays flagged. */
            for (size_t a = 0; a < m; ++a) {
                if (fl[a] != 1 && fl[a] != 3) continue;
                size_t mate = m, others = 0;
                for (size_t b = 0; b < m; ++b)
                    if (b != a && (fl[b] == 1 || fl[b] == 3 || fl[b] == 4) && fabs(xs[b] - xs[a]) <= 2.0 * eps) { ++others; mate = b; }
                if (others != 1 || fl[mate] == 4) continue;
                if (fl[a] == fl[mate]) { fl[a] = 0; fl[mate] = 5; }     /* one clear crossing; 5 = dropped */
                else { fl[a] = 5; fl[mate] = 5; }
            }
            for (uint32_t ix = 0; ix < N; ++ix) {
                const double ox = (ix + 0.5) / N * 2.0 - 1.0;
                uint32_t c = 0;
                uint8_t nr = 0;
                for (size_t j = 0; j < m; ++j) {
                    if (fl[j] == 2) { nr = 1; continue; }
                    if (fabs(xs[j] - ox) <= eps) nr = 1;               /* the voxel centre lies on the surface (t > 0 is at rounding level) */
                    if (fl[j] == 5) continue;
                    if (xs[j] > ox) { if (fl[j]) nr = 1; else ++c; }
                }
                occ[(size_t)r * N + ix] = (uint8_t)(c & 1u);
                near[(size_t)r * N + ix] = nr;
            }
        }
        free(xs); free(fl);
    }
}
