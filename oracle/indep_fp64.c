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
