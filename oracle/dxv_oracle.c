/*
 * dxv_oracle.c -- CPU ORACLE for the DXRVoxelizer hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The shipped path (dxrvoxelizer_amd/csrc) does not
 * include, link or call anything in oracle/.
 *
 * PARITY STATUS
 *   - mesh ingest (A1) and bound (A2): PINNED against the reference's own ObjLoader compiled
 *     from /root/reference (oracle/_ref/ref_objloader, see oracle/Makefile) on the three
 *     shipped assets, and against the survey's probe values (tests/test_oracle_objloader.py).
 *   - traversal + predicate (A3-A5): "parity unpinned".  The reference holds no tests, golden
 *     vectors or fixtures for this path and its arithmetic lives in the closed D3D12/DXR driver
 *     reached through XUSGRayTracing.dll (github.com/StarsX/XUSG, no pinned version, binaries
 *     only), which cannot run here.  What follows restates
 *     DXRVoxelizer/Content/Shaders/DXRVoxelizer.hlsl line by line and fixes, once, every choice
 *     DXR leaves implementation-defined (float op order, ray/triangle test, equal-t ties).
 *
 * Plain C, float32 throughout, no fused contraction except the explicit fmaf() calls
 * (compile with -ffp-contract=off; see oracle/Makefile).
 *
 * Reference citations are relative to /root/reference/DXRVoxelizer/.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Canonical constants
 * ---------------------------------------------------------------------------------------- */
#define ORC_THRESHOLD 0.12f   /* Content/Shaders/DXRVoxelizer.hlsl:5   */
#define ORC_TMAX 10000.0f     /* Content/Shaders/DXRVoxelizer.hlsl:77  */
#define ORC_PAD 1.52587890625e-05f /* 2^-16: outward pad of every per-triangle box (canonical) */

enum { ORC_MODE_REFERENCE = 0, ORC_MODE_PARITY = 1 };
enum { ORC_ALGO_BRUTE = 0, ORC_ALGO_BVH = 1, ORC_ALGO_PLAIN = 2 };

typedef struct { float x, y, z; } f3;

/* ==========================================================================================
 * A1 -- mesh ingest.  Restates XUSG/Optional/XUSGObjLoader.cpp (Import :18-40) as called by
 * Content/Voxelizer.cpp:46-47: Import(file, needNorm=true, needAABB=true, forDX=true,
 * swapYZ=false).  Line based where the reference is fscanf-token based; identical on
 * well-formed OBJ text.
 * ======================================================================================== */
typedef struct {
    float* vb;       /* V x 6 floats {pos.xyz, nrm.xyz}; stride 24 (XUSGObjLoader.cpp:25-26) */
    uint32_t V, capV;
    uint32_t* ib;    /* 3T indices */
    uint32_t nIdx, capIdx;
    uint32_t* nidx;  /* per-corner vn index (only when the file has vn) */
    float* normals;  /* file vn list, already z-negated */
    uint32_t numNorm, capNorm;
    uint32_t numTexc;
} objbuild;

static void ob_push_vertex(objbuild* o, const float* six)
{
    if (o->V == o->capV) {
        o->capV = o->capV ? o->capV * 2 : 1024;
        o->vb = (float*)realloc(o->vb, (size_t)o->capV * 6 * sizeof(float));
    }
    memcpy(o->vb + (size_t)o->V * 6, six, 6 * sizeof(float));
    o->V++;
}

static void ob_push_corner(objbuild* o, uint32_t v, uint32_t vn)
{
    if (o->nIdx == o->capIdx) {
        o->capIdx = o->capIdx ? o->capIdx * 2 : 4096;
        o->ib = (uint32_t*)realloc(o->ib, (size_t)o->capIdx * sizeof(uint32_t));
        o->nidx = (uint32_t*)realloc(o->nidx, (size_t)o->capIdx * sizeof(uint32_t));
    }
    o->ib[o->nIdx] = v;
    o->nidx[o->nIdx] = vn;
    o->nIdx++;
}

/* One face corner "v", "v/vt", "v//vn" or "v/vt/vn" (XUSGObjLoader.cpp:230-298). Returns 0 when
 * the token does not start with an integer. Negative indices are relative to the counts seen
 * in the FIRST pass, i.e. the file totals (XUSGObjLoader.cpp:238,243,249,257). */
static int parse_corner(const char** pp, uint32_t numVert, uint32_t numTexc, uint32_t numNorm,
                        uint32_t* v, uint32_t* vn)
{
    const char* p = *pp;
    while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
    char* e;
    long long vi = strtoll(p, &e, 10);
    if (e == p) return 0;
    *v = (uint32_t)(vi < 0 ? vi + (long long)numVert : vi - 1);
    *vn = 0;
    p = e;
    if (*p == '/') {
        ++p;
        if (*p != '/') { /* vt present */
            long long ti = strtoll(p, &e, 10);
            (void)ti; (void)numTexc;
            p = e;
        }
        if (*p == '/') {
            ++p;
            long long ni = strtoll(p, &e, 10);
            if (e != p) *vn = (uint32_t)(ni < 0 ? ni + (long long)numNorm : ni - 1);
            p = e;
        }
    }
    *pp = p;
    return 1;
}

ORC_API void orc_free(void* p) { free(p); }

/* Returns 0 on success.  *vb (V*6 floats) and *ib (nIdx uint32) are malloc'd; free with orc_free.
 * aabb = {min.xyz, max.xyz} (XUSGObjLoader.cpp:386-416). */
ORC_API int orc_obj_load(const char* path, float** vb_out, uint32_t* V_out, uint32_t** ib_out,
                         uint32_t* nIdx_out, float aabb[6])
{
    FILE* f = fopen(path, "r");
    if (!f) return 1;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* text = (char*)malloc((size_t)sz + 2);
    if (fread(text, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(text); return 2; }
    fclose(f);
    text[sz] = '\n';
    text[sz + 1] = 0;

    /* first pass: totals (XUSGObjLoader.cpp:72-164) */
    uint32_t numVert = 0, numTexc = 0, numNorm = 0;
    for (char* p = text; *p;) {
        char* eol = strchr(p, '\n');
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) ++numVert;
        else if (p[0] == 'v' && p[1] == 't' && isspace((unsigned char)p[2])) ++numTexc;
        else if (p[0] == 'v' && p[1] == 'n' && isspace((unsigned char)p[2])) ++numNorm;
        p = eol + 1;
    }

    objbuild o;
    memset(&o, 0, sizeof(o));
    o.numTexc = numTexc;

    /* second pass (XUSGObjLoader.cpp:166-228) */
    for (char* p = text; *p;) {
        char* eol = strchr(p, '\n');
        *eol = 0;
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            float six[6] = {0, 0, 0, 0, 0, 0}; /* normals start at zero (vector resize, :162) */
            char* q = p + 1;
            six[0] = strtof(q, &q);
            six[1] = strtof(q, &q);
            six[2] = strtof(q, &q);
            six[2] = -six[2]; /* forDX: z <- -z (:198) */
            ob_push_vertex(&o, six);
        } else if (p[0] == 'v' && p[1] == 'n' && isspace((unsigned char)p[2])) {
            if (o.numNorm == o.capNorm) {
                o.capNorm = o.capNorm ? o.capNorm * 2 : 1024;
                o.normals = (float*)realloc(o.normals, (size_t)o.capNorm * 3 * sizeof(float));
            }
            char* q = p + 2;
            float* n = o.normals + (size_t)o.numNorm * 3;
            n[0] = strtof(q, &q);
            n[1] = strtof(q, &q);
            n[2] = strtof(q, &q);
            n[2] = -n[2]; /* :213 */
            o.numNorm++;
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            /* fan triangulation (v0, v_{i-1}, v_i) (:263-297) */
            const char* q = p + 1;
            uint32_t v[3], vn[3];
            int k = 0;
            uint32_t cv, cn;
            while (parse_corner(&q, numVert, numTexc, numNorm, &cv, &cn)) {
                if (k < 3) {
                    v[k] = cv; vn[k] = cn; ++k;
                    if (k == 3) {
                        for (int i = 0; i < 3; ++i) ob_push_corner(&o, v[i], vn[i]);
                        v[1] = v[2]; vn[1] = vn[2];
                    }
                } else {
                    v[2] = cv; vn[2] = cn;
                    for (int i = 0; i < 3; ++i) ob_push_corner(&o, v[i], vn[i]);
                    v[1] = v[2]; vn[1] = vn[2];
                }
            }
        }
        p = eol + 1;
    }
    free(text);
    if (o.V != numVert || o.V == 0) { free(o.vb); free(o.ib); free(o.nidx); free(o.normals); return 3; }

    /* computePerVertexNormals (XUSGObjLoader.cpp:300-335): a vertex used with a second, different
     * vn index is split (copied to the end of the VB); runs BEFORE the index reversal. */
    if (o.numNorm) {
        uint32_t* vni = (uint32_t*)malloc((size_t)o.V * sizeof(uint32_t));
        uint32_t nOrig = o.V;
        for (uint32_t i = 0; i < nOrig; ++i) vni[i] = UINT32_MAX;
        for (uint32_t i = 0; i < o.nIdx; ++i) {
            uint32_t vi = o.ib[i];
            if (vni[vi] == o.nidx[i]) continue;
            if (vni[vi] < UINT32_MAX) {
                float six[6];
                memcpy(six, o.vb + (size_t)o.ib[i] * 6, sizeof(six));
                vi = o.V;
                ob_push_vertex(&o, six);
                o.ib[i] = vi;
            } else vni[vi] = o.nidx[i];
            const float* n = o.normals + (size_t)o.nidx[i] * 3;
            float nx = n[0], ny = n[1], nz = n[2];
            const float l = sqrtf(nx * nx + ny * ny + nz * nz);
            nx /= l; ny /= l; nz /= l;
            float* dst = o.vb + (size_t)vi * 6 + 3;
            dst[0] = nx; dst[1] = ny; dst[2] = nz;
        }
        free(vni);
    }

    /* forDX && !swapYZ: reverse the WHOLE index array (:227): flips winding and triangle order */
    for (uint32_t i = 0, j = o.nIdx ? o.nIdx - 1 : 0; i < j; ++i, --j) {
        uint32_t t = o.ib[i]; o.ib[i] = o.ib[j]; o.ib[j] = t;
    }

    /* recomputeNormals when the file has no vn (XUSGObjLoader.cpp:36, :337-384): face normal
     * normalize(cross(v1-v0, v2-v1)) added unweighted to the 3 vertices, then normalised. */
    if (!o.numNorm) {
        const uint32_t numTri = o.nIdx / 3;
        for (uint32_t i = 0; i < numTri; ++i) {
            const float* p0 = o.vb + (size_t)o.ib[i * 3] * 6;
            const float* p1 = o.vb + (size_t)o.ib[i * 3 + 1] * 6;
            const float* p2 = o.vb + (size_t)o.ib[i * 3 + 2] * 6;
            const float e1x = p1[0] - p0[0], e1y = p1[1] - p0[1], e1z = p1[2] - p0[2];
            const float e2x = p2[0] - p1[0], e2y = p2[1] - p1[1], e2z = p2[2] - p1[2];
            float nx = e1y * e2z - e1z * e2y;
            float ny = e1z * e2x - e1x * e2z;
            float nz = e1x * e2y - e1y * e2x;
            const float l = sqrtf(nx * nx + ny * ny + nz * nz);
            nx /= l; ny /= l; nz /= l;
            for (int c = 0; c < 3; ++c) {
                float* vn = o.vb + (size_t)o.ib[i * 3 + c] * 6 + 3;
                vn[0] += nx; vn[1] += ny; vn[2] += nz;
            }
        }
        for (uint32_t i = 0; i < o.V; ++i) {
            float* vn = o.vb + (size_t)i * 6 + 3;
            const float l = sqrtf(vn[0] * vn[0] + vn[1] * vn[1] + vn[2] * vn[2]);
            vn[0] /= l; vn[1] /= l; vn[2] /= l;
        }
    }

    /* computeAABB over every VB position (:386-416) */
    if (aabb) {
        const float* p = o.vb;
        float mn[3] = {p[0], p[1], p[2]}, mx[3] = {p[0], p[1], p[2]};
        for (uint32_t i = 1; i < o.V; ++i) {
            p = o.vb + (size_t)i * 6;
            for (int a = 0; a < 3; ++a) {
                if (p[a] < mn[a]) mn[a] = p[a];
                else if (p[a] > mx[a]) mx[a] = p[a];
            }
        }
        memcpy(aabb, mn, sizeof(mn));
        memcpy(aabb + 3, mx, sizeof(mx));
    }
    free(o.nidx);
    free(o.normals);
    *vb_out = o.vb; *V_out = o.V; *ib_out = o.ib; *nIdx_out = o.nIdx;
    return 0;
}

/* ==========================================================================================
 * A2 -- bound.  Content/Voxelizer.cpp:52-57: centre = (max+min)/2, w = max extent / 2.
 * The TLAS instance transform is inverse(Scale(w)*Translate(c)) (Voxelizer.cpp:304-306), i.e.
 * object -> normalised space p' = (p - c) / w.  Canonical float form: subtract, then divide.
 * ======================================================================================== */
ORC_API void orc_aabb(const float* vb, uint32_t V, float aabb[6])
{
    float mn[3] = {vb[0], vb[1], vb[2]}, mx[3] = {vb[0], vb[1], vb[2]};
    for (uint32_t i = 1; i < V; ++i) {
        const float* p = vb + (size_t)i * 6;
        for (int a = 0; a < 3; ++a) {
            if (p[a] < mn[a]) mn[a] = p[a];
            else if (p[a] > mx[a]) mx[a] = p[a];
        }
    }
    memcpy(aabb, mn, sizeof(mn));
    memcpy(aabb + 3, mx, sizeof(mx));
}

ORC_API void orc_bound(const float aabb[6], float bound[4])
{
    const float ex = aabb[3] - aabb[0], ey = aabb[4] - aabb[1], ez = aabb[5] - aabb[2];
    bound[0] = (aabb[3] + aabb[0]) / 2.0f;
    bound[1] = (aabb[4] + aabb[1]) / 2.0f;
    bound[2] = (aabb[5] + aabb[2]) / 2.0f;
    const float m = ex > (ey > ez ? ey : ez) ? ex : (ey > ez ? ey : ez);
    bound[3] = m / 2.0f;
}

/* ==========================================================================================
 * Scene: triangles pre-mapped to normalised space + canonical padded per-triangle boxes.
 * Replaces the driver BLAS/TLAS of Content/Voxelizer.cpp:264-326 (A3).
 * ======================================================================================== */
typedef struct {
    int32_t left, right; /* internal: child node ids; leaf: left = -1 - first, right = count */
    float lo[3], hi[3];
} orc_node;

typedef struct orc_scene {
    uint32_t T, V;
    float bound[4];
    f3* v0; f3* v1; f3* v2;    /* normalised positions, triangle order of the index buffer */
    f3* n0; f3* n1; f3* n2;    /* vertex normals (object space; uniform scale keeps directions) */
    f3* lo; f3* hi;            /* canonical padded boxes */
    /* CPU BVH (algo = ORC_ALGO_BVH): balanced median split over Morton order, leaves <= 4 */
    uint32_t* order;           /* permutation: BVH slot -> triangle index */
    orc_node* nodes;
    uint32_t nNodes;
} orc_scene;

static inline float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
static inline float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

static uint32_t expand10(uint32_t v)
{
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

typedef struct { uint64_t key; } mkey;
static int cmp_key(const void* a, const void* b)
{
    const uint64_t x = ((const mkey*)a)->key, y = ((const mkey*)b)->key;
    return x < y ? -1 : x > y;
}

static uint32_t build_rec(orc_scene* s, uint32_t lo, uint32_t hi)
{
    const uint32_t id = s->nNodes++;
    orc_node* n = &s->nodes[id];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (hi - lo <= 4) {
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t k = s->order[i];
            mn[0] = fminf(mn[0], s->lo[k].x); mn[1] = fminf(mn[1], s->lo[k].y); mn[2] = fminf(mn[2], s->lo[k].z);
            mx[0] = fmaxf(mx[0], s->hi[k].x); mx[1] = fmaxf(mx[1], s->hi[k].y); mx[2] = fmaxf(mx[2], s->hi[k].z);
        }
        n->left = -1 - (int32_t)lo;
        n->right = (int32_t)(hi - lo);
    } else {
        const uint32_t mid = lo + (hi - lo) / 2;
        const uint32_t l = build_rec(s, lo, mid);
        const uint32_t r = build_rec(s, mid, hi);
        n = &s->nodes[id];
        n->left = (int32_t)l; n->right = (int32_t)r;
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(s->nodes[l].lo[a], s->nodes[r].lo[a]);
            mx[a] = fmaxf(s->nodes[l].hi[a], s->nodes[r].hi[a]);
        }
    }
    memcpy(n->lo, mn, sizeof(mn));
    memcpy(n->hi, mx, sizeof(mx));
    return id;
}

ORC_API void orc_scene_destroy(orc_scene* s)
{
    if (!s) return;
    free(s->v0); free(s->v1); free(s->v2); free(s->n0); free(s->n1); free(s->n2);
    free(s->lo); free(s->hi); free(s->order); free(s->nodes); free(s);
}

/* vb: V x {pos, nrm}; ib: 3T indices as produced by A1 (already reversed). */
ORC_API orc_scene* orc_scene_create(const float* vb, uint32_t V, const uint32_t* ib, uint32_t T)
{
    if (!T || !V) return NULL;
    for (uint32_t i = 0; i < 3 * T; ++i) if (ib[i] >= V) return NULL;
    orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
    s->T = T; s->V = V;
    float aabb[6];
    orc_aabb(vb, V, aabb);
    orc_bound(aabb, s->bound);
    const float cx = s->bound[0], cy = s->bound[1], cz = s->bound[2], w = s->bound[3];
    if (!(w > 0.0f) || !isfinite(w)) { free(s); return NULL; }
    s->v0 = (f3*)malloc(sizeof(f3) * T); s->v1 = (f3*)malloc(sizeof(f3) * T); s->v2 = (f3*)malloc(sizeof(f3) * T);
    s->n0 = (f3*)malloc(sizeof(f3) * T); s->n1 = (f3*)malloc(sizeof(f3) * T); s->n2 = (f3*)malloc(sizeof(f3) * T);
    s->lo = (f3*)malloc(sizeof(f3) * T); s->hi = (f3*)malloc(sizeof(f3) * T);
    mkey* keys = (mkey*)malloc(sizeof(mkey) * T);
    for (uint32_t k = 0; k < T; ++k) {
        f3* pv[3] = {&s->v0[k], &s->v1[k], &s->v2[k]};
        f3* pn[3] = {&s->n0[k], &s->n1[k], &s->n2[k]};
        for (int c = 0; c < 3; ++c) {
            const float* src = vb + (size_t)ib[3 * k + c] * 6;
            pv[c]->x = (src[0] - cx) / w;
            pv[c]->y = (src[1] - cy) / w;
            pv[c]->z = (src[2] - cz) / w;
            pn[c]->x = src[3]; pn[c]->y = src[4]; pn[c]->z = src[5];
        }
        s->lo[k].x = min3f(pv[0]->x, pv[1]->x, pv[2]->x) - ORC_PAD;
        s->lo[k].y = min3f(pv[0]->y, pv[1]->y, pv[2]->y) - ORC_PAD;
        s->lo[k].z = min3f(pv[0]->z, pv[1]->z, pv[2]->z) - ORC_PAD;
        s->hi[k].x = max3f(pv[0]->x, pv[1]->x, pv[2]->x) + ORC_PAD;
        s->hi[k].y = max3f(pv[0]->y, pv[1]->y, pv[2]->y) + ORC_PAD;
        s->hi[k].z = max3f(pv[0]->z, pv[1]->z, pv[2]->z) + ORC_PAD;
        /* Morton key of the box centre, only used to order the oracle's own BVH */
        float c3[3] = {(s->lo[k].x + s->hi[k].x) * 0.5f, (s->lo[k].y + s->hi[k].y) * 0.5f, (s->lo[k].z + s->hi[k].z) * 0.5f};
        uint32_t q[3];
        for (int a = 0; a < 3; ++a) {
            float u = (c3[a] * 0.5f + 0.5f) * 1024.0f;
            if (!(u > 0.0f)) u = 0.0f;
            if (u > 1023.0f) u = 1023.0f;
            q[a] = (uint32_t)u;
        }
        const uint32_t m = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
        keys[k].key = ((uint64_t)m << 32) | k;
    }
    qsort(keys, T, sizeof(mkey), cmp_key);
    s->order = (uint32_t*)malloc(sizeof(uint32_t) * T);
    for (uint32_t i = 0; i < T; ++i) s->order[i] = (uint32_t)(keys[i].key & 0xffffffffu);
    free(keys);
    s->nodes = (orc_node*)malloc(sizeof(orc_node) * (size_t)(2 * T));
    s->nNodes = 0;
    build_rec(s, 0, T);
    return s;
}

ORC_API void orc_scene_bound(const orc_scene* s, float bound[4]) { memcpy(bound, s->bound, 16); }
ORC_API uint32_t orc_scene_num_tris(const orc_scene* s) { return s->T; }
/* normalised positions + padded box of triangle k (for unit tests of the product's tri prep) */
ORC_API void orc_scene_tri(const orc_scene* s, uint32_t k, float pos9[9], float box6[6])
{
    pos9[0] = s->v0[k].x; pos9[1] = s->v0[k].y; pos9[2] = s->v0[k].z;
    pos9[3] = s->v1[k].x; pos9[4] = s->v1[k].y; pos9[5] = s->v1[k].z;
    pos9[6] = s->v2[k].x; pos9[7] = s->v2[k].y; pos9[8] = s->v2[k].z;
    box6[0] = s->lo[k].x; box6[1] = s->lo[k].y; box6[2] = s->lo[k].z;
    box6[3] = s->hi[k].x; box6[4] = s->hi[k].y; box6[5] = s->hi[k].z;
}

/* ==========================================================================================
 * A4 -- ray generation.  DXRVoxelizer.hlsl:44-53 (generateRay) and :64-67 (index un-flatten).
 * ======================================================================================== */
typedef struct {
    float o[3], d[3];
    float inv[3], nod[3];  /* slab constants: inv = 1/d, nod = -(o*inv) */
    int kx, ky, kz;        /* watertight shear permutation */
    float Sx, Sy, Sz;
} orc_ray;

static void ray_setup_shear(orc_ray* r)
{
    /* Woop/Benthin/Wald 2013: kz = dominant axis of d, winding preserved by swapping kx,ky when
     * d[kz] < 0. */
    int kz = 0;
    if (fabsf(r->d[1]) > fabsf(r->d[kz])) kz = 1;
    if (fabsf(r->d[2]) > fabsf(r->d[kz])) kz = 2;
    int kx = (kz + 1) % 3, ky = (kx + 1) % 3;
    if (r->d[kz] < 0.0f) { int t = kx; kx = ky; ky = t; }
    r->kx = kx; r->ky = ky; r->kz = kz;
    r->Sx = r->d[kx] / r->d[kz];
    r->Sy = r->d[ky] / r->d[kz];
    r->Sz = 1.0f / r->d[kz];
}

static void ray_origin(uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz, float o[3])
{
    /* hlsl:46  pos = (index + 0.5) / DispatchRaysDimensions().x * 2.0 - 1.0 ; :49 pos.y = -pos.y */
    const float fn = (float)N;
    o[0] = ((float)ix + 0.5f) / fn * 2.0f - 1.0f;
    o[1] = -(((float)iy + 0.5f) / fn * 2.0f - 1.0f);
    o[2] = ((float)iz + 0.5f) / fn * 2.0f - 1.0f;
}

/* reference mode: direction = normalize(pos) (hlsl:52), canonical p / sqrtf((xx+yy)+zz) */
static void ray_make_reference(orc_ray* r, uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz)
{
    ray_origin(N, ix, iy, iz, r->o);
    const float len = sqrtf((r->o[0] * r->o[0] + r->o[1] * r->o[1]) + r->o[2] * r->o[2]);
    for (int a = 0; a < 3; ++a) {
        r->d[a] = r->o[a] / len;
        r->inv[a] = 1.0f / r->d[a];
        r->nod[a] = -(r->o[a] * r->inv[a]);
    }
    ray_setup_shear(r);
}

ORC_API void orc_ray_reference(uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz, float o[3], float d[3])
{
    orc_ray r;
    ray_make_reference(&r, N, ix, iy, iz);
    memcpy(o, r.o, 12); memcpy(d, r.d, 12);
}

/* canonical slab test; returns 1 and *tn when the ray interval [max(0,entry), exit] is non-empty */
static inline int slab(const orc_ray* r, const float lo[3], const float hi[3], float* tn_out)
{
    const float t0x = fmaf(lo[0], r->inv[0], r->nod[0]), t1x = fmaf(hi[0], r->inv[0], r->nod[0]);
    const float t0y = fmaf(lo[1], r->inv[1], r->nod[1]), t1y = fmaf(hi[1], r->inv[1], r->nod[1]);
    const float t0z = fmaf(lo[2], r->inv[2], r->nod[2]), t1z = fmaf(hi[2], r->inv[2], r->nod[2]);
    const float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
    const float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    *tn_out = tn;
    return tn <= tf;
}

ORC_API int orc_slab(const float o[3], const float d[3], const float lo[3], const float hi[3], float* tn)
{
    orc_ray r;
    for (int a = 0; a < 3; ++a) {
        r.o[a] = o[a]; r.d[a] = d[a];
        r.inv[a] = 1.0f / d[a];
        r.nod[a] = -(o[a] * r.inv[a]);
    }
    return slab(&r, lo, hi, tn);
}

/* Canonical watertight ray/triangle test (Woop, Benthin, Wald, JCGT 2013), both faces, strict
 * 0 < t < TMax (DXR: TMin < t < TMax for triangles; hlsl:76-77).  Barycentrics follow DXR:
 * b1 = weight of vertex 1, b2 = weight of vertex 2 (hlsl:110-116).
 * fill != 0 (parity mode only): an exactly-zero edge function takes the sign it has at the
 * symbolically perturbed origin o + (eps, eps^2) in the sheared plane, so a ray through a shared
 * edge or vertex is counted for exactly one of the incident triangles. */
static inline int tri_test(const orc_ray* r, const f3* v0, const f3* v1, const f3* v2, int fill,
                           float* t_out, float* b1_out, float* b2_out)
{
    const float a[3] = {v0->x - r->o[0], v0->y - r->o[1], v0->z - r->o[2]};
    const float b[3] = {v1->x - r->o[0], v1->y - r->o[1], v1->z - r->o[2]};
    const float c[3] = {v2->x - r->o[0], v2->y - r->o[1], v2->z - r->o[2]};
    const int kx = r->kx, ky = r->ky, kz = r->kz;
    const float Ax = fmaf(-r->Sx, a[kz], a[kx]), Ay = fmaf(-r->Sy, a[kz], a[ky]);
    const float Bx = fmaf(-r->Sx, b[kz], b[kx]), By = fmaf(-r->Sy, b[kz], b[ky]);
    const float Cx = fmaf(-r->Sx, c[kz], c[kx]), Cy = fmaf(-r->Sy, c[kz], c[ky]);
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    float su = U, sv = V, sw = W; /* values used for the sign test */
    if (fill) {
        /* edge of U runs C->B, of V A->C, of W B->A; perturbed sign = (ey != 0) ? -ey : ex */
        if (su == 0.0f) { const float ey = By - Cy, ex = Bx - Cx; su = ey != 0.0f ? -ey : ex; }
        if (sv == 0.0f) { const float ey = Cy - Ay, ex = Cx - Ax; sv = ey != 0.0f ? -ey : ex; }
        if (sw == 0.0f) { const float ey = Ay - By, ex = Ax - Bx; sw = ey != 0.0f ? -ey : ex; }
        if (su == 0.0f || sv == 0.0f || sw == 0.0f) return 0; /* zero-length projected edge */
    }
    if ((su < 0.0f || sv < 0.0f || sw < 0.0f) && (su > 0.0f || sv > 0.0f || sw > 0.0f)) return 0;
    const float det = (U + V) + W;
    if (det == 0.0f) return 0;
    const float Az = r->Sz * a[kz], Bz = r->Sz * b[kz], Cz = r->Sz * c[kz];
    const float T = (U * Az + V * Bz) + W * Cz;
    const float t = T / det;
    if (!(t > 0.0f && t < ORC_TMAX)) return 0;
    *t_out = t;
    *b1_out = V / det;
    *b2_out = W / det;
    return 1;
}

ORC_API int orc_tri_test(const float o[3], const float d[3], const float v0[3], const float v1[3],
                         const float v2[3], int fill, float* t, float* b1, float* b2)
{
    orc_ray r;
    for (int a = 0; a < 3; ++a) { r.o[a] = o[a]; r.d[a] = d[a]; }
    ray_setup_shear(&r);
    f3 p0 = {v0[0], v0[1], v0[2]}, p1 = {v1[0], v1[1], v1[2]}, p2 = {v2[0], v2[1], v2[2]};
    return tri_test(&r, &p0, &p1, &p2, fill, t, b1, b2);
}

/* ==========================================================================================
 * A5 -- closest-hit predicate.  hlsl:110-116 (attribute interpolation), :137-138.
 * ======================================================================================== */
static inline int predicate(const orc_scene* s, const orc_ray* r, uint32_t k, float b1, float b2, float nrm_out[3])
{
    const f3 n0 = s->n0[k], n1 = s->n1[k], n2 = s->n2[k];
    float nx = (n0.x + b1 * (n1.x - n0.x)) + b2 * (n2.x - n0.x);
    float ny = (n0.y + b1 * (n1.y - n0.y)) + b2 * (n2.y - n0.y);
    float nz = (n0.z + b1 * (n1.z - n0.z)) + b2 * (n2.z - n0.z);
    const float l = sqrtf((nx * nx + ny * ny) + nz * nz);
    nx /= l; ny /= l; nz /= l;
    if (nrm_out) { nrm_out[0] = nx; nrm_out[1] = ny; nrm_out[2] = nz; }
    return ((nx * r->d[0] + ny * r->d[1]) + nz * r->d[2]) > ORC_THRESHOLD;
}

typedef struct { float t, b1, b2; uint32_t k; } orc_hit;

/* canonical per-triangle acceptance in reference mode: own padded box passes the slab test with
 * entry tn <= t; closest = lexicographic min of (t, k). */
static inline void consider_ref(const orc_scene* s, const orc_ray* r, uint32_t k, orc_hit* best)
{
    float tn;
    const float lo[3] = {s->lo[k].x, s->lo[k].y, s->lo[k].z}, hi[3] = {s->hi[k].x, s->hi[k].y, s->hi[k].z};
    if (!slab(r, lo, hi, &tn)) return;
    if (tn > best->t) return;
    float t, b1, b2;
    if (!tri_test(r, &s->v0[k], &s->v1[k], &s->v2[k], 0, &t, &b1, &b2)) return;
    if (tn > t) return;
    if (t < best->t || (t == best->t && k < best->k)) { best->t = t; best->k = k; best->b1 = b1; best->b2 = b2; }
}

static void trace_ref_brute(const orc_scene* s, const orc_ray* r, orc_hit* best)
{
    for (uint32_t k = 0; k < s->T; ++k) consider_ref(s, r, k, best);
}

/* ORC_ALGO_PLAIN: what a tracer WITHOUT the two rules this restatement adds would compute -- every triangle takes the
 * fp32 watertight test, no padded-box candidacy, no tn <= t; closest = min (t, k).  Not a product path and not the
 * canonical rule: tests/test_oracle_anchor.py uses it to COUNT the voxels those two rules change (none on the assets). */
static void trace_ref_plain(const orc_scene* s, const orc_ray* r, orc_hit* best)
{
    for (uint32_t k = 0; k < s->T; ++k) {
        float t, b1, b2;
        if (!tri_test(r, &s->v0[k], &s->v1[k], &s->v2[k], 0, &t, &b1, &b2)) continue;
        if (t < best->t || (t == best->t && k < best->k)) { best->t = t; best->k = k; best->b1 = b1; best->b2 = b2; }
    }
}

static void trace_ref_bvh(const orc_scene* s, const orc_ray* r, orc_hit* best)
{
    uint32_t stack[128];
    int sp = 0;
    float tn;
    if (!slab(r, s->nodes[0].lo, s->nodes[0].hi, &tn)) return;
    uint32_t cur = 0;
    for (;;) {
        const orc_node* n = &s->nodes[cur];
        if (n->left < 0) {
            const uint32_t first = (uint32_t)(-1 - n->left), cnt = (uint32_t)n->right;
            for (uint32_t i = 0; i < cnt; ++i) consider_ref(s, r, s->order[first + i], best);
        } else {
            float tl, tr;
            const int hl = slab(r, s->nodes[n->left].lo, s->nodes[n->left].hi, &tl) && tl <= best->t;
            const int hr = slab(r, s->nodes[n->right].lo, s->nodes[n->right].hi, &tr) && tr <= best->t;
            if (hl && hr) {
                if (tr < tl) { stack[sp++] = (uint32_t)n->left; cur = (uint32_t)n->right; }
                else { stack[sp++] = (uint32_t)n->right; cur = (uint32_t)n->left; }
                continue;
            } else if (hl) { cur = (uint32_t)n->left; continue; }
            else if (hr) { cur = (uint32_t)n->right; continue; }
        }
        if (!sp) break;
        cur = stack[--sp];
    }
}

/* ------------------------------------------------------------------------------------------
 * Parity mode (north_star's axis-aligned hit count; no reference counterpart): ray +X from the
 * voxel centre, all hits with t > 0, occupancy = count & 1.
 * ---------------------------------------------------------------------------------------- */
static void ray_make_parity(orc_ray* r, uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz)
{
    ray_origin(N, ix, iy, iz, r->o);
    r->d[0] = 1.0f; r->d[1] = 0.0f; r->d[2] = 0.0f;
    r->kz = 0; r->kx = 1; r->ky = 2;
    r->Sx = 0.0f; r->Sy = 0.0f; r->Sz = 1.0f;
}

static inline int box_parity(const orc_ray* r, const float lo[3], const float hi[3])
{
    return lo[1] <= r->o[1] && r->o[1] <= hi[1] && lo[2] <= r->o[2] && r->o[2] <= hi[2] && hi[0] >= r->o[0];
}

static inline uint32_t consider_par(const orc_scene* s, const orc_ray* r, uint32_t k)
{
    const float lo[3] = {s->lo[k].x, s->lo[k].y, s->lo[k].z}, hi[3] = {s->hi[k].x, s->hi[k].y, s->hi[k].z};
    if (!box_parity(r, lo, hi)) return 0;
    float t, b1, b2;
    return (uint32_t)tri_test(r, &s->v0[k], &s->v1[k], &s->v2[k], 1, &t, &b1, &b2);
}

static uint32_t count_par_brute(const orc_scene* s, const orc_ray* r)
{
    uint32_t c = 0;
    for (uint32_t k = 0; k < s->T; ++k) c += consider_par(s, r, k);
    return c;
}

static uint32_t count_par_bvh(const orc_scene* s, const orc_ray* r)
{
    uint32_t stack[128];
    int sp = 0;
    uint32_t c = 0;
    if (!box_parity(r, s->nodes[0].lo, s->nodes[0].hi)) return 0;
    uint32_t cur = 0;
    for (;;) {
        const orc_node* n = &s->nodes[cur];
        if (n->left < 0) {
            const uint32_t first = (uint32_t)(-1 - n->left), cnt = (uint32_t)n->right;
            for (uint32_t i = 0; i < cnt; ++i) c += consider_par(s, r, s->order[first + i]);
        } else {
            const int hl = box_parity(r, s->nodes[n->left].lo, s->nodes[n->left].hi);
            const int hr = box_parity(r, s->nodes[n->right].lo, s->nodes[n->right].hi);
            if (hl && hr) { stack[sp++] = (uint32_t)n->right; cur = (uint32_t)n->left; continue; }
            else if (hl) { cur = (uint32_t)n->left; continue; }
            else if (hr) { cur = (uint32_t)n->right; continue; }
        }
        if (!sp) break;
        cur = stack[--sp];
    }
    return c;
}

/* D3D UNORM conversion of the value the shader writes, float4(Normal, 1) into R10G10B10A2_UNORM
 * (hlsl:84, Content/Voxelizer.cpp:65): clamp to [0,1], scale, round to nearest. */
static inline uint32_t unorm(float v, float scale)
{
    if (!(v > 0.0f)) v = 0.0f; /* NaN -> 0 */
    if (v > 1.0f) v = 1.0f;
    return (uint32_t)(v * scale + 0.5f);
}

/* One voxel, reference mode.  Returns occupancy; optional outputs for unit tests. */
ORC_API int orc_voxel_reference(const orc_scene* s, uint32_t N, uint32_t ix, uint32_t iy, uint32_t iz, int algo,
                                float* t_out, uint32_t* k_out, float* b_out /*2*/, uint32_t* texel_out)
{
    orc_ray r;
    ray_make_reference(&r, N, ix, iy, iz);
    orc_hit best = {ORC_TMAX, 0.0f, 0.0f, UINT32_MAX};
    if (algo == ORC_ALGO_BRUTE) trace_ref_brute(s, &r, &best);
    else if (algo == ORC_ALGO_PLAIN) trace_ref_plain(s, &r, &best);
    else trace_ref_bvh(s, &r, &best);
    if (t_out) *t_out = best.t;
    if (k_out) *k_out = best.k;
    if (b_out) { b_out[0] = best.b1; b_out[1] = best.b2; }
    if (texel_out) *texel_out = 0;
    if (best.k == UINT32_MAX) return 0; /* missMain: hlsl:145-148 */
    float n[3];
    const int in = predicate(s, &r, best.k, best.b1, best.b2, n);
    if (in && texel_out)
        *texel_out = unorm(n[0], 1023.0f) | (unorm(n[1], 1023.0f) << 10) | (unorm(n[2], 1023.0f) << 20) | (3u << 30);
    return in;
}

/* Voxelize slices [z0, z0+nz) of an N^3 grid into out (N*N*nz bytes, x fastest, then y, then z:
 * id = (iz*N + iy)*N + ix, hlsl:64-67).  texels (optional, reference mode): the
 * R10G10B10A2_UNORM value the reference would have written, 0 where it writes nothing.
 * threads <= 0: all OpenMP threads. Returns 0 on success. */
ORC_API int orc_voxelize(const orc_scene* s, uint32_t N, int mode, int algo, uint32_t z0, uint32_t nz,
                         int threads, uint8_t* out, uint32_t* texels)
{
    if (!s || !out || N < 2 || (N & 1u) || z0 + nz > N) return 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads); else omp_set_num_threads(omp_get_num_procs());
#else
    (void)threads;
#endif
    const int64_t rows = (int64_t)nz * N;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t row = 0; row < rows; ++row) {
        const uint32_t iz = z0 + (uint32_t)(row / N), iy = (uint32_t)(row % N);
        uint8_t* dst = out + (size_t)row * N;
        uint32_t* tex = texels ? texels + (size_t)row * N : NULL;
        for (uint32_t ix = 0; ix < N; ++ix) {
            if (mode == ORC_MODE_REFERENCE) {
                uint32_t tx = 0;
                dst[ix] = (uint8_t)orc_voxel_reference(s, N, ix, iy, iz, algo, NULL, NULL, NULL, tex ? &tx : NULL);
                if (tex) tex[ix] = tx;
            } else {
                orc_ray r;
                ray_make_parity(&r, N, ix, iy, iz);
                const uint32_t c = algo == ORC_ALGO_BRUTE ? count_par_brute(s, &r) : count_par_bvh(s, &r);
                dst[ix] = (uint8_t)(c & 1u);
            }
        }
    }
    return 0;
}

/* Same as orc_voxelize for an arbitrary list of slices (one parallel region over all their rows:
 * the cpu_baseline sample of bench.py). out = nlist * N * N bytes, slice j of the list first. */
ORC_API int orc_voxelize_slices(const orc_scene* s, uint32_t N, int mode, int algo, const uint32_t* zlist,
                                uint32_t nlist, int threads, uint8_t* out)
{
    if (!s || !out || !zlist || N < 2 || (N & 1u)) return 1;
    for (uint32_t j = 0; j < nlist; ++j) if (zlist[j] >= N) return 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads); else omp_set_num_threads(omp_get_num_procs());
#else
    (void)threads;
#endif
    const int64_t rows = (int64_t)nlist * N;
#pragma omp parallel for schedule(dynamic, 2)
    for (int64_t row = 0; row < rows; ++row) {
        const uint32_t iz = zlist[row / N], iy = (uint32_t)(row % N);
        uint8_t* dst = out + (size_t)row * N;
        for (uint32_t ix = 0; ix < N; ++ix) {
            if (mode == ORC_MODE_REFERENCE) {
                dst[ix] = (uint8_t)orc_voxel_reference(s, N, ix, iy, iz, algo, NULL, NULL, NULL, NULL);
            } else {
                orc_ray r;
                ray_make_parity(&r, N, ix, iy, iz);
                const uint32_t c = algo == ORC_ALGO_BRUTE ? count_par_brute(s, &r) : count_par_bvh(s, &r);
                dst[ix] = (uint8_t)(c & 1u);
            }
        }
    }
    return 0;
}

/* ==========================================================================================
 * N3 -- the grid's consumer (display pass).  Restates Content/Shaders/PSRayCast.hlsl:61-187 with
 * the constants Voxelizer::UpdateFrame builds (Content/Voxelizer.cpp:81-106).  float32 in place
 * of min16float, float trilinear CLAMP sampling of alpha.  Matrices: row-major, row vectors
 * (v' = v * M), the DirectXMath convention of the reference.
 * ======================================================================================== */
typedef struct { float light[3], eye[3], s2l[16]; } orc_cb;

static void m4_mul(const float* a, const float* b, float* o)
{
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c)
        o[4 * r + c] = ((a[4 * r] * b[c] + a[4 * r + 1] * b[4 + c]) + a[4 * r + 2] * b[8 + c]) + a[4 * r + 3] * b[12 + c];
}

static int m4_inv(const float* m, float* o)   /* Gauss-Jordan in double */
{
    double a[4][8];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = m[4 * r + c]; a[r][4 + c] = r == c; }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r) if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0) return 0;
        for (int c = 0; c < 8; ++c) { double t = a[col][c]; a[col][c] = a[piv][c]; a[piv][c] = t; }
        const double d = a[col][col];
        for (int c = 0; c < 8; ++c) a[col][c] /= d;
        for (int r = 0; r < 4; ++r) if (r != col) { const double f = a[r][col]; for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c]; }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) o[4 * r + c] = (float)a[r][4 + c];
    return 1;
}

static void m4_coord(const float* p, const float* m, float* o)
{
    const float x = ((p[0] * m[0] + p[1] * m[4]) + p[2] * m[8]) + m[12], y = ((p[0] * m[1] + p[1] * m[5]) + p[2] * m[9]) + m[13];
    const float z = ((p[0] * m[2] + p[1] * m[6]) + p[2] * m[10]) + m[14], w = ((p[0] * m[3] + p[1] * m[7]) + p[2] * m[11]) + m[15];
    o[0] = x / w; o[1] = y / w; o[2] = z / w;
}

/* Voxelizer::UpdateFrame, Content/Voxelizer.cpp:81-106 */
ORC_API int orc_update_frame(const float bound[4], const float ps[4], const float eye[3], const float vp[16], float w, float h,
                             float light_out[3], float eye_out[3], float s2l_out[16])
{
    float S1[16] = {0}, T1[16] = {0}, S2[16] = {0}, T2[16] = {0}, a[16], b[16], world[16], worldI[16], wvp[16], ts[16] = {0}, l2s[16];
    S1[0] = S1[5] = S1[10] = bound[3]; S1[15] = 1;
    T1[0] = T1[5] = T1[10] = T1[15] = 1; T1[12] = bound[0]; T1[13] = bound[1]; T1[14] = bound[2];
    S2[0] = S2[5] = S2[10] = ps[3]; S2[15] = 1;
    T2[0] = T2[5] = T2[10] = T2[15] = 1; T2[12] = ps[0]; T2[13] = ps[1]; T2[14] = ps[2];
    m4_mul(S1, T1, a); m4_mul(a, S2, b); m4_mul(b, T2, world);
    if (!m4_inv(world, worldI)) return 1;
    m4_mul(world, vp, wvp);
    const float lp[3] = {-10.0f, 45.0f, -75.0f};
    m4_coord(lp, worldI, light_out);
    m4_coord(eye, worldI, eye_out);
    ts[0] = 0.5f * w; ts[5] = -0.5f * h; ts[10] = 1.0f; ts[12] = 0.5f * w; ts[13] = 0.5f * h; ts[15] = 1.0f;
    m4_mul(wvp, ts, l2s);
    return m4_inv(l2s, s2l_out) ? 0 : 1;
}

static float tex_alpha(const uint8_t* g, uint32_t N, float tx, float ty, float tz)
{
    const float fn = (float)N;
    const float u[3] = {tx * fn - 0.5f, ty * fn - 0.5f, tz * fn - 0.5f};
    int i0[3], i1[3];
    float w[3];
    for (int a = 0; a < 3; ++a) {
        const float f = floorf(u[a]);
        w[a] = u[a] - f;
        int lo = (int)f, hi = (int)(f + 1.0f);
        if (lo < 0) lo = 0;
        if (lo > (int)N - 1) lo = (int)N - 1;
        if (hi < 0) hi = 0;
        if (hi > (int)N - 1) hi = (int)N - 1;
        i0[a] = lo; i1[a] = hi;
    }
#define G(x, y, z) (g[((size_t)(z) * N + (y)) * N + (x)] ? 1.0f : 0.0f)
    const float c00 = G(i0[0], i0[1], i0[2]) + w[0] * (G(i1[0], i0[1], i0[2]) - G(i0[0], i0[1], i0[2]));
    const float c10 = G(i0[0], i1[1], i0[2]) + w[0] * (G(i1[0], i1[1], i0[2]) - G(i0[0], i1[1], i0[2]));
    const float c01 = G(i0[0], i0[1], i1[2]) + w[0] * (G(i1[0], i0[1], i1[2]) - G(i0[0], i0[1], i1[2]));
    const float c11 = G(i0[0], i1[1], i1[2]) + w[0] * (G(i1[0], i1[1], i1[2]) - G(i0[0], i1[1], i1[2]));
#undef G
    const float c0 = c00 + w[1] * (c10 - c00), c1 = c01 + w[1] * (c11 - c01);
    return c0 + w[2] * (c1 - c0);
}

static float density_at(const uint8_t* g, uint32_t N, const float* p)   /* PSRayCast.hlsl:104-113, :137 */
{
    const float d = tex_alpha(g, N, 0.5f * p[0] + 0.5f, -0.5f * p[1] + 0.5f, 0.5f * p[2] + 0.5f);
    return fminf(d * 8.0f, 16.0f);
}

static void shade_pixel(const orc_cb* cb, const uint8_t* g, uint32_t N, float sx, float sy, float* rgba)
{
    static const float clear[3] = {0.0f, 0.2f, 0.4f};
    const float maxDist = 2.0f * sqrtf(3.0f), stepScale = maxDist / 128.0f, lightScale = maxDist / 32.0f;
    const float* m = cb->s2l;
    const float hx = (sx * m[0] + sy * m[4]) + m[12], hy = (sx * m[1] + sy * m[5]) + m[13];
    const float hz = (sx * m[2] + sy * m[6]) + m[14], hw = (sx * m[3] + sy * m[7]) + m[15];
    float pos[3] = {hx / hw, hy / hw, hz / hw}, dir[3];
    for (int a = 0; a < 3; ++a) dir[a] = pos[a] - cb->eye[a];
    const float dl = sqrtf((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
    for (int a = 0; a < 3; ++a) dir[a] /= dl;
    if (!(fabsf(pos[0]) <= 1.0f && fabsf(pos[1]) <= 1.0f && fabsf(pos[2]) <= 1.0f)) {      /* ComputeStartPoint */
        float U = 3.402823466e+38f;
        int hit = 0;
        for (int i = 0; i < 3; ++i) {
            const float sg = dir[i] > 0.0f ? 1.0f : (dir[i] < 0.0f ? -1.0f : 0.0f);
            const float u = (-sg - pos[i]) / dir[i];
            if (u < 0.0f) continue;
            const int j = (i + 1) % 3, k = (i + 2) % 3;
            if (fabsf(dir[j] * u + pos[j]) > 1.0f) continue;
            if (fabsf(dir[k] * u + pos[k]) > 1.0f) continue;
            if (u < U) { U = u; hit = 1; }
        }
        for (int a = 0; a < 3; ++a) pos[a] = fminf(fmaxf(dir[a] * U + pos[a], -1.0f), 1.0f);
        if (!hit) { rgba[0] = clear[0]; rgba[1] = clear[1]; rgba[2] = clear[2]; rgba[3] = 0.0f; return; }
    }
    float step[3], ls[3];
    const float ll = sqrtf((cb->light[0] * cb->light[0] + cb->light[1] * cb->light[1]) + cb->light[2] * cb->light[2]);
    for (int a = 0; a < 3; ++a) { step[a] = dir[a] * stepScale; ls[a] = cb->light[a] / ll * lightScale; }
    float transmit = 1.0f, scatter = 0.0f;
    for (int i = 0; i < 128; ++i) {
        if (fabsf(pos[0]) > 1.0f || fabsf(pos[1]) > 1.0f || fabsf(pos[2]) > 1.0f) break;
        const float dens = density_at(g, N, pos);
        if (dens > 0.01f) {
            const float sd = dens * stepScale;
            transmit *= fminf(fmaxf(1.0f - sd * 1.0f, 0.0f), 1.0f);
            if (transmit < 0.01f) break;
            float lt = 1.0f, lp[3] = {pos[0] + ls[0], pos[1] + ls[1], pos[2] + ls[2]};
            for (int j = 0; j < 32; ++j) {
                if (fabsf(lp[0]) > 1.0f || fabsf(lp[1]) > 1.0f || fabsf(lp[2]) > 1.0f) break;
                const float ld = density_at(g, N, lp);
                lt *= fminf(fmaxf(1.0f - 1.0f * lightScale * ld, 0.0f), 1.0f);
                if (lt < 0.01f) break;
                for (int a = 0; a < 3; ++a) lp[a] += ls[a];
            }
            scatter += lt * transmit * sd;
        }
        for (int a = 0; a < 3; ++a) pos[a] += step[a];
    }
    for (int c = 0; c < 3; ++c) {
        float r = scatter * 0.8f + 0.2f;
        r = r + transmit * (clear[c] * clear[c] - r);
        rgba[c] = sqrtf(r);
    }
    rgba[3] = 1.0f;
}

/* cb_in = {light[3], eye[3], s2l[16]} (22 floats) or NULL to derive it with orc_update_frame. */
ORC_API int orc_render(const uint8_t* grid, uint32_t N, const float bound[4], const float ps[4], const float eye[3],
                       const float vp[16], uint32_t width, uint32_t height, const float* cb_in, uint8_t* rgba8)
{
    orc_cb cb;
    if (cb_in) { memcpy(cb.light, cb_in, 12); memcpy(cb.eye, cb_in + 3, 12); memcpy(cb.s2l, cb_in + 6, 64); }
    else if (orc_update_frame(bound, ps, eye, vp, (float)width, (float)height, cb.light, cb.eye, cb.s2l)) return 1;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t py = 0; py < (int64_t)height; ++py)
        for (uint32_t px = 0; px < width; ++px) {
            float c[4];
            shade_pixel(&cb, grid, N, (float)px + 0.5f, (float)py + 0.5f, c);
            for (int k = 0; k < 4; ++k) {
                float v = c[k];
                if (!(v > 0.0f)) v = 0.0f;
                if (v > 1.0f) v = 1.0f;
                rgba8[((size_t)py * width + px) * 4 + k] = (uint8_t)(v * 255.0f + 0.5f);
            }
        }
    return 0;
}

ORC_API int orc_num_procs(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}
