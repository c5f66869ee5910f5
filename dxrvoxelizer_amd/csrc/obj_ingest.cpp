// obj_ingest.cpp -- mesh ingest for the voxelizer: dxv_obj_load / dxv_free (include/dxv.h).
//
// Produces exactly what XUSG::ObjLoader::Import(file, needNorm=true, needAABB=true, forDX=true,
// swapYZ=false) hands to Voxelizer::Init (XUSG/Optional/XUSGObjLoader.cpp:18-40,
// Content/Voxelizer.cpp:46-57): an interleaved {pos, nrm} vertex buffer with z negated, an index
// buffer whose whole array is reversed, per-vertex normals (split per distinct vn, or recomputed
// from unweighted face normals when the file has none) and the AABB.  Own single-pass parser
// over the file image; numbers go through strtof/strtoll so values round as the reference's
// fscanf does.  Texture coordinates are parsed and dropped (the reference never stores them).
#include "../../include/dxv.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct Corner { long long v, vn; bool hasVn; };

inline const char* skip_blanks(const char* p) { while (*p == ' ' || *p == '\t' || *p == '\r') ++p; return p; }

// Parses "v", "v/vt", "v//vn", "v/vt/vn".  Returns false at end of line / non-numeric token.
bool next_corner(const char*& p, Corner& c)
{
    p = skip_blanks(p);
    char* e;
    c.v = strtoll(p, &e, 10);
    if (e == p) return false;
    p = e;
    c.vn = 0;
    c.hasVn = false;
    if (*p == '/') {
        ++p;
        if (*p != '/') { (void)strtoll(p, &e, 10); p = e; }
        if (*p == '/') {
            ++p;
            c.vn = strtoll(p, &e, 10);
            c.hasVn = e != p;
            p = e;
        }
    }
    return true;
}

struct Mesh {
    std::vector<float> pos;       // file order, z negated
    std::vector<float> vn;        // file order, z negated
    std::vector<Corner> corners;  // 3 per triangle after fan triangulation, raw OBJ indices
};

} // namespace

extern "C" {

void dxv_free(void* p) { free(p); }

int dxv_obj_load(const char* path, float** vbOut, uint32_t* numVerts, uint32_t** ibOut, uint32_t* numIndices,
                 float aabb[6])
{
    if (!path || !vbOut || !numVerts || !ibOut || !numIndices) return 1;
    *vbOut = nullptr; *ibOut = nullptr; *numVerts = 0; *numIndices = 0;
    FILE* f = fopen(path, "rb");
    if (!f) return 1;
    std::vector<char> text;
    {
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz < 0) { fclose(f); return 1; }
        text.resize((size_t)sz + 2);
        if (fread(text.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return 1; }
        fclose(f);
        text[(size_t)sz] = '\n';
        text[(size_t)sz + 1] = 0;
    }

    Mesh m;
    for (char* line = text.data(); *line;) {
        char* eol = line;
        while (*eol != '\n') ++eol;
        *eol = 0;
        const char* p = skip_blanks(line);
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char* q = const_cast<char*>(p + 1);
            const float x = strtof(q, &q), y = strtof(q, &q), z = strtof(q, &q);
            m.pos.push_back(x); m.pos.push_back(y); m.pos.push_back(-z);        // XUSGObjLoader.cpp:198
        } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
            char* q = const_cast<char*>(p + 2);
            const float x = strtof(q, &q), y = strtof(q, &q), z = strtof(q, &q);
            m.vn.push_back(x); m.vn.push_back(y); m.vn.push_back(-z);          // :213
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            const char* q = p + 1;
            Corner first, prev, cur;
            if (next_corner(q, first) && next_corner(q, prev)) {
                while (next_corner(q, cur)) {                                     // fan: :263-297
                    m.corners.push_back(first); m.corners.push_back(prev); m.corners.push_back(cur);
                    prev = cur;
                }
            }
        }
        line = eol + 1;
    }

    const uint32_t V0 = (uint32_t)(m.pos.size() / 3), NN = (uint32_t)(m.vn.size() / 3);
    const size_t nIdx = m.corners.size();
    if (!V0 || !nIdx || nIdx > 0xfffffff0u) return 2;

    std::vector<float> vb((size_t)V0 * 6, 0.0f);
    for (uint32_t i = 0; i < V0; ++i) memcpy(&vb[(size_t)i * 6], &m.pos[(size_t)i * 3], 12);
    std::vector<uint32_t> ib(nIdx);
    for (size_t i = 0; i < nIdx; ++i) {
        const long long v = m.corners[i].v;
        const long long r = v < 0 ? v + (long long)V0 : v - 1;                    // :243
        if (r < 0 || r >= (long long)V0) return 3;
        ib[i] = (uint32_t)r;
    }

    if (NN) {
        // one normal per vertex; a vertex met again with another vn is duplicated (:300-335)
        std::vector<uint32_t> owner(V0, UINT32_MAX);
        for (size_t i = 0; i < nIdx; ++i) {
            const long long n = m.corners[i].vn;
            const long long rn = m.corners[i].hasVn ? (n < 0 ? n + (long long)NN : n - 1) : 0;
            if (rn < 0 || rn >= (long long)NN) return 3;
            const uint32_t ni = (uint32_t)rn;
            uint32_t vi = ib[i];
            if (owner[vi] == ni) continue;
            if (owner[vi] != UINT32_MAX) {
                const uint32_t nv = (uint32_t)(vb.size() / 6);
                vb.resize(vb.size() + 6);
                memcpy(&vb[(size_t)nv * 6], &vb[(size_t)vi * 6], 24);
                ib[i] = vi = nv;
            } else owner[vi] = ni;
            const float x = m.vn[(size_t)ni * 3], y = m.vn[(size_t)ni * 3 + 1], z = m.vn[(size_t)ni * 3 + 2];
            const float l = sqrtf(x * x + y * y + z * z);
            float* d = &vb[(size_t)vi * 6 + 3];
            d[0] = x / l; d[1] = y / l; d[2] = z / l;
        }
    }

    // forDX: the WHOLE index array is reversed (:227): winding and triangle order flip
    for (size_t a = 0, b = nIdx - 1; a < b; ++a, --b) { const uint32_t t = ib[a]; ib[a] = ib[b]; ib[b] = t; }

    const uint32_t V = (uint32_t)(vb.size() / 6);
    if (!NN) {
        // recomputeNormals (:337-384): n = normalize(cross(v1-v0, v2-v1)) added unweighted
        for (size_t t = 0; t + 2 < nIdx; t += 3) {
            const float* a = &vb[(size_t)ib[t] * 6];
            const float* b = &vb[(size_t)ib[t + 1] * 6];
            const float* c = &vb[(size_t)ib[t + 2] * 6];
            const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
            const float e2[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
            float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            const float l = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            n[0] /= l; n[1] /= l; n[2] /= l;
            for (int k = 0; k < 3; ++k) {
                float* d = &vb[(size_t)ib[t + k] * 6 + 3];
                d[0] += n[0]; d[1] += n[1]; d[2] += n[2];
            }
        }
        for (uint32_t i = 0; i < V; ++i) {
            float* d = &vb[(size_t)i * 6 + 3];
            const float l = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] /= l; d[1] /= l; d[2] /= l;
        }
    }

    if (aabb) {                                                                   // :386-416
        for (int a = 0; a < 3; ++a) aabb[a] = aabb[3 + a] = vb[a];
        for (uint32_t i = 1; i < V; ++i)
            for (int a = 0; a < 3; ++a) {
                const float x = vb[(size_t)i * 6 + a];
                if (x < aabb[a]) aabb[a] = x;
                else if (x > aabb[3 + a]) aabb[3 + a] = x;
            }
    }

    float* ovb = static_cast<float*>(malloc(vb.size() * sizeof(float)));
    uint32_t* oib = static_cast<uint32_t*>(malloc(ib.size() * sizeof(uint32_t)));
    if (!ovb || !oib) { free(ovb); free(oib); return 4; }
    memcpy(ovb, vb.data(), vb.size() * sizeof(float));
    memcpy(oib, ib.data(), ib.size() * sizeof(uint32_t));
    *vbOut = ovb; *numVerts = V; *ibOut = oib; *numIndices = (uint32_t)nIdx;
    return 0;
}

} // extern "C"
