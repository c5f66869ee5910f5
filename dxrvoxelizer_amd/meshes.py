"""Deterministic synthetic meshes for tests and bench.py (no RNG state, no files): the inputs
BASELINE.md section 4 names.  Every generator returns (vb [V,6] float32 {pos, nrm}, ib [3T] uint32)
in the layout the reference's ObjLoader hands to Voxelizer::Init."""
import numpy as np


def _vb(pos, nrm):
    return np.ascontiguousarray(np.concatenate([pos, nrm], axis=1), dtype=np.float32)


def face_normals_like_reference(pos, ib):
    """Per-vertex normals by the reference's recomputeNormals rule
    (XUSG/Optional/XUSGObjLoader.cpp:337-384) for meshes whose vertices are private to one
    triangle: the normalised face normal cross(v1-v0, v2-v1), normalised once more."""
    p = pos.astype(np.float32)
    t = ib.reshape(-1, 3)
    e1 = p[t[:, 1]] - p[t[:, 0]]
    e2 = p[t[:, 2]] - p[t[:, 1]]
    n = np.cross(e1, e2).astype(np.float32)
    n /= np.sqrt((n * n).sum(1, dtype=np.float32))[:, None].astype(np.float32)
    out = np.zeros_like(p)
    for c in range(3):
        out[t[:, c]] = n
    out /= np.sqrt((out * out).sum(1, dtype=np.float32))[:, None]
    return out.astype(np.float32)


def torus(nu=1000, nv=500, R=0.6, r=0.3):
    """Closed torus around the Y axis, nu x nv quads -> 2*nu*nv triangles (1,000,000 by default),
    analytic outward vertex normals."""
    u = (np.arange(nu, dtype=np.float64) / nu) * 2 * np.pi
    v = (np.arange(nv, dtype=np.float64) / nv) * 2 * np.pi
    uu, vv = np.meshgrid(u, v, indexing="ij")
    cx, cz = np.cos(uu), np.sin(uu)
    pos = np.stack([(R + r * np.cos(vv)) * cx, r * np.sin(vv), (R + r * np.cos(vv)) * cz], -1).reshape(-1, 3)
    nrm = np.stack([np.cos(vv) * cx, np.sin(vv), np.cos(vv) * cz], -1).reshape(-1, 3)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    a = (i * nv + j).ravel()
    b = (((i + 1) % nu) * nv + j).ravel()
    c = (((i + 1) % nu) * nv + (j + 1) % nv).ravel()
    d = (i * nv + (j + 1) % nv).ravel()
    # outward-facing counter-clockwise as seen from outside
    tris = np.concatenate([np.stack([a, d, c], 1), np.stack([a, c, b], 1)], 0)
    return _vb(pos, nrm), np.ascontiguousarray(tris.reshape(-1), dtype=np.uint32)


def uv_sphere(nlon=64, nlat=32, radius=0.8, center=(0.0, 0.0, 0.0)):
    lon = (np.arange(nlon, dtype=np.float64) / nlon) * 2 * np.pi
    lat = (np.arange(1, nlat, dtype=np.float64) / nlat) * np.pi
    ll, tt = np.meshgrid(lon, lat, indexing="ij")
    n = np.stack([np.sin(tt) * np.cos(ll), np.cos(tt), np.sin(tt) * np.sin(ll)], -1).reshape(-1, 3)
    n = np.concatenate([n, [[0, 1, 0]], [[0, -1, 0]]], 0)
    pos = n * radius + np.asarray(center)
    top, bot = nlon * (nlat - 1), nlon * (nlat - 1) + 1
    tris = []
    for i in range(nlon):
        i1 = (i + 1) % nlon
        tris.append([top, i1 * (nlat - 1), i * (nlat - 1)])
        tris.append([bot, i * (nlat - 1) + nlat - 2, i1 * (nlat - 1) + nlat - 2])
        for j in range(nlat - 2):
            a, b = i * (nlat - 1) + j, i1 * (nlat - 1) + j
            tris.append([a, b, b + 1])
            tris.append([a, b + 1, a + 1])
    return _vb(pos, n), np.asarray(tris, np.uint32).reshape(-1)


def cube(half=0.7):
    """Axis-aligned cube, 12 triangles, private vertices per face, face normals."""
    p, n, t = [], [], []
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            u, v = (axis + 1) % 3, (axis + 2) % 3
            base = len(p)
            for su, sv in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                q = [0.0, 0.0, 0.0]
                q[axis], q[u], q[v] = sgn * half, su * half, sv * half
                p.append(q)
                nn = [0.0, 0.0, 0.0]
                nn[axis] = sgn
                n.append(nn)
            quad = [0, 1, 2, 0, 2, 3] if sgn > 0 else [0, 2, 1, 0, 3, 2]
            t += [base + k for k in quad]
    return _vb(np.asarray(p), np.asarray(n)), np.asarray(t, np.uint32)


def tetrahedron(s=0.8):
    p = np.asarray([[s, s, s], [s, -s, -s], [-s, s, -s], [-s, -s, s]], np.float64)
    t = np.asarray([0, 1, 2, 0, 3, 1, 0, 2, 3, 1, 3, 2], np.uint32)
    pos = p[t]                       # private vertices, flat normals
    ib = np.arange(12, dtype=np.uint32)
    e1, e2 = pos[1::3] - pos[0::3], pos[2::3] - pos[1::3]
    fn = np.cross(e1, e2)
    fn /= np.linalg.norm(fn, axis=1)[:, None]
    return _vb(pos, np.repeat(fn, 3, 0)), ib


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _uniform(seed, n, stream):
    """n float64 in [0,1) from a counter-based splitmix64 hash: reproducible across numpy versions."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(16) + np.uint64(stream)
        bits = _splitmix64(idx + np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def soup(num_tris=10_000_000, seed=0x5EED1234, extent=0.95, edge=0.02):
    """Triangle soup: centre ~ U[-extent, extent]^3, two edge vectors
    ~ U[-edge, edge]^3, three private vertices per triangle, face normals by the reference's rule."""
    c = np.stack([_uniform(seed, num_tris, s) for s in (0, 1, 2)], 1) * 2 * extent - extent
    e1 = np.stack([_uniform(seed, num_tris, s) for s in (3, 4, 5)], 1) * 2 * edge - edge
    e2 = np.stack([_uniform(seed, num_tris, s) for s in (6, 7, 8)], 1) * 2 * edge - edge
    pos = np.empty((num_tris, 3, 3), np.float32)
    pos[:, 0] = c
    pos[:, 1] = c + e1
    pos[:, 2] = c + e2
    # private vertices: the reference's recomputeNormals rule reduces to the face normal
    # normalize(cross(v1-v0, v2-v1)), normalised once more (float32 arithmetic throughout)
    a, b = pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 1]
    n = np.empty((num_tris, 3), np.float32)
    n[:, 0] = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]
    n[:, 1] = a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2]
    n[:, 2] = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    for _ in range(2):
        n /= np.sqrt(n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1] + n[:, 2] * n[:, 2])[:, None]
    vb = np.empty((num_tris, 3, 6), np.float32)
    vb[:, :, :3] = pos
    vb[:, :, 3:] = n[:, None, :]
    return vb.reshape(-1, 6), np.arange(3 * num_tris, dtype=np.uint32)


def trisect(vb, ib):
    """Split every triangle 3 x 3 by edge trisection (x9 triangles). New vertices carry linearly
    interpolated, NOT re-normalised attributes, so surface and interpolated normal field are
    mathematically those of the input."""
    vb = np.asarray(vb, np.float64).reshape(-1, 6)
    t = np.asarray(ib).reshape(-1, 3)
    a, b, c = vb[t[:, 0]], vb[t[:, 1]], vb[t[:, 2]]
    pts = {}
    for i in range(4):
        for j in range(4 - i):
            k = 3 - i - j
            pts[(i, j)] = (a * k + b * i + c * j) / 3.0
    tri_list = []
    for i in range(3):
        for j in range(3 - i):
            tri_list.append(((i, j), (i + 1, j), (i, j + 1)))
            if i + j < 2:
                tri_list.append(((i + 1, j), (i + 1, j + 1), (i, j + 1)))
    out = np.empty((len(t), len(tri_list), 3, 6), np.float32)
    for n, (p, q, r) in enumerate(tri_list):
        out[:, n, 0], out[:, n, 1], out[:, n, 2] = pts[p], pts[q], pts[r]
    out = out.reshape(-1, 6)
    return np.ascontiguousarray(out), np.arange(len(out), dtype=np.uint32)


def midpoint_subdivide(vb, ib, rounds=1):
    """1 -> 4 midpoint subdivision (x4 triangles per round), interpolated attributes."""
    vb = np.asarray(vb, np.float32).reshape(-1, 6)
    ib = np.asarray(ib, np.uint32).reshape(-1)
    for _ in range(rounds):
        t = ib.reshape(-1, 3)
        a, b, c = vb[t[:, 0]].astype(np.float64), vb[t[:, 1]].astype(np.float64), vb[t[:, 2]].astype(np.float64)
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        out = np.empty((len(t), 4, 3, 6), np.float32)
        for n, (p, q, r) in enumerate(((a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca))):
            out[:, n, 0], out[:, n, 1], out[:, n, 2] = p, q, r
        vb = out.reshape(-1, 6)
        ib = np.arange(len(vb), dtype=np.uint32)
    return np.ascontiguousarray(vb), ib
