"""The oracle against what the reference's SHIPPED BINARIES say about the hot path.

The reference holds no test vector for the path, but it ships the path compiled: Bin/DXRVoxelizer.cso (DXIL of
Content/Shaders/DXRVoxelizer.hlsl), Bin/PSRayCast.cso, Bin/DXRVoxelizer.exe (contains XUSG::ObjLoader).  oracle/shipped_binaries.py
disassembled them in the build container and committed the FACTS (tests/golden/shipped_binaries.json -- data, not the
disassembly).  Here the CPU oracle -- the restatement every GPU parity test is measured against -- is checked against them by
behaviour: which origin component is negated, how the direction is normalised, the interpolation order, the comparison and the
threshold's bits, the empty miss shader, the loader's z negation.  What the binaries leave open is stated too: the compiled
normalize is `x * rsqrt(dot3(x, x))` under `fast` flags, whose last bits DXIL leaves to the driver -- the oracle's
`x / sqrtf(...)` must lie within the tolerance any conforming rsqrt gives, and it does."""
import json
import os
import struct

import numpy as np
import pytest

from conftest import ROOT

FACTS = json.load(open(os.path.join(ROOT, "tests", "golden", "shipped_binaries.json")))
VOX, EXE = FACTS["voxelizer_dxil"], FACTS["exe"]


def test_the_facts_the_restatement_relies_on():
    rg, ch = VOX["raygen"], VOX["closest_hit"]
    assert rg["origin_steps"] == ["+0.5", "/dim", "*2", "-1"] and rg["origin_negated_components"] == ["y"]      # hlsl:46, :49 -- no x mirror in the shipped shader
    assert rg["trace_ray"] == {"flags": 0, "instance_mask": 255, "hit_group_offset": 0, "geometry_stride": 1, "miss_index": 0, "tmin": 0.0, "tmax": 10000.0}
    assert rg["direction"] == "origin * rsqrt(dot3(origin, origin))"
    assert rg["stores"]["only_when"] == "payload.isInside != 0" and rg["stores"]["value"] == "float4(payload.normal, 1.0)"
    assert ch["vertex_fetch"] == {"loads": 3, "byte_offset_in_vertex": [12], "component_mask": [7]} and ch["vertex_stride_bytes"] == 24   # only the normals are fetched
    assert ch["interpolation"] == "(n0 + (n1 - n0) * b.x) + (n2 - n0) * b.y"
    assert ch["predicate"]["compare"] == "ogt" and ch["predicate"]["threshold_f32_bits"] == "0x3df5c28f"
    assert VOX["miss"]["instructions"] == ["ret void"]
    assert FACTS["display_dxil"]["texcoord"] == "(0.5, -0.5, 0.5) * pos + 0.5" and FACTS["display_dxil"]["loop_trip_counts"] == [32, 128]
    assert FACTS["display_dxil"]["grid_channel_sampled"] == ["3"]                                                # alpha is all the consumer reads
    # the binaries are of today's source's era: the exe's loader negates z like XUSGObjLoader.cpp:198,213 -- the x-mirror of the
    # screenshots (tests/test_visual_ab.py) is older than every binary the reference ships
    assert EXE["loader_scanf_sites"] == 2 and EXE["loader_negates_third_float_after_each"] == [True, True]
    assert EXE["default_mesh"] == "Assets/bunny.obj" and EXE["link_date_utc"]["DXRVoxelizer.exe"] >= "2025-01-01"


def test_oracle_ray_is_the_compiled_raygen(orc):
    rng = np.random.default_rng(7)
    for N in (2, 64, 254, 512, 2048):
        for ix, iy, iz in rng.integers(0, N, (200, 3)):
            o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
            orc.lib().orc_ray_reference(N, int(ix), int(iy), int(iz), o, d)
            f = lambda i: np.float32(np.float32(np.float32(np.float32(i) + np.float32(0.5)) / np.float32(N)) * np.float32(2.0)) - np.float32(1.0)   # noqa: E731
            assert o[0] == f(ix) and o[1] == -f(iy) and o[2] == f(iz)                     # +0.5, /dim, *2, -1; y negated, x and z not
            # direction: origin * rsqrt(dot3) in the binary, last bits left to the driver (1 ulp rsqrt, fused or unfused dot3): the
            # oracle's p / sqrtf((xx + yy) + zz) stays within 3 ulp of the exact quotient (three roundings in the sum, one in the root,
            # one in the division) -- the same envelope as x * rsqrt(dot3) with a 1 ulp rsqrt and an unfused dot3
            exact = o.astype(np.float64) / np.sqrt((o.astype(np.float64) ** 2).sum())
            ulp = np.spacing(np.abs(exact).astype(np.float32)).astype(np.float64)
            assert np.all(np.abs(d.astype(np.float64) - exact) <= 3.0 * ulp), (N, ix, iy, iz)


def _scene(orc, nrm3):
    """a large triangle in the plane z = 0.5 with the three vertex normals `nrm3`, and a tiny one in a corner of the plane z = -0.5
    that no ray of these tests meets: the mesh's box is [-1, 1] x [-1, 1] x [-0.5, 0.5], so the normalising transform
    (Content/Voxelizer.cpp:52-57) is the identity"""
    pos = np.array([[-1, -1, 0.5], [1, -1, 0.5], [0, 1, 0.5], [-1, -1, -0.5], [-0.99, -1, -0.5], [-1, -0.99, -0.5]], np.float32)
    nrm = np.concatenate([np.asarray(nrm3, np.float32).reshape(3, 3), np.tile(np.float32([0, 0, -1]), (3, 1))])
    return orc.Scene(np.concatenate([pos, nrm], axis=1), np.array([0, 1, 2, 3, 4, 5], np.uint32))


def _one_triangle_scene(orc, normal):
    return _scene(orc, np.tile(np.asarray(normal, np.float32), (3, 1)))


def test_oracle_predicate_is_the_compiled_closest_hit(orc):
    N, v = 64, (33, 30, 40)                        # a voxel below the plane whose radial ray crosses the triangle
    o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
    orc.lib().orc_ray_reference(N, *v, o, d)
    thr = struct.unpack("<f", struct.pack("<I", int(VOX["closest_hit"]["predicate"]["threshold_f32_bits"], 16)))[0]
    assert thr == np.float32(0.12)
    # a unit normal n with n . d = c: n = c d + sqrt(1 - c^2) e, e a unit vector perpendicular to d
    e = np.cross(d.astype(np.float64), [1.0, 0.0, 0.0]); e /= np.linalg.norm(e)
    occ = {}
    for c in (0.12 - 1e-4, 0.12 + 1e-4, -0.5, 0.9):
        n = c * d.astype(np.float64) + np.sqrt(1 - c * c) * e
        s = _one_triangle_scene(orc, 3.0 * n)      # (not unit: closestHitMain normalises, hlsl:137)
        got, t, k, b, _ = s.voxel(N, *v)
        assert k == 0 and 0.0 < t < 1e4            # the ray does hit
        occ[c] = got
    assert occ == {0.12 - 1e-4: 0, 0.12 + 1e-4: 1, -0.5: 0, 0.9: 1}                       # `fcmp ogt 0.12`: strictly greater
    # missMain is `ret void` and raygenMain stores only when isInside != 0: a ray that hits nothing leaves its voxel untouched (0)
    s = _one_triangle_scene(orc, d)
    got, t, k, _, _ = s.voxel(N, 33, 30, 10)       # the same direction mirrored in z: leaves the scene on the other side
    assert got == 0 and k == 0xFFFFFFFF


def test_oracle_interpolation_order_is_the_compiled_one(orc):
    # (n0 + (n1 - n0) * b.x) + (n2 - n0) * b.y in float32 -- b.x weighs vertex 1, b.y vertex 2 (DXR's barycentrics): three
    # different vertex normals, the oracle's texel normal against that expression evaluated here
    nrm = np.array([[0.1, 0.2, 1.0], [-0.3, 0.1, 0.8], [0.2, -0.4, 0.9]], np.float32)
    s = _scene(orc, nrm)
    N, v = 64, (33, 30, 40)
    occ, t, k, b, texel = s.voxel(N, *v)
    assert k == 0
    f = np.float32
    n = [f(f(nrm[0][a] + f(b[0] * f(nrm[1][a] - nrm[0][a]))) + f(b[1] * f(nrm[2][a] - nrm[0][a]))) for a in range(3)]
    ln = np.sqrt(f(f(f(n[0] * n[0]) + f(n[1] * n[1])) + f(n[2] * n[2])), dtype=np.float32)
    n = [f(c / ln) for c in n]
    o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
    orc.lib().orc_ray_reference(N, *v, o, d)
    dot = f(f(f(n[0] * d[0]) + f(n[1] * d[1])) + f(n[2] * d[2]))
    assert occ == int(dot > f(0.12))
    if occ:                                        # the stored texel is float4(normal, 1) in R10G10B10A2_UNORM (hlsl:83-84)
        want = [int(np.floor(min(max(float(c), 0.0), 1.0) * 1023.0 + 0.5)) for c in n]
        assert [(texel >> s_) & 1023 for s_ in (0, 10, 20)] == want and texel >> 30 == 3


def test_oracle_loader_negates_z_like_the_shipped_exe(orc, tmp_path):
    p = tmp_path / "one.obj"
    p.write_text("v 1 2 3\nv 4 5 7\nv -1 0 2\nvn 0 0 1\nvn 0 0 1\nvn 0 0 1\nf 1//1 2//2 3//3\n")
    vb, ib, _ = orc.obj_load(str(p))
    assert sorted(vb[:, 2].tolist()) == [-7.0, -3.0, -2.0] and set(vb[:, 5].tolist()) == {-1.0}
    assert sorted(vb[:, 0].tolist()) == [-1.0, 1.0, 4.0]                                   # x and y as written
