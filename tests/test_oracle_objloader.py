"""A1 mesh ingest: the oracle's restatement and the PRODUCT's loader (csrc/obj_ingest.cpp) against
the reference's own ObjLoader (oracle/_ref build, outputs committed under tests/golden)."""
import os

import numpy as np
import pytest

from conftest import GOLD, REF_ASSETS, load_mesh

HAND = ["quad_poly_neg", "split_vn"]


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("name", HAND)
def test_oracle_loader_handwritten(orc, name):
    want = np.load(os.path.join(GOLD, "obj", name + ".npz"))
    vb, ib, aabb = orc.obj_load(os.path.join(GOLD, "obj", name + ".obj"))
    assert same_bits(vb, want["vb"]) and same_bits(ib, want["ib"]) and same_bits(aabb, want["aabb"])


@pytest.mark.parametrize("name", HAND)
def test_product_loader_handwritten(dxvlib, name):
    import dxrvoxelizer_amd as dxv
    want = np.load(os.path.join(GOLD, "obj", name + ".npz"))
    vb, ib, aabb = dxv.obj_load(os.path.join(GOLD, "obj", name + ".obj"))
    assert same_bits(vb, want["vb"]) and same_bits(ib, want["ib"]) and same_bits(aabb, want["aabb"])


def test_split_vn_semantics():
    """XUSGObjLoader.cpp:300-335: vertices 1 and 3 meet a second vn -> duplicated at the end."""
    want = np.load(os.path.join(GOLD, "obj", "split_vn.npz"))
    assert want["vb"].shape[0] > 5           # 5 file vertices + splits
    n = want["vb"][:, 3:]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)   # file normals are normalised
    assert np.all(want["vb"][:, 2] <= 0.0)   # z negated (:198): the file has z >= 0 only


def test_polygon_fan_and_reversal():
    """Fan triangulation (:263-297), negative indices (:243) and whole-array reversal (:227)."""
    want = np.load(os.path.join(GOLD, "obj", "quad_poly_neg.npz"))
    ib = want["ib"]
    assert ib.size == 6 * 3                  # quad -> 2, four triangles
    # file order: (0,1,2) (0,2,3) (0,1,4) (1,2,4) (2,3,4) (3,0,4); reversed as one array
    file_order = np.array([0, 1, 2, 0, 2, 3, 0, 1, 4, 1, 2, 4, 2, 3, 4, 3, 0, 4], np.uint32)
    assert np.array_equal(ib, file_order[::-1])


@pytest.mark.parametrize("name,file,V,nidx", [("bunny", "bunny.obj", 34835, 208998),
                                              ("dragon", "dragon.obj", 50000, 300000),
                                              ("turingbowl", "TuringBowl.obj", 23188, 68232)])
def test_asset_fixtures_and_loaders(orc, dxvlib, name, file, V, nidx):
    vb, ib, aabb = load_mesh(name)           # produced by the reference's loader (gen_fixtures.py)
    assert vb.shape == (V, 6) and ib.shape == (nidx,)
    assert not np.isnan(vb).any()
    path = os.path.join(REF_ASSETS, file)
    if not os.path.exists(path):
        pytest.skip("reference assets not present on this machine")
    import dxrvoxelizer_amd as dxv
    for loader in (orc.obj_load, dxv.obj_load):
        v2, i2, a2 = loader(path)
        assert same_bits(v2, vb) and same_bits(i2, ib) and same_bits(a2, aabb)
    if os.path.exists(os.path.join(os.path.dirname(orc.__file__), "_ref", "ref_objloader")):
        v3, i3, a3 = orc.ref_objloader(path)
        assert same_bits(v3, vb) and same_bits(i3, ib) and same_bits(a3, aabb)


def test_survey_probe_values(bunny, dragon):
    """Pins recorded in SURVEY.md section 8(a) row A1 from the reference loader."""
    vb, ib, aabb = bunny
    assert np.allclose(aabb, [-5.0151, -0.0442, -3.8870, 5.0151, 9.8982, 3.8870], atol=1e-4)
    assert np.allclose(vb[0], [1.487, 0.3736, -2.2576, -0.23844783, -0.895268261, 0.376347363], atol=1e-6)
    assert list(ib[:6]) == [34834, 33422, 12706, 34834, 12706, 22064]
    vb, ib, aabb = dragon
    assert np.allclose(aabb, [-7.0467, 0.0, -3.1513, 7.0467, 9.9399, 3.1513], atol=1e-4)
    assert list(ib[:6]) == [47437, 42256, 29824, 29823, 47437, 29824]


def test_bound_matches_reference_formula(orc, bunny, dragon):
    """Content/Voxelizer.cpp:52-57."""
    for (vb, ib, aabb), want in ((bunny, (0.0, 4.927, 0.0, 5.0151)), (dragon, (0.0, 4.96995, 0.0, 7.0467))):
        a2, b = orc.bound(vb)
        assert np.array_equal(a2, aabb)
        assert np.allclose(b, want, atol=1e-4)
