"""Visual A/B against the reference's own screenshots (oracle/visual_ab.py made the fixtures from Doc/Images/*.jpg in the build
container; here they are data).  The only output of the reference's hot path that the reference holds: its default mesh at its own
64^3 through its own DXR pipeline and ray-cast pass, start-up camera, 1280 x 720 client area.  Re-rendered here from the oracle
(voxelizer -> display pass) and compared by silhouette (distance from the clear colour) and shading.

FINDING kept under test: the screenshots are the restated pipeline's image of the scene mirrored in x -- IoU 0.998 and 0.7 / 255 mean
colour difference at 64^3 -- while the scene as today's source reads it (loader output pinned to the compiled reference loader)
gives 0.55: the screenshots predate one reflection of the chain.  Given that one reflection, every convention a silhouette can show
is pinned: y flip (hlsl:49), loader z negation and index reversal, sign of the predicate (hlsl:137-138), alpha (hlsl:84), the
normalising transform, camera, march and lighting; a shift of one voxel is visible.  Not pinned: single voxels (the reference rule and
the parity rule give the same silhouette), the grid size of the second screenshot."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))


@pytest.fixture(scope="module")
def ab():
    PIL = pytest.importorskip("PIL.Image")
    import visual_ab as va
    d = np.load(os.path.join(ROOT, "tests", "golden", "meshes", "bunny.npz"))
    shot = np.asarray(PIL.open(os.path.join(ROOT, "tests", "golden", "visual", "reference_client_area_1280x720.jpg")).convert("RGB")).astype(np.float64)
    res, _, mask = va.compare(shot, d["vb"], d["ib"], 64)
    return res, int(mask.sum()), json.load(open(os.path.join(ROOT, "tests", "golden", "visual_ab.json")))


def test_screenshot_is_the_restated_pipeline_mirrored_in_x(orc, ab):
    res, pixels, fixture = ab
    assert pixels == fixture["screenshot_object_pixels"]
    best = res["restated, scene mirrored in x"]
    assert best["iou"] >= 0.995 and best["mean_abs_rgb_diff_inside"] <= 1.5
    assert max(abs(v) for v in best["centroid_offset_px"]) < 1.0 and max(abs(v) for v in best["bbox_offset_px"]) <= 1
    # (the documented discrepancy: the scene as the current source reads it is NOT what the screenshot shows)
    assert 0.45 < res["restated, as the current source reads"]["iou"] < 0.65
    # the committed numbers are these numbers
    for name, m in res.items():
        assert abs(m["iou"] - fixture["variants"][name]["iou"]) < 2e-3, name


def test_the_measure_tells_wrong_conventions_apart(orc, ab):
    res, _, _ = ab
    best = res["restated, scene mirrored in x"]["iou"]
    for name, m in res.items():
        if not name.startswith("mirrored + ") or "parity" in name:
            continue
        assert m["iou"] < best - 0.05, (name, m["iou"])                 # every wrong variant loses clearly, a one-voxel shift included
        if "shifted" not in name and "10 %" not in name:
            assert m["iou"] < 0.7, (name, m["iou"])
    # ... and what it cannot tell apart is said so: ~60 voxels of 262,144 do not move a silhouette
    assert abs(res["mirrored + parity rule instead of the reference rule"]["iou"] - best) < 1e-3


def test_second_screenshot_agrees(orc):
    PIL = pytest.importorskip("PIL.Image")
    import visual_ab as va
    from dxrvoxelizer_amd import camera
    d = np.load(os.path.join(ROOT, "tests", "golden", "meshes", "bunny.npz"))
    shot = np.asarray(PIL.open(os.path.join(ROOT, "tests", "golden", "visual", "reference_hires_client_area_1280x720.jpg")).convert("RGB")).astype(np.float64)
    scene = orc.Scene(d["vb"], d["ib"])
    g = scene.voxelize(128)                                              # (128^3 here for the suite's time; 256^3 in the fixture: 0.996)
    eye, vp = camera.default_view_proj(va.W, va.H)
    ref = va.silhouette(shot)
    m = va.measures(va.silhouette(orc.render(np.ascontiguousarray(g[:, :, ::-1]), scene.bound, eye, vp, va.W, va.H)), ref)
    w = va.measures(va.silhouette(orc.render(g, scene.bound, eye, vp, va.W, va.H)), ref)
    assert m["iou"] > 0.98 and w["iou"] < 0.65


def test_root_cause_rows_one_candidate_matches_silhouette_and_shading(orc):
    """Which single difference from today's source gives the screenshot (oracle/visual_ab.py: root_cause_rows): only the file's
    (-x, y, -z) -- its raw coordinates turned by 180 degrees about y -- matches silhouette AND colours; raw coordinates do not, a camera
    on the other side or a mirrored image match the silhouette alone.  The shipped exe and DXIL are today's source
    (tests/test_shipped_binaries.py): the screenshots are older than every binary the reference ships."""
    PIL = pytest.importorskip("PIL.Image")
    import visual_ab as va
    d = np.load(os.path.join(ROOT, "tests", "golden", "meshes", "bunny.npz"))
    shot = np.asarray(PIL.open(os.path.join(ROOT, "tests", "golden", "visual", "reference_client_area_1280x720.jpg")).convert("RGB")).astype(np.float64)
    rows = va.root_cause_rows(shot, d["vb"], d["ib"], 64)
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "visual_ab.json")))["root_cause"]["rows"]
    assert set(rows) == set(fixture)
    for name, m in rows.items():
        assert abs(m["iou"] - fixture[name]["iou"]) < 2e-3 and abs(m["mean_abs_rgb_diff_inside"] - fixture[name]["mean_abs_rgb_diff_inside"]) < 0.2, name
    both = [n for n, m in rows.items() if m["iou"] > 0.99 and m["mean_abs_rgb_diff_inside"] < 1.5]
    assert both == ["(-x, y, -z) of the file = the raw OBJ turned by 180 degrees about y, eye (8, 12, -14)"]
    silhouette_only = [n for n, m in rows.items() if m["iou"] > 0.99 and m["mean_abs_rgb_diff_inside"] >= 1.5]
    assert len(silhouette_only) == 2 and all("eye (-8" in n for n in silhouette_only)
    assert rows["raw OBJ (x, y, z) (no z negation, with or without index reversal), eye (8, 12, -14)"]["iou"] < 0.6
