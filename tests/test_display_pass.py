"""N3 -- the grid's consumer (Voxelizer::UpdateFrame + renderRayCast, Content/Voxelizer.cpp:81-106,
:371-399, Shaders/PSRayCast.hlsl): the product's per-pixel march against the oracle's restatement.
Tolerance: none needed on the CPU (same float32 operation order); the GPU test allows 1/255."""
import ctypes as C

import numpy as np
import pytest

from dxrvoxelizer_amd import camera

W, H = 160, 90


def host_render(hostcheck, grid, cb):
    L = hostcheck.lib
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
    L.hc_render.argtypes = [u8p, C.c_uint32, f32p, C.c_uint32, C.c_uint32, u8p]
    img = np.zeros((H, W, 4), np.uint8)
    L.hc_render(np.ascontiguousarray(grid).reshape(-1), grid.shape[0], cb, W, H, img.reshape(-1))
    return img


def host_cb(hostcheck, bound, eye, vp, pos_scale=(0, 0, 0, 1)):
    L = hostcheck.lib
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
    L.hc_update_frame.argtypes = [f32p, f32p, f32p, f32p, C.c_float, C.c_float, f32p]
    cb = np.zeros(22, np.float32)
    assert L.hc_update_frame(np.ascontiguousarray(bound, np.float32), np.asarray(pos_scale, np.float32), eye,
                             np.ascontiguousarray(vp).reshape(-1), W, H, cb) == 0
    return cb


def test_update_frame_constants(orc, hostcheck, bunny):
    """Voxelizer.cpp:81-106: light/eye in local space, screenToLocal; default camera of the app."""
    vb, ib, _ = bunny
    _, bound = orc.bound(vb)
    eye, vp = camera.default_view_proj(W, H)
    cb = host_cb(hostcheck, bound, eye, vp)
    li, ey, m = orc.update_frame(bound, eye, vp, W, H)
    assert np.allclose(cb[:3], li, rtol=1e-6) and np.allclose(cb[3:6], ey, rtol=1e-6)
    assert np.allclose(cb[6:], m, rtol=1e-5, atol=1e-7 * np.abs(m).max())
    # the eye in local space: (eye - c) / w
    assert np.allclose(ey, (np.asarray(camera.DEFAULT_EYE) - bound[:3]) / bound[3], rtol=1e-5)
    # posScale moves/scales the display transform only (Voxelizer.cpp:84-87)
    li2, ey2, _ = orc.update_frame(bound, eye, vp, W, H, pos_scale=(1.0, 2.0, 3.0, 2.0))
    assert np.allclose(ey2, ((np.asarray(camera.DEFAULT_EYE) - (1, 2, 3)) / 2.0 - bound[:3]) / bound[3], rtol=1e-5)


def test_product_march_equals_oracle(orc, hostcheck, bunny):
    vb, ib, _ = bunny
    s = orc.Scene(vb, ib)
    grid = s.voxelize(32)
    for eye_pos in (camera.DEFAULT_EYE, (-6.0, 3.0, 13.0)):
        eye, vp = camera.default_view_proj(W, H, eye=eye_pos)
        cb = host_cb(hostcheck, s.bound, eye, vp)
        img = host_render(hostcheck, grid, cb)
        want = orc.render(grid, s.bound, eye, vp, W, H)
        assert np.array_equal(img, want)
        assert 0.05 < (want[..., 3] == 255).mean() < 0.9            # the cube is in view, not the whole screen


def test_march_known_answers(orc):
    """Empty grid: inside the cube's silhouette the pixel is sqrt(clear^2) = clear with alpha 1,
    outside it is the clear colour with alpha 0 (PSRayCast.hlsl:122, :184-186).  Full grid: opaque,
    brighter than the background."""
    eye, vp = camera.default_view_proj(W, H)
    bound = np.array([0, 4, 0, 5], np.float32)
    empty = np.zeros((8, 8, 8), np.uint8)
    img = orc.render(empty, bound, eye, vp, W, H)
    inside = img[..., 3] == 255
    assert inside.any() and (~inside).any()
    assert np.all(img[inside][:, :3] == np.array([0, 51, 102]))    # round(255 * (0, .2, .4))
    assert np.all(img[~inside] == np.array([0, 51, 102, 0]))
    full = orc.render(np.ones((8, 8, 8), np.uint8), bound, eye, vp, W, H)
    assert np.array_equal(full[..., 3] == 255, inside)
    assert np.median(full[inside][:, 0]) > 100                      # scatter * 0.8 + 0.2 under sqrt (edge pixels graze)


@pytest.mark.gpu
def test_gpu_render_equals_oracle(dxvlib, orc, bunny, tmp_path):
    import dxrvoxelizer_amd as dxv
    vb, ib, _ = bunny
    v = dxv.Voxelizer(0)
    v.InitFromArrays(vb, ib)
    v.Voxelize(64)
    grid = v.Grid()
    s = orc.Scene(vb, ib)
    for (w, h), eye_pos in (((320, 180), camera.DEFAULT_EYE), ((200, 120), (-6.0, 3.0, 13.0))):
        eye, vp = camera.default_view_proj(w, h, eye=eye_pos)
        img = v.Render(eye, vp, w, h)
        want = orc.render(grid, s.bound, eye, vp, w, h)
        diff = np.abs(img.astype(np.int16) - want.astype(np.int16))
        assert diff.max() <= 1 and (diff != 0).mean() < 1e-3, (int(diff.max()), float((diff != 0).mean()))
    camera.write_png(str(tmp_path / "bunny.png"), img)
    assert (tmp_path / "bunny.png").stat().st_size > 1000
    v.Voxelize(64, 0, 0, 32)                                        # a slab is not renderable
    with pytest.raises(dxv.DxvError):
        v.Render(eye, vp, w, h)
    v.close()


@pytest.mark.gpu
def test_gpu_empty_brick_skip_changes_no_pixel(dxvlib, bunny):
    """The display pass skips the samples of empty 8^3 bricks (option skipempty, default on): the image
    must equal the plain march byte for byte, also at grid sizes that are not multiples of the brick."""
    import dxrvoxelizer_amd as dxv
    vb, ib, _ = bunny
    v = dxv.Voxelizer(0)
    v.InitFromArrays(vb, ib)
    for n in (16, 50, 64, 100):
        v.Voxelize(n)
        for (w, h), eye_pos in (((320, 180), camera.DEFAULT_EYE), ((128, 128), (-6.0, 3.0, 13.0)), ((96, 64), (0.5, 0.2, 1.5))):
            eye, vp = camera.default_view_proj(w, h, eye=eye_pos)
            v.set_option("skipempty", 1)
            a = v.Render(eye, vp, w, h)
            v.set_option("skipempty", 0)
            b = v.Render(eye, vp, w, h)
            assert np.array_equal(a, b), (n, w, h)
        assert a.any()
    with pytest.raises(dxv.DxvError):
        v.set_option("skipempty", 2)
    v.close()
