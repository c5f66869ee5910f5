"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (dxrvoxelizer_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODE_REFERENCE, MODE_PARITY = 0, 1
ALGO_BRUTE, ALGO_BVH, ALGO_PLAIN = 0, 1, 2

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "dxv_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_obj_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32),
                                   C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32), _f32p]
        L.orc_obj_load.restype = C.c_int
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_aabb.argtypes = [_f32p, C.c_uint32, _f32p]
        L.orc_bound.argtypes = [_f32p, _f32p]
        L.orc_scene_create.argtypes = [_f32p, C.c_uint32, _u32p, C.c_uint32]
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_bound.argtypes = [C.c_void_p, _f32p]
        L.orc_scene_tri.argtypes = [C.c_void_p, C.c_uint32, _f32p, _f32p]
        L.orc_ray_reference.argtypes = [C.c_uint32] * 4 + [_f32p, _f32p]
        L.orc_slab.argtypes = [_f32p, _f32p, _f32p, _f32p, C.POINTER(C.c_float)]
        L.orc_slab.restype = C.c_int
        L.orc_tri_test.argtypes = [_f32p] * 5 + [C.c_int] + [C.POINTER(C.c_float)] * 3
        L.orc_tri_test.restype = C.c_int
        L.orc_voxel_reference.argtypes = [C.c_void_p] + [C.c_uint32] * 4 + [C.c_int, C.POINTER(C.c_float),
                                          C.POINTER(C.c_uint32), _f32p, C.POINTER(C.c_uint32)]
        L.orc_voxel_reference.restype = C.c_int
        L.orc_voxelize.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int,
                                   _u8p, C.c_void_p]
        L.orc_voxelize.restype = C.c_int
        L.orc_voxelize_slices.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, _u32p, C.c_uint32, C.c_int, _u8p]
        L.orc_voxelize_slices.restype = C.c_int
        L.orc_update_frame.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_float, C.c_float, _f32p, _f32p, _f32p]
        L.orc_update_frame.restype = C.c_int
        L.orc_render.argtypes = [_u8p, C.c_uint32, _f32p, _f32p, _f32p, _f32p, C.c_uint32, C.c_uint32, C.c_void_p, _u8p]
        L.orc_render.restype = C.c_int
        L.orc_num_procs.restype = C.c_int
        _LIB = L
    return _LIB


def obj_load(path):
    """(vb [V,6] f32, ib [3T] u32, aabb [6] f32) as XUSG::ObjLoader::Import(path, true, true)."""
    L = lib()
    vb, ib = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)()
    V, n = C.c_uint32(), C.c_uint32()
    aabb = np.zeros(6, np.float32)
    rc = L.orc_obj_load(os.fsencode(path), C.byref(vb), C.byref(V), C.byref(ib), C.byref(n), aabb)
    if rc:
        raise RuntimeError(f"orc_obj_load({path}) -> {rc}")
    vba = np.ctypeslib.as_array(vb, (V.value, 6)).copy()
    iba = np.ctypeslib.as_array(ib, (n.value,)).copy()
    L.orc_free(vb)
    L.orc_free(ib)
    return vba, iba, aabb


def bound(vb):
    L = lib()
    vb = np.ascontiguousarray(vb, np.float32)
    aabb, b = np.zeros(6, np.float32), np.zeros(4, np.float32)
    L.orc_aabb(vb, len(vb), aabb)
    L.orc_bound(aabb, b)
    return aabb, b


class Scene:
    def __init__(self, vb, ib):
        self.vb = np.ascontiguousarray(vb, np.float32).reshape(-1, 6)
        self.ib = np.ascontiguousarray(ib, np.uint32).reshape(-1)
        self.T = len(self.ib) // 3
        self._h = lib().orc_scene_create(self.vb, len(self.vb), self.ib, self.T)
        if not self._h:
            raise RuntimeError("orc_scene_create failed (empty / out-of-range / degenerate mesh)")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_destroy(self._h)
            self._h = None

    @property
    def bound(self):
        b = np.zeros(4, np.float32)
        lib().orc_scene_bound(self._h, b)
        return b

    def tri(self, k):
        p, b = np.zeros(9, np.float32), np.zeros(6, np.float32)
        lib().orc_scene_tri(self._h, k, p, b)
        return p.reshape(3, 3), b

    def voxel(self, N, ix, iy, iz, algo=ALGO_BVH):
        t, k, tex = C.c_float(), C.c_uint32(), C.c_uint32()
        b = np.zeros(2, np.float32)
        occ = lib().orc_voxel_reference(self._h, N, ix, iy, iz, algo, C.byref(t), C.byref(k), b, C.byref(tex))
        return occ, t.value, k.value, b, tex.value

    def voxelize(self, N, mode=MODE_REFERENCE, algo=ALGO_BVH, z0=0, nz=None, threads=0, texels=False):
        nz = N - z0 if nz is None else nz
        out = np.zeros((nz, N, N), np.uint8)
        tex = np.zeros((nz, N, N), np.uint32) if texels else None
        rc = lib().orc_voxelize(self._h, N, mode, algo, z0, nz, threads, out,
                                tex.ctypes.data_as(C.c_void_p) if texels else None)
        if rc:
            raise RuntimeError(f"orc_voxelize -> {rc}")
        return (out, tex) if texels else out


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota (a
    container limited to 16 CPUs of a 256-thread box must not run 256 OpenMP threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except (OSError, ValueError):
            pass
    return n


def voxelize_slices(scene, N, zlist, mode=MODE_REFERENCE, algo=ALGO_BVH, threads=0):
    z = np.ascontiguousarray(zlist, np.uint32)
    out = np.zeros((len(z), N, N), np.uint8)
    rc = lib().orc_voxelize_slices(scene._h, N, mode, algo, z, len(z), threads, out)
    if rc:
        raise RuntimeError(f"orc_voxelize_slices -> {rc}")
    return out


def update_frame(bound, eye, view_proj, width, height, pos_scale=(0, 0, 0, 1)):
    """(lightPt[3], eyePt[3], screenToLocal[16]) as Voxelizer::UpdateFrame builds them."""
    li, ey, m = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(16, np.float32)
    rc = lib().orc_update_frame(np.ascontiguousarray(bound, np.float32), np.ascontiguousarray(pos_scale, np.float32),
                                np.ascontiguousarray(eye, np.float32), np.ascontiguousarray(view_proj, np.float32).reshape(-1),
                                float(width), float(height), li, ey, m)
    if rc:
        raise RuntimeError("orc_update_frame: singular matrix chain")
    return li, ey, m


def render(grid, bound, eye, view_proj, width, height, pos_scale=(0, 0, 0, 1), cb=None):
    """uint8 [height, width, 4] image of the display pass over a full uint8 grid [N, N, N]."""
    g = np.ascontiguousarray(grid, np.uint8)
    out = np.zeros((height, width, 4), np.uint8)
    cbp = None
    if cb is not None:
        cbv = np.ascontiguousarray(np.concatenate([np.ravel(c) for c in cb]), np.float32)
        cbp = cbv.ctypes.data_as(C.c_void_p)
    rc = lib().orc_render(g.reshape(-1), g.shape[0], np.ascontiguousarray(bound, np.float32),
                          np.ascontiguousarray(pos_scale, np.float32), np.ascontiguousarray(eye, np.float32),
                          np.ascontiguousarray(view_proj, np.float32).reshape(-1), width, height, cbp, out.reshape(-1))
    if rc:
        raise RuntimeError("orc_render failed")
    return out


def ref_objloader(path, out_bin=None):
    """Run the reference's own ObjLoader (oracle/_ref/ref_objloader). Returns (vb, ib, aabb)."""
    exe = os.path.join(_HERE, "_ref", "ref_objloader")
    if not os.path.exists(exe):
        raise FileNotFoundError(exe)
    import tempfile
    tmp = out_bin or tempfile.mktemp(suffix=".bin")
    subprocess.check_call([exe, path, tmp])
    raw = np.fromfile(tmp, np.uint8)
    if out_bin is None:
        os.unlink(tmp)
    V, n, stride = np.frombuffer(raw[:12].tobytes(), np.uint32)
    aabb = np.frombuffer(raw[12:36].tobytes(), np.float32).copy()
    vb = np.frombuffer(raw[36:36 + V * stride].tobytes(), np.float32).reshape(V, stride // 4).copy()
    ib = np.frombuffer(raw[36 + V * stride:36 + V * stride + 4 * n].tobytes(), np.uint32).copy()
    return vb, ib, aabb
