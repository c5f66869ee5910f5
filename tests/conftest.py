import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")
REF_ASSETS = "/root/reference/Bin/Assets"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = None


def has_gpu():
    global HAS_GPU
    if HAS_GPU is None:
        HAS_GPU = _has_gpu()
    return HAS_GPU


def pytest_collection_modifyitems(config, items):
    # a gpu test on a box without a GPU is an error of the invocation, not a silent pass
    for item in items:
        if "gpu" in item.keywords and not has_gpu():
            item.add_marker(pytest.mark.skip(reason="no GPU in this container (run with gpurun)"))


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc
    _orc.lib()
    return _orc


def load_mesh(name):
    d = np.load(os.path.join(GOLD, "meshes", name + ".npz"))
    return d["vb"], d["ib"], d["aabb"]


@pytest.fixture(scope="session")
def bunny():
    return load_mesh("bunny")


@pytest.fixture(scope="session")
def dragon():
    return load_mesh("dragon")


@pytest.fixture(scope="session")
def turingbowl():
    return load_mesh("turingbowl")


@pytest.fixture(scope="session")
def grids_json():
    import json
    with open(os.path.join(GOLD, "grids.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def grids64():
    return np.load(os.path.join(GOLD, "grids64.npz"))


@pytest.fixture(scope="session")
def dxvlib():
    """Build (if needed) and load libdxv.so."""
    from dxrvoxelizer_amd import build as _b
    _b.build()
    import dxrvoxelizer_amd
    return dxrvoxelizer_amd.load_library()


@pytest.fixture(scope="session")
def hostcheck():
    """The product's __host__ __device__ code compiled for the CPU (tests/hostcheck)."""
    src = os.path.join(ROOT, "tests", "hostcheck", "hostcheck.cpp")
    so = os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so")
    deps = [src] + [os.path.join(ROOT, "dxrvoxelizer_amd", "csrc", h) for h in ("dxv_math.h", "dxv_trace.h", "dxv_types.h", "dxv_dirmap.h", "dxv_raycast.h", "dxv_policy.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
                               "-mavx2", "-mfma", "-Wno-unknown-pragmas", "-o", so, src])
    L = C.CDLL(so)
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
    u32p = np.ctypeslib.ndpointer(np.uint32, flags="C")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
    L.hc_scene_create.restype = C.c_void_p
    L.hc_scene_create.argtypes = [f32p, C.c_uint32, u32p, C.c_uint32, f32p]
    L.hc_scene_destroy.argtypes = [C.c_void_p]
    L.hc_scene_height.argtypes = [C.c_void_p]
    L.hc_scene_height.restype = C.c_uint32
    L.hc_scene_nodes.argtypes = [C.c_void_p, C.c_void_p]
    L.hc_scene_keys.argtypes = [C.c_void_p, C.c_void_p]
    L.hc_scene_nodes32.argtypes = [C.c_void_p, C.c_void_p]
    L.hc_scene_nodes64.argtypes = [C.c_void_p, C.c_void_p]
    L.hc_dirmap_build.argtypes = [C.c_void_p, C.c_uint32]
    L.hc_dirmap_build.restype = C.c_uint64
    L.hc_dirmap_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hc_half_down.argtypes = [C.c_float]
    L.hc_half_down.restype = C.c_uint32
    L.hc_half_up.argtypes = [C.c_float]
    L.hc_half_up.restype = C.c_uint32
    L.hc_dm_half_pos.argtypes = [C.c_float, C.c_int]
    L.hc_dm_half_pos.restype = C.c_uint32
    L.hc_half_to_float.argtypes = [C.c_uint32]
    L.hc_half_to_float.restype = C.c_float
    L.hc_voxelize.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_int, u8p, C.c_void_p]

    class Host:
        def __init__(self, vb, ib, bound):
            self.vb = np.ascontiguousarray(vb, np.float32)
            self.ib = np.ascontiguousarray(ib, np.uint32)
            self.T = len(self.ib) // 3
            self.h = L.hc_scene_create(self.vb, len(self.vb), self.ib, self.T, np.ascontiguousarray(bound, np.float32))

        def __del__(self):
            if getattr(self, "h", None):
                L.hc_scene_destroy(self.h)
                self.h = None

        @property
        def height(self):
            return L.hc_scene_height(self.h)

        def nodes(self):
            out = np.empty((max(self.T - 1, 1), 16), np.uint32)
            L.hc_scene_nodes(self.h, out.ctypes.data_as(C.c_void_p))
            return out

        def lists(self, R):
            """direction-space lists built on the host with the product's footprint code: (cells, entries)"""
            n = L.hc_dirmap_build(self.h, R)
            cells = np.empty((6 * R * R, 4), np.uint32)
            entries = np.empty((n, 4), np.uint32)
            L.hc_dirmap_get(self.h, cells.ctypes.data_as(C.c_void_p), entries.ctypes.data_as(C.c_void_p))
            return cells, entries

        def nodes32(self):
            out = np.empty((max(self.T - 1, 1), 8), np.uint32)
            L.hc_scene_nodes32(self.h, out.ctypes.data_as(C.c_void_p))
            return out

        def nodes64(self):
            out = np.empty((max(self.T - 1, 1), 16), np.uint32)
            L.hc_scene_nodes64(self.h, out.ctypes.data_as(C.c_void_p))
            return out

        def keys(self):
            out = np.empty(self.T, np.uint64)
            L.hc_scene_keys(self.h, out.ctypes.data_as(C.c_void_p))
            return out

        def voxelize(self, N, mode=0, z0=0, nz=None, stack=64, texels=False):
            nz = N - z0 if nz is None else nz
            g = np.zeros((nz, N, N), np.uint8)
            t = np.zeros((nz, N, N), np.uint32) if texels else None
            ovf = L.hc_voxelize(self.h, N, mode, z0, nz, stack, g, t.ctypes.data_as(C.c_void_p) if texels else None)
            return (g, t, ovf) if texels else (g, ovf)

    L.hc_plan_check.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.hc_dirmap_mip.argtypes = [C.c_void_p, C.c_void_p]

    def plan_check(self, N, z0=0, nz=None, zBlock=None, zPeriod=None):
        """(live voxels, bricks with a live voxel, bricks the work queue's box test keeps, violations) for a partition, against the
        lists built last (Host.lists)"""
        nz = N - z0 if nz is None else nz
        out = np.zeros(4, np.uint64)
        L.hc_plan_check(self.h, N, z0, nz, nz if zBlock is None else zBlock, nz if zPeriod is None else zPeriod, out.ctypes.data_as(C.c_void_p))
        return tuple(int(v) for v in out)

    def mip(self, R):
        out = np.zeros(sum(6 * (R >> l) ** 2 for l in range(R.bit_length())), np.uint16)
        L.hc_dirmap_mip(self.h, out.ctypes.data_as(C.c_void_p))
        return out

    Host.plan_check = plan_check
    Host.mip = mip
    Host.lib = L
    return Host
