"""Python mirror of the reference's Voxelizer component (Content/Voxelizer.h:10-24) over the
C-ABI, used by tests and bench.py.  Same call order as the reference: Init (load, upload, bound,
build acceleration structure) then Voxelize per frame."""
import ctypes as C
import os

import numpy as np

from ._lib import DxvError, Stats, load_library

MODE_REFERENCE, MODE_PARITY = 0, 1
DBG_SORTED_KEYS, DBG_NODES, DBG_TRI_POS, DBG_TRI_NRM, DBG_PARENTS, DBG_NODES32, DBG_NODES64, DBG_LIST_CELLS, DBG_LIST_ENTRIES, DBG_LIST_MIP = range(10)


def obj_load(path):
    """(vb [V,6] f32, ib [3T] u32, aabb [6] f32) exactly as XUSG::ObjLoader::Import(path, true, true)
    (XUSG/Optional/XUSGObjLoader.cpp:18-40) -- the product's own loader (csrc/obj_ingest.cpp)."""
    lib = load_library()
    vb, ib = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)()
    nv, ni = C.c_uint32(), C.c_uint32()
    aabb = np.zeros(6, np.float32)
    rc = lib.dxv_obj_load(os.fsencode(path), C.byref(vb), C.byref(nv), C.byref(ib), C.byref(ni), aabb)
    if rc:
        raise DxvError(f"dxv_obj_load({path!r}) failed with code {rc}")
    try:
        return (np.ctypeslib.as_array(vb, (nv.value, 6)).copy(), np.ctypeslib.as_array(ib, (ni.value,)).copy(), aabb)
    finally:
        lib.dxv_free(vb)
        lib.dxv_free(ib)


class Voxelizer:
    """bool-returning calls of the reference become exceptions (DxvError) here."""

    FrameCount = 3  # Content/Voxelizer.h:24

    def __init__(self, device=0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        if self._lib.dxv_create(C.byref(self._ctx), int(device)):
            raise DxvError(self._lib.dxv_last_error(None).decode())
        self.device = int(device)
        self._frame = 0
        self._lasts = {}
        # DXV_OPTIONS="key=value,...": options for every context of a process (A/B runs of the tools without touching them)
        import os
        for kv in filter(None, os.environ.get("DXV_OPTIONS", "").split(",")):
            k, val = kv.split("=")
            self.set_option(k.strip(), int(val))

    # ---- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.dxv_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise DxvError(self._lib.dxv_last_error(self._ctx).decode())

    # ---- reference surface ------------------------------------------------------------------
    def Init(self, fileName, posScale=(0.0, 0.0, 0.0, 1.0), dynamicMesh=False, gridDim=0):
        """Voxelizer::Init (Content/Voxelizer.cpp:30-79) minus the D3D12 arguments."""
        vb, ib, _ = obj_load(fileName)
        return self.InitFromArrays(vb, ib, posScale, dynamicMesh, gridDim)

    def InitFromArrays(self, vb, ib, posScale=(0.0, 0.0, 0.0, 1.0), dynamicMesh=False, gridDim=0):
        """Upload, bound, LBVH -- and, like the reference's Init (Content/Voxelizer.cpp:73), everything else the launches trace
        through: the candidate lists of the reference rule on the map a static scene is launched with, so that the first
        Voxelize costs what every later one costs.  gridDim (the reference's GRID_SIZE, a compile-time constant there,
        Content/Voxelizer.cpp:8): the grid the scene will be voxelized at -- its work queue is then Init-time structure too
        (dxv_prepare_launch) and a Voxelize(gridDim) is one clear + one hardware-dispatched launch; 0: every launch builds its
        queue itself.  dynamicMesh=True (a mesh that is refitted every frame): the LBVH only."""
        self.posScale = tuple(posScale)  # display only in the reference (Voxelizer.cpp:84-87)
        vb = np.ascontiguousarray(vb, np.float32).reshape(-1, 6)
        ib = np.ascontiguousarray(ib, np.uint32).reshape(-1)
        if ib.size % 3:
            raise DxvError("index count is not a multiple of 3")
        self._check(self._lib.dxv_set_mesh(self._ctx, vb, len(vb), ib, ib.size // 3))
        self._check(self._lib.dxv_build(self._ctx))
        if not dynamicMesh:
            self._check(self._lib.dxv_build_lists_for_grid(self._ctx, int(gridDim)))
        return True

    def PrepareLaunch(self, gridDim, z0=0, nz=None):
        """dxv_prepare_launch: the work queue of slices [z0, z0 + nz) of a gridDim^3 grid, built now and kept with the scene."""
        nz = gridDim - z0 if nz is None else nz
        self._check(self._lib.dxv_prepare_launch(self._ctx, int(gridDim), int(z0), int(nz)))
        return True

    def PrepareLaunchInterleaved(self, gridDim, rank, world, zblock=8):
        """dxv_prepare_launch_interleaved: the same for this rank's share of the block-cyclic partition."""
        self._check(self._lib.dxv_prepare_launch_interleaved(self._ctx, int(gridDim), int(rank), int(world), int(zblock)))
        return True

    def InitDynamic(self, vb, ib, posScale=(0.0, 0.0, 0.0, 1.0)):
        return self.InitFromArrays(vb, ib, posScale, dynamicMesh=True)

    def UpdateVertices(self, vb, refit=True):
        """Animated vertices on fixed topology: upload + refit of the existing hierarchy (N4)."""
        vb = np.ascontiguousarray(vb, np.float32).reshape(-1, 6)
        self._check(self._lib.dxv_update_vertices(self._ctx, vb, len(vb)))
        if refit:
            self._check(self._lib.dxv_refit(self._ctx))
        return True

    def Refit(self):
        """dxv_refit on its own: after UpdateVertices(vb, refit=False), whose upload overlapped a launch still in flight."""
        self._check(self._lib.dxv_refit(self._ctx))
        return True

    def UpdateVerticesDevice(self, device_ptr, num_verts, refit=True):
        """The same from a device buffer (6 floats per vertex on this GPU, e.g. a torch tensor's data_ptr()): a mesh animated
        on the GPU never passes through the host."""
        self._check(self._lib.dxv_update_vertices_device(self._ctx, C.c_void_p(int(device_ptr)), int(num_verts)))
        if refit:
            self._check(self._lib.dxv_refit(self._ctx))
        return True

    def SetFrame(self, frameIndex):
        """The frame (0 .. FrameCount-1) the following Voxelize / Sync / Grid / Texels / Render / stats calls refer
        to: the reference's frameIndex argument (Content/Voxelizer.h:20-22) and its m_grids[FrameCount] (:110)."""
        self._check(self._lib.dxv_set_frame(self._ctx, int(frameIndex)))
        self._frame = int(frameIndex)

    def Voxelize(self, gridDim, mode=MODE_REFERENCE, z0=0, nz=None, sync=True, frameIndex=None):
        """Voxelizer::voxelize(pCommandList, frameIndex) (Content/Voxelizer.cpp:351-369) with gridDim as a parameter."""
        if frameIndex is not None:
            self.SetFrame(frameIndex)
        nz = gridDim - z0 if nz is None else nz
        fn = self._lib.dxv_voxelize if sync else self._lib.dxv_voxelize_async
        self._check(fn(self._ctx, int(gridDim), int(mode), int(z0), int(nz)))
        self._lasts[self._frame] = (int(gridDim), int(nz))
        return True

    def VoxelizeInterleaved(self, gridDim, rank, world, zblock=8, mode=MODE_REFERENCE, sync=True, frameIndex=None):
        """This rank's share of a block-cyclic Z partition (dxv_voxelize_interleaved)."""
        if frameIndex is not None:
            self.SetFrame(frameIndex)
        fn = self._lib.dxv_voxelize_interleaved if sync else self._lib.dxv_voxelize_interleaved_async
        self._check(fn(self._ctx, int(gridDim), int(mode), int(rank), int(world), int(zblock)))
        self._lasts[self._frame] = (int(gridDim), int(gridDim) // int(world))
        return True

    def Sync(self):
        self._check(self._lib.dxv_sync(self._ctx))

    def SyncAll(self):
        """Wait for the launches of every frame (dxv_sync_all)."""
        self._check(self._lib.dxv_sync_all(self._ctx))

    @property
    def _last(self):
        return self._lasts.get(self._frame)

    # ---- the grid's consumer (Voxelizer::UpdateFrame + Render's ray-cast pass) -----------------
    def Render(self, eyePt, viewProj, width=1280, height=720, posScale=None):
        """uint8 [height, width, 4] R8G8B8A8 image of the last full grid (dxv_render)."""
        eye = np.ascontiguousarray(eyePt, np.float32).reshape(3)
        vp = np.ascontiguousarray(viewProj, np.float32).reshape(16)
        ps = None if posScale is None else np.ascontiguousarray(posScale, np.float32).ctypes.data_as(C.c_void_p)
        out = np.empty((int(height), int(width), 4), np.uint8)
        self._check(self._lib.dxv_render(self._ctx, eye, vp, ps, int(width), int(height), out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- results ----------------------------------------------------------------------------
    def Grid(self):
        """uint8 [nz, N, N] (z, y top->bottom, x) copy of the device grid."""
        n, nz = self._last
        out = np.empty((nz, n, n), np.uint8)
        self._check(self._lib.dxv_grid_download(self._ctx, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def GridBits(self, out=None):
        """The grid as one bit per voxel, packed on the device (dxv_grid_download_packed): uint8
        [ceil(nz*N*N/8)], voxel 8j+i in bit i of byte j == np.packbits(Grid().ravel(),
        bitorder="little").  `out` may be any writable uint8 buffer of that size, e.g. pinned."""
        nbytes = self._lib.dxv_grid_packed_bytes(self._ctx)
        if out is None:
            out = np.empty(nbytes, np.uint8)
        ptr = out.ctypes.data_as(C.c_void_p) if isinstance(out, np.ndarray) else C.c_void_p(out.data_ptr())
        size = out.nbytes if isinstance(out, np.ndarray) else out.numel() * out.element_size()
        self._check(self._lib.dxv_grid_download_packed(self._ctx, ptr, size))
        return out

    def Texels(self):
        n, nz = self._last
        out = np.empty((nz, n, n), np.uint32)
        self._check(self._lib.dxv_texels_download(self._ctx, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def EnableTexels(self, on=True):
        self._check(self._lib.dxv_enable_texels(self._ctx, int(bool(on))))

    def CountSolid(self):
        v = C.c_uint64()
        self._check(self._lib.dxv_grid_count(self._ctx, C.byref(v)))
        return v.value

    def grid_device_ptr(self, writable=True):
        """Device pointer of the selected frame's grid.  writable=True (dxv_grid_device_ptr) tells the library that the caller
        may write through it at any later time: the frame's launches then clear the grid every time.  writable=False
        (dxv_grid_device_ptr_ro) is for consumers that only read."""
        if writable:
            return self._lib.dxv_grid_device_ptr(self._ctx)
        return self._lib.dxv_grid_device_ptr_ro(self._ctx)

    def grid_bytes(self):
        return self._lib.dxv_grid_bytes(self._ctx)

    # ---- plumbing ---------------------------------------------------------------------------
    def set_stream(self, hip_stream):
        self._check(self._lib.dxv_set_stream(self._ctx, C.c_void_p(hip_stream) if hip_stream else None))

    def set_option(self, key, value):
        self._check(self._lib.dxv_set_option(self._ctx, key.encode(), int(value)))

    def stats(self):
        s = Stats()
        self._check(self._lib.dxv_get_stats(self._ctx, C.byref(s)))
        return s.as_dict()

    def build_lists(self, parity=False, grid=0):
        """Build the candidate lists of the reference rule now (they then travel with scene_export); parity=True: the parity
        rule's row lists as well; grid: the grid size the scene will be launched at (the lists' map follows it)."""
        self._check(self._lib.dxv_build_lists_for_grid(self._ctx, int(grid)) if grid else self._lib.dxv_build_lists(self._ctx))
        if parity:
            self._check(self._lib.dxv_build_parity_lists(self._ctx))

    def scene_bytes(self):
        return self._lib.dxv_scene_bytes(self._ctx)

    def scene_export(self, device_ptr, nbytes):
        self._check(self._lib.dxv_scene_export(self._ctx, C.c_void_p(device_ptr), nbytes))

    def scene_import(self, device_ptr, nbytes):
        self._check(self._lib.dxv_scene_import(self._ctx, C.c_void_p(device_ptr), nbytes))

    def list_check(self, gridDim, z0=0, nz=None):
        """(accepted (ray, triangle) pairs, violations, [(voxel id, triangle slot), ...]) of dxv_debug_list_check over slices
        [z0, z0 + nz) (default: the whole grid)."""
        out = np.zeros(34, np.uint64)
        nz = gridDim - z0 if nz is None else nz
        self._check(self._lib.dxv_debug_list_check(self._ctx, int(gridDim), int(z0), int(nz), out.ctypes.data_as(C.c_void_p)))
        nv = int(min(out[1], 16))
        return int(out[0]), int(out[1]), [(int(out[2 + 2 * k]), int(out[3 + 2 * k])) for k in range(nv)]

    def class_check(self, gridDim, z0=0, nz=None):
        """(hits on classified triangles, disagreements with the predicate, all hits, [(voxel id, triangle slot), ...]) of
        dxv_debug_class_check."""
        out = np.zeros(34, np.uint64)
        nz = gridDim - z0 if nz is None else nz
        self._check(self._lib.dxv_debug_class_check(self._ctx, int(gridDim), int(z0), int(nz), out.ctypes.data_as(C.c_void_p)))
        nv = int(min(out[1], 15))
        return int(out[0]), int(out[1]), int(out[2]), [(int(out[3 + 2 * k]), int(out[4 + 2 * k])) for k in range(nv)]

    def plan_check(self):
        """dxv_debug_plan_check for the current frame's last launch (which went through a work queue): dict with live_voxels,
        live_bricks (bricks holding a live voxel, exact), queued_bricks, violations (live bricks that are not queued: must be 0),
        duplicates (bricks queued twice: must be 0) and the first violating brick words."""
        out = np.zeros(16, np.uint64)
        self._check(self._lib.dxv_debug_plan_check(self._ctx, out.ctypes.data_as(C.c_void_p)))
        return {"live_voxels": int(out[0]), "live_bricks": int(out[1]), "queued_bricks": int(out[2]), "violations": int(out[3]),
                "duplicates": int(out[4]), "first": [int(v) for v in out[5:5 + int(min(out[3], 11))]]}

    def division_check(self, n_first, n_last):
        """dxv_debug_division_check: (voxel origins checked, origins where a set-up word differs from the IEEE quotient's, first ids) over
        every even grid size in [n_first, n_last]."""
        out = np.zeros(8, np.uint64)
        self._check(self._lib.dxv_debug_division_check(self._ctx, int(n_first), int(n_last), out.ctypes.data_as(C.c_void_p)))
        return int(out[0]), int(out[1]), [int(v) for v in out[2:2 + int(min(out[1], 6))]]

    def far_check(self, gridDim, z0=0, nz=None, lists_mip=False):
        """dxv_debug_far_check: dict with bricks, dead_bricks (the brick test of the brick-box launches calls them dead), rays_walked
        (their rays, walked through the LBVH), violations (rays among them that hit something: must be 0) and the first voxel ids."""
        out = np.zeros(12, np.uint64)
        nz = gridDim - z0 if nz is None else nz
        self._check(self._lib.dxv_debug_far_check(self._ctx, int(gridDim), int(z0), int(nz), int(bool(lists_mip)), out.ctypes.data_as(C.c_void_p)))
        return {"bricks": int(out[0]), "dead_bricks": int(out[1]), "rays_walked": int(out[2]), "violations": int(out[3]),
                "first": [int(v) for v in out[4:4 + int(min(out[3], 8))]]}

    def trim(self):
        self._check(self._lib.dxv_trim(self._ctx))

    def debug(self, what):
        st = self.stats()
        T = st["num_tris"]
        shapes = {DBG_SORTED_KEYS: ((T,), np.uint64), DBG_NODES: ((st["num_nodes"], 16), np.uint32),
                  DBG_TRI_POS: ((T, 12), np.float32), DBG_TRI_NRM: ((T, 12), np.float32),
                  DBG_PARENTS: ((2 * T - 1,), np.uint32), DBG_NODES32: ((st["num_nodes"], 8), np.uint32),
                  DBG_NODES64: ((st["num_nodes"], 16), np.uint32),
                  DBG_LIST_CELLS: ((6 * st["list_res"] ** 2, 4), np.uint32), DBG_LIST_ENTRIES: ((st["list_entries"], 4), np.uint32),
                  DBG_LIST_MIP: ((sum(6 * (st["list_res"] >> l) ** 2 for l in range(max(st["list_res"], 1).bit_length())),), np.uint16)}
        shape, dt = shapes[what]
        out = np.empty(shape, dt)
        self._check(self._lib.dxv_debug_download(self._ctx, what, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out
