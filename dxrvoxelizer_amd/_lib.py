"""ctypes loader for libdxv.so.  Fails loudly when the HIP library is missing: no fallback."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
API_VERSION = 6          # DXV_API_VERSION of include/dxv.h (tests/test_cabi.py compares the two)


class DxvError(RuntimeError):
    pass


def library_path():
    # DXV_LIBRARY: load another build of the same library (A/B timing of kernel variants in tools/)
    return os.environ.get("DXV_LIBRARY") or os.path.join(_HERE, "libdxv.so")


class Stats(C.Structure):
    _fields_ = [("num_tris", C.c_uint32), ("num_verts", C.c_uint32), ("num_nodes", C.c_uint32),
                ("tree_height", C.c_uint32), ("bound", C.c_float * 4), ("upload_ms", C.c_float),
                ("prep_ms", C.c_float), ("sort_ms", C.c_float), ("hierarchy_ms", C.c_float),
                ("refit_ms", C.c_float), ("build_ms", C.c_float), ("voxelize_ms", C.c_float),
                ("grid_dim", C.c_uint32), ("z0", C.c_uint32), ("nz", C.c_uint32),
                ("stack_entries", C.c_uint32), ("render_ms", C.c_float), ("redo_rays", C.c_uint32),
                ("row_block", C.c_uint32), ("tri_extent", C.c_float), ("list_entries", C.c_uint32), ("list_res", C.c_uint32),
                ("list_ms", C.c_float), ("plan_bricks", C.c_uint32), ("plan_waves", C.c_uint32), ("plan_ms", C.c_float),
                ("plan_prepared", C.c_uint32), ("prepare_ms", C.c_float), ("warmup_ms", C.c_float)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("bound", "reserved")}
        d["bound"] = list(self.bound)
        return d


# name -> (restype, argtypes); every symbol include/dxv.h declares
_F32P = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_U32P = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
SYMBOLS = {
    "dxv_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "dxv_warmup": (C.c_int, [C.c_int, C.POINTER(C.c_float)]),
    "dxv_destroy": (None, [C.c_void_p]),
    "dxv_last_error": (C.c_char_p, [C.c_void_p]),
    "dxv_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dxv_obj_load": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32),
                               C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32), _F32P]),
    "dxv_free": (None, [C.c_void_p]),
    "dxv_set_mesh": (C.c_int, [C.c_void_p, _F32P, C.c_uint32, _U32P, C.c_uint32]),
    "dxv_build": (C.c_int, [C.c_void_p]),
    "dxv_update_vertices": (C.c_int, [C.c_void_p, _F32P, C.c_uint32]),
    "dxv_update_vertices_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "dxv_refit": (C.c_int, [C.c_void_p]),
    "dxv_voxelize": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32]),
    "dxv_voxelize_async": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32]),
    "dxv_voxelize_interleaved": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]),
    "dxv_voxelize_interleaved_async": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]),
    "dxv_prepare_launch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "dxv_prepare_launch_interleaved": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "dxv_sync": (C.c_int, [C.c_void_p]),
    "dxv_set_frame": (C.c_int, [C.c_void_p, C.c_uint32]),
    "dxv_sync_all": (C.c_int, [C.c_void_p]),
    "dxv_grid_device_ptr": (C.c_void_p, [C.c_void_p]),
    "dxv_grid_device_ptr_ro": (C.c_void_p, [C.c_void_p]),
    "dxv_grid_bytes": (C.c_size_t, [C.c_void_p]),
    "dxv_grid_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dxv_grid_packed_bytes": (C.c_size_t, [C.c_void_p]),
    "dxv_grid_download_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dxv_grid_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "dxv_enable_texels": (C.c_int, [C.c_void_p, C.c_int]),
    "dxv_texels_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dxv_render": (C.c_int, [C.c_void_p, _F32P, _F32P, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "dxv_build_lists": (C.c_int, [C.c_void_p]),
    "dxv_build_lists_for_grid": (C.c_int, [C.c_void_p, C.c_uint32]),
    "dxv_build_parity_lists": (C.c_int, [C.c_void_p]),
    "dxv_scene_bytes": (C.c_size_t, [C.c_void_p]),
    "dxv_scene_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dxv_scene_import": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dxv_scene_checksum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "dxv_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "dxv_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "dxv_debug_download": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "dxv_debug_list_check": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "dxv_debug_class_check": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "dxv_debug_plan_check": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dxv_debug_division_check": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "dxv_debug_far_check": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]),
    "dxv_trim": (C.c_int, [C.c_void_p]),
    "dxv_api_version": (C.c_int, []),
}


def load_library():
    """Load libdxv.so and bind every C-ABI symbol.  Raises DxvError when it is not built."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise DxvError(f"{path} is missing: build it with `python -m dxrvoxelizer_amd.build` "
                           "(the voxelizer has no CPU fallback)")
        lib = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        # include/dxv.h: a binding compares the library's version with the header's it was written against before its first
        # call (an older build loaded through DXV_LIBRARY differs in options and in the meaning of the dxv_stats plan fields);
        # DXV_ALLOW_API_MISMATCH=1: the A/B tools that load an older build on purpose
        got = lib.dxv_api_version()
        if got != API_VERSION and os.environ.get("DXV_ALLOW_API_MISMATCH") != "1":
            raise DxvError(f"{path} reports DXV_API_VERSION {got}, this binding was written against {API_VERSION} "
                           "(rebuild with `python -m dxrvoxelizer_amd.build`, or set DXV_ALLOW_API_MISMATCH=1 for an A/B against an older build)")
        _LIB = lib
    return _LIB
