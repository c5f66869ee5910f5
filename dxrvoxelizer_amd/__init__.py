"""MI355X-native solid voxelizer: the ray-traced occupancy hot path of StarsX/DXRVoxelizer.

Python here is plumbing (ctypes over the C-ABI of libdxv.so, torch.distributed for the one-off
scene broadcast); the product is the HIP library built from dxrvoxelizer_amd/csrc.
"""
from ._lib import DxvError, load_library, library_path  # noqa: F401
from .voxelizer import MODE_PARITY, MODE_REFERENCE, Voxelizer, obj_load  # noqa: F401
