import sys, numpy as np
sys.path.insert(0, ".")
import dxrvoxelizer_amd as dxv
from dxrvoxelizer_amd.voxelizer import DBG_LIST_CELLS, DBG_LIST_ENTRIES
from bench import make_mesh
v = dxv.Voxelizer(0)
v.set_option("lists", 2)
for mesh, N in (("soup1m", 256), ("bunny", 256), ("torus1m", 256)):
    if mesh == "soup1m":
        from dxrvoxelizer_amd import meshes
        vb, ib = meshes.soup(1000000)
    else:
        vb, ib, _ = make_mesh(mesh)
    out = []
    for rep in range(3):
        v.InitFromArrays(vb, ib)
        v.Voxelize(N)
        st = v.stats()
        if rep == 2:
            v.Voxelize(N); st = v.stats()        # second launch: policy may move the map
        c, e = v.debug(DBG_LIST_CELLS).copy(), v.debug(DBG_LIST_ENTRIES).copy()
        out.append((st["list_res"], st["list_entries"], st["plan_bricks"], c, e))
        print(mesh, rep, st["list_res"], st["list_entries"], st["plan_bricks"])
    a, b = out[0], out[1]
    print(" same cells", np.array_equal(a[3], b[3]), "same entries", np.array_equal(a[4], b[4]))
    if not np.array_equal(a[3], b[3]):
        d = np.nonzero((a[3] != b[3]).any(1))[0]; print("  cells differ at", d[:5], a[3][d[:3]], b[3][d[:3]])
    if a[4].shape == b[4].shape and not np.array_equal(a[4], b[4]):
        d = np.nonzero((a[4] != b[4]).any(1))[0]; print("  entries differ", len(d), d[:5], a[4][d[:3]], b[4][d[:3]])
