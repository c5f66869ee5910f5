"""N > 1 host path on CPU: world_size-2 gloo run of the slab partition and the one-off scene
broadcast (dxrvoxelizer_amd/slabs.py).  The per-rank engine is a host stand-in whose scene blob
is the mesh itself and whose voxelizer is the oracle; on the GPU box the same code moves the real
device blob over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from dxrvoxelizer_amd import meshes
from dxrvoxelizer_amd.slabs import gather_slabs, interleaved_blocks, slab_range


def test_slab_ranges_cover_the_grid():
    for N in (64, 100, 1024):
        for W in (1, 2, 3, 8):
            zs = [slab_range(N, r, W) for r in range(W)]
            assert zs[0][0] == 0 and sum(nz for _, nz in zs) == N
            for (a, na), (b, _) in zip(zs, zs[1:]):
                assert a + na == b
    assert slab_range(1024, 3, 8) == (384, 128)            # config 4: 128 slices per GPU
    blocks = [interleaved_blocks(64, r, 4, 8) for r in range(4)]
    assert sorted(sum(blocks, [])) == [(z, 8) for z in range(0, 64, 8)]
    from dxrvoxelizer_amd.slabs import interleaved_slices
    allz = np.concatenate([interleaved_slices(512, r, 8, 8) for r in range(8)])
    assert sorted(allz) == list(range(512)) and list(interleaved_slices(512, 3, 8, 8)[:9]) == [24, 25, 26, 27, 28, 29, 30, 31, 88]


def test_prepare_share_names_the_rank_s_own_partition():
    """slabs.prepare_share: every rank prepares the work queue of ITS share after the import -- the block-cyclic share when the grid
    divides into blocks of zblock slices for every rank, else its contiguous slab, nothing for an empty slab."""
    from dxrvoxelizer_amd.slabs import prepare_share

    class Recorder:
        def __init__(self):
            self.calls = []

        def PrepareLaunch(self, N, z0, nz):
            self.calls.append(("slab", N, z0, nz))

        def PrepareLaunchInterleaved(self, N, rank, world, zblock):
            self.calls.append(("interleaved", N, rank, world, zblock))

    for N, W, zb, want in ((512, 8, 4, "interleaved"), (512, 8, 0, "slab"), (100, 8, 4, "slab"), (512, 1, 0, "slab")):
        for r in range(W):
            e = Recorder()
            assert prepare_share(e, N, r, W, zb) == want
            assert e.calls == ([("interleaved", N, r, W, zb)] if want == "interleaved" else [("slab", N) + slab_range(N, r, W)])
    e = Recorder()
    assert prepare_share(e, 4, 7, 8) is None and e.calls == []          # (more ranks than slices: rank 7 owns nothing)


class HostEngine:
    """scene blob = [V, T, vb, ib] in host memory; voxelize through the oracle."""

    def __init__(self):
        self.blob = None
        self.prepared = None

    def PrepareLaunch(self, N, z0, nz):
        self.prepared = (N, z0, nz)

    def set_mesh(self, vb, ib):
        hdr = np.array([len(vb), len(ib) // 3], np.uint64)
        self.blob = np.concatenate([hdr.view(np.uint8), vb.reshape(-1).view(np.uint8), ib.view(np.uint8)])

    def scene_bytes(self):
        return self.blob.nbytes

    def scene_export(self, ptr, n):
        import ctypes as C
        C.memmove(ptr, self.blob.ctypes.data, n)

    def scene_import(self, ptr, n):
        import ctypes as C
        self.blob = np.empty(n, np.uint8)
        C.memmove(self.blob.ctypes.data, ptr, n)

    def voxelize(self, N, z0, nz):
        from oracle import orc
        V, T = (int(x) for x in self.blob[:16].view(np.uint64))
        vb = self.blob[16:16 + V * 24].view(np.float32).reshape(V, 6)
        ib = self.blob[16 + V * 24:].view(np.uint32)
        return orc.Scene(vb, ib).voxelize(N, z0=z0, nz=nz)


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from dxrvoxelizer_amd.slabs import broadcast_scene
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = HostEngine()
    if rank == 0:
        eng.set_mesh(*meshes.uv_sphere(32, 16))
    info = {}
    nbytes = broadcast_scene(eng, dist, "cpu", info=info)
    # the broadcast is timed and what arrived is checked against the source: one 8-byte all-gather per mesh
    assert info["bytes"] == nbytes and info["broadcast_ms"] >= 0.0 and len(info["checksums"]) == world
    assert len(set(info["checksums"])) == 1 and info["checksum"] == info["checksums"][0] != 0
    assert info["checksum"] == int(eng.blob[: nbytes - nbytes % 8].view(np.uint64).sum(dtype=np.uint64))
    z0, nz = slab_range(32, rank, world)
    from dxrvoxelizer_amd.slabs import prepare_share
    assert prepare_share(eng, 32, rank, world) == "slab" and eng.prepared == (32, z0, nz)   # (after the import, before the launches)
    g = eng.voxelize(32, z0, nz)
    np.save(os.path.join(outdir, f"slab{rank}.npy"), g)
    np.save(os.path.join(outdir, f"meta{rank}.npy"), np.array([z0, nz, nbytes]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_slabs_equal_single(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    parts = []
    for r in range(2):
        z0, nz, nbytes = np.load(tmp_path / f"meta{r}.npy")
        parts.append((int(z0), np.load(tmp_path / f"slab{r}.npy")))
        assert nbytes > 0
    eng = HostEngine()
    eng.set_mesh(*meshes.uv_sphere(32, 16))
    assert np.array_equal(gather_slabs(parts), eng.voxelize(32, 0, 32))
