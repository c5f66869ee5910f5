"""Multi-GPU host logic: Z-slab partition of the grid and the one-off scene broadcast.

The path shards without any per-frame exchange: every voxel is an independent ray against a
read-only scene (Content/Shaders/DXRVoxelizer.hlsl:58-85, each thread writes only its own
texel, :84) and Z is the slowest grid axis (iz = dyz / N, :66), so rank r of W owns slices
[z0, z0 + nz) = one contiguous N*N*nz-byte block of the output.  The only collective is one
broadcast of the built scene blob (rank 0 builds the LBVH; RCCL over xGMI when the backend is
nccl), once per mesh.  One process per GPU.
"""
import numpy as np


def slab_range(N, rank, world):
    """Contiguous, near-equal split of N slices over `world` ranks: (z0, nz), nz may be 0."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(N), int(world))
    nz = base + (1 if rank < rem else 0)
    z0 = rank * base + min(rank, rem)
    return z0, nz


def interleaved_blocks(N, rank, world, block):
    """Load-balanced alternative (SURVEY section 8(e) caveat): blocks of `block` slices dealt
    round-robin; returns the list of (z0, nz) this rank owns. Still no collective."""
    out = []
    for b, z0 in enumerate(range(0, int(N), int(block))):
        if b % world == rank:
            out.append((z0, min(int(block), int(N) - z0)))
    return out


def interleaved_slices(N, rank, world, block):
    """Global slice index of every local slice of `rank` in the block-cyclic partition, ascending
    (what dxv_voxelize_interleaved computes)."""
    lz = np.arange(int(N) // int(world))
    return (lz // block * world + rank) * block + lz % block


def scatter_interleaved(parts, N, world, block):
    """Reassemble the full grid from per-rank interleaved grids [(rank, grid[N/world, N, N]), ...]."""
    full = np.empty((N, N, N), np.uint8)
    for rank, g in parts:
        full[interleaved_slices(N, rank, world, block)] = g
    return full


def broadcast_scene(engine, dist, device, src=0, parity_lists=False, info=None, grid=0):
    """Broadcast the built scene from rank `src` to every rank of the default process group.

    engine: object with scene_bytes() / scene_export(ptr, n) / scene_import(ptr, n) working on
    memory of `device` (dxrvoxelizer_amd.Voxelizer for 'cuda', a host stand-in in the gloo tests).
    dist: torch.distributed (initialised).  Returns the blob size in bytes; with info (a dict) also
    info["bytes"], info["broadcast_ms"] (this rank's wall time of the blob's broadcast, synchronised on both sides),
    info["checksum"] (wrapping 64-bit sum of the blob's 8-byte words on this rank) and info["checksums"] (every rank's, one
    8-byte all-gather): a rank whose blob differs from the source's raises before it imports anything.  Once per mesh."""
    import time

    import torch

    info = info if info is not None else {}
    rank = dist.get_rank()
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        if hasattr(engine, "build_lists"):
            # the candidate lists travel with the blob: the other ranks adopt them (parity_lists: the parity rule's row lists too)
            # (grid: the grid size of the launches to come -- the source builds the map they will want, once, for everybody)
            kw = {"grid": int(grid)} if grid else {}
            engine.build_lists(parity=True, **kw) if parity_lists else engine.build_lists(**kw)
        n[0] = engine.scene_bytes()
    dist.broadcast(n, src=src)
    nbytes = int(n.item())
    if nbytes <= 0:
        raise RuntimeError("broadcast_scene: source rank has no built scene")
    blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if rank == src:
        engine.scene_export(blob.data_ptr(), nbytes)
        if device != "cpu" and str(device) != "cpu":
            torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.broadcast(blob, src=src)
    if str(device) != "cpu":
        torch.cuda.synchronize()
    info["broadcast_ms"] = (time.perf_counter() - t0) * 1e3
    info["bytes"] = nbytes
    # what arrived is what was sent: wrapping sum of the blob's 64-bit words (its sections are 256-byte aligned), every rank's
    # against the source's
    words = blob[: nbytes - nbytes % 8].view(torch.int64)
    mine = words.sum().reshape(1) if words.numel() else torch.zeros(1, dtype=torch.int64, device=device)
    every = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    sums = [int(x.item()) & 0xFFFFFFFFFFFFFFFF for x in every]
    info["checksum"], info["checksums"] = sums[rank], sums
    if sums[rank] != sums[src]:
        raise RuntimeError(f"broadcast_scene: rank {rank} received a blob whose checksum {sums[rank]:#x} differs from the source's {sums[src]:#x}")
    if rank != src:
        engine.scene_import(blob.data_ptr(), nbytes)
    return nbytes


def prepare_share(engine, N, rank, world, zblock=0):
    """After Init / broadcast_scene, when the grid is known: every rank prepares the work queue of ITS share of the N^3 grid
    (dxv_prepare_launch*; the queue is a function of lists, grid and partition and does not travel with the blob).  zblock: the
    block-cyclic partition of bench.py (blocks of zblock slices dealt round-robin) when N divides, else contiguous slabs; 0: slabs.
    Returns "interleaved", "slab" or None (an empty slab)."""
    if zblock and N % (zblock * world) == 0:
        engine.PrepareLaunchInterleaved(N, rank, world, zblock)
        return "interleaved"
    z0, nz = slab_range(N, rank, world)
    if not nz:
        return None
    engine.PrepareLaunch(N, z0, nz)
    return "slab"


def gather_slabs(parts):
    """Concatenate per-rank slabs [(z0, grid[nz,N,N]), ...] into the full grid (host side)."""
    parts = sorted(parts, key=lambda p: p[0])
    return np.concatenate([g for _, g in parts if g.shape[0]], axis=0)


class _DeviceBuffer:
    """Zero-copy view of a device allocation for torch.as_tensor (__cuda_array_interface__)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_grid_tensor(vox, device):
    """The context's device-resident grid of the last Voxelize as a torch uint8 tensor (no copy).  torch tensors are writable
    (torch refuses a read-only __cuda_array_interface__), so the pointer is taken through dxv_grid_device_ptr: the library then
    knows the caller may write into the grid at any time and never relies on zeros an earlier launch left there (option plan = 1
    would; the default, plan = 2, clears the grid in every launch anyway)."""
    import torch

    return torch.as_tensor(_DeviceBuffer(vox.grid_device_ptr(writable=True), vox.grid_bytes()), device=device)


def allgather_grid(vox, dist, N, world, zblock, device):
    """Optional collective ("or one ncclAllGather of N^3/G bytes per rank if a
    device-resident full grid is wanted"): every rank ends up with the full [N, N, N] uint8 grid on
    its device.  Ranks must have called VoxelizeInterleaved(N, rank, world, zblock) before."""
    import torch

    mine = device_grid_tensor(vox, device)
    assert mine.numel() == N * N * (N // world)
    out = torch.empty(world * mine.numel(), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, mine)
    nblk = N // (zblock * world)
    # rank-major [W, nblk, zblock, N, N] -> z-major [nblk, W, zblock, N, N]
    return out.view(world, nblk, zblock, N, N).permute(1, 0, 2, 3, 4).reshape(N, N, N).contiguous()
