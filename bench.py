#!/usr/bin/env python3
"""bench.py -- solid Mvoxels/s of the voxelize hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid 512] [--mesh torus1m]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Called as `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment, this process touches no GPU: it
starts the N ranks as a child `python -m torch.distributed.run` (127.0.0.1 rendezvous), waits and relays rank 0's line.

A step = one Voxelize pass (the reference's per-frame DispatchRays, Content/Voxelizer.cpp:366)
over the whole grid with the scene already resident in HBM: LBVH and candidate lists are built once in
Init like the reference's acceleration structure (Content/Voxelizer.cpp:73) and are reported separately.
Init is told the grid (the reference's GRID_SIZE is a compile-time constant its Init knows too, Content/Voxelizer.cpp:8), so
the work queue of the rank's share -- which bricks can hold a live ray: a pure function of lists, grid and partition -- is
Init-time structure like the lists (dxv_prepare_launch).  `value` is the step of such a scene: the grid CLEARED and every queued
brick written inside the step, one dispatch dealt out by the hardware; nothing of the output is carried from launch to launch.
The same steps with the queue built inside every launch (what a launch of a grid Init was not told does; round 5's `value`) are
config.unprepared_step, with queue and zeros kept (plan = 1) config.kept_step, and config.first_voxelize_after_init is the cold call.
With N > 1 the grid is Z-slab partitioned, one process per GPU; rank 0 builds the LBVH and the
scene blob is broadcast once over RCCL; there is no per-step collective.  Total work is fixed
(strong scaling).  ONE voxelization is in flight per GPU at every N (values at different N compare like for like);
config.frames_in_flight_2/3 are the figures with the reference's FrameCount = 3 grids used to overlap launches.
Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # ranks started by a launcher: dmabuf IPC (what RCCL's sharing needs on this pool)

HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def make_mesh(name):
    from dxrvoxelizer_amd import meshes
    import numpy as np
    if name == "torus1m":
        vb, ib = meshes.torus()                       # exactly 1,000,000 triangles
        return vb, ib, "torus-1M (R=0.6, r=0.3, 1000x500 quads)"
    if name == "soup10m":
        # (a minute of single-threaded generation: tools that build it again and again keep a copy in /tmp)
        cache = os.path.join(os.environ.get("DXV_MESH_CACHE", "/tmp/dxv_mesh_cache"), "soup10m")
        try:
            vb, ib = np.load(cache + "_vb.npy"), np.load(cache + "_ib.npy")
            if vb.shape != (30_000_000, 6) or ib.shape != (30_000_000,):
                raise ValueError
        except (OSError, ValueError):
            vb, ib = meshes.soup()
            try:
                os.makedirs(os.path.dirname(cache), exist_ok=True)
                np.save(cache + "_vb.npy", vb), np.save(cache + "_ib.npy", ib)
            except OSError:
                pass
        return vb, ib, "soup-10M (seed 0x5EED1234)"
    if name in ("bunny", "dragon"):
        d = np.load(os.path.join(ROOT, "tests", "golden", "meshes", name + ".npz"))
        return d["vb"], d["ib"], name
    if name == "dragon9":
        d = np.load(os.path.join(ROOT, "tests", "golden", "meshes", "dragon.npz"))
        vb, ib = meshes.trisect(d["vb"], d["ib"])
        return vb, ib, "dragon x9 (900,000 triangles)"
    if name == "bunny16":
        d = np.load(os.path.join(ROOT, "tests", "golden", "meshes", "bunny.npz"))
        vb, ib = meshes.midpoint_subdivide(*meshes.midpoint_subdivide(d["vb"], d["ib"]))
        return vb, ib, "bunny x16 (1,114,656 triangles)"
    raise SystemExit(f"unknown mesh {name}")


def algorithmic_bytes(N, nz, T, V):
    """occupancy store + every BVH node, index and vertex read once."""
    return N * N * nz * 1 + (2 * T - 1) * 32 + T * 12 + V * 24


def build_bytes(T, passes=3):
    """SURVEY.md section 8(d), B_build: three positions per triangle read, one 64-bit key written, `passes` radix passes of
    8 B read + 8 B written per key, every 32-byte node written once, two child boxes read per internal node."""
    return T * 36 + T * 8 + passes * 16 * T + (2 * T - 1) * 32 + (T - 1) * 64


def source_hash():
    """sha256 over the sources libdxv.so is built from (csrc/ + include/dxv.h): what ties a committed PMC figure
    (profiles/traffic.json) to the kernels this run executes."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "dxrvoxelizer_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "dxv.h"), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(vb, ib, N, mode, budget_s=15.0):  # (N > 1: a shorter sample, the other ranks wait for rank 0)
    """The oracle's scalar BVH voxelizer ('port': the reference has no CPU path) on a bounded
    sample of the same workload: evenly spaced Z slices, all host cores (OpenMP over rows)."""
    from oracle import orc
    import numpy as np
    scene = orc.Scene(vb, ib)
    cores = orc.usable_cores()                      # affinity mask capped by the cgroup CPU quota
    probe = sorted(set(int(z) for z in np.linspace(0, N - 1, 16).round()))
    t0 = time.perf_counter()
    orc.voxelize_slices(scene, N, probe, mode=mode, threads=cores)
    per_slice = (time.perf_counter() - t0) / len(probe)
    n = int(max(16, min(N, budget_s / max(per_slice, 1e-6))))
    zs = sorted(set(int(z) for z in np.linspace(0, N - 1, n).round()))
    t0 = time.perf_counter()
    orc.voxelize_slices(scene, N, zs, mode=mode, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": len(zs) * N * N / dt / 1e6, "unit": "Mvoxels/s", "cores": cores, "kind": "port",
            "sample": f"{len(zs)} evenly spaced Z slices of the {N}^3 grid ({len(zs) * N * N} voxels, {dt:.1f} s), "
                      f"oracle BVH traversal, OpenMP over rows, {cores} threads (cgroup quota / affinity)"}


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the ranks as a child process BEFORE anything here has
    touched the GPU (no torch import, no HIP call in this process), wait, relay rank 0's JSON line and exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        try:
            if isinstance(json.loads(ln), dict) and "metric" in ln:
                line = ln
                continue
        except ValueError:
            pass
        print(ln, file=sys.stderr)                         # anything else the ranks wrote to stdout
    if line is not None:
        print(line, flush=True)
    if r.returncode == 0 and line is None:
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        return 1
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)     # 0.63 s of timed steps on one GPU, 0.12 s at 8 ranks (a region the driver's sampler sees)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--mesh", default="torus1m")
    ap.add_argument("--mode", default="reference", choices=["reference", "parity"])
    ap.add_argument("--brick", type=int, default=-1)
    ap.add_argument("--stack", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary figures (frames in flight, tree walk, second rule, 256^3, bunny x16)")
    ap.add_argument("--interleave", action="store_true", help="use the block-cyclic partition call even on one GPU")
    ap.add_argument("--frames", type=int, default=0,
                    help="voxelizations in flight per GPU in the headline region (frames of ONE context, dxv_set_frame; the "
                         "reference keeps FrameCount = 3 grids in flight, Content/Voxelizer.h:24).  Default: 1 at every N (one discipline on "
                         "both ends of a scaling ratio); the figures with 2 and 3 in flight are reported beside it")
    ap.add_argument("--no-prepare", action="store_true", help="Init is not told the grid: every launch builds its work queue itself (round 5's headline)")
    ap.add_argument("--spin-ms", type=float, default=100.0, help="untimed launches for this long before the warm-up steps (GPU clocks out of idle); 0: none")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--same-device", action="store_true",
                    help="plumbing test on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing test without a GPU: ranks rendezvous, reduce and print the line, no voxelizer (needs --backend gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torch.distributed.run
    if args.dry_run:
        if use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if use_dist:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ranks = dist.get_world_size() if use_dist else 1
        if use_dist:                                      # (the same order as the real run: the group goes away first, then rank 0's CPU baseline
            dist.barrier()                                # and the line -- no rank waits in a collective while rank 0 computes)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "dry run (no GPU work)", "value": 0.0, "unit": "Mvoxels/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                              "config": {"rccl_ranks": ranks, "backend": args.backend, "rank_sum": float(t.item()),
                                         "process_group_alive_when_rank0_finishes": bool(use_dist and dist.is_initialized())}}), flush=True)
        return

    import dxrvoxelizer_amd as dxv
    from dxrvoxelizer_amd.slabs import broadcast_scene, slab_range
    from dxrvoxelizer_amd.slabs import prepare_share as prepare_share_of

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the voxelizer has no CPU path)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    N, mode = args.grid, (dxv.MODE_REFERENCE if args.mode == "reference" else dxv.MODE_PARITY)
    vox = dxv.Voxelizer(local_rank)
    stream = torch.cuda.Stream()                 # a real (non-null) stream: frame 0's kernels, torch events and
    torch.cuda.set_stream(stream)                # RCCL all share it
    vox.set_stream(stream.cuda_stream)
    if args.brick >= 0:
        vox.set_option("brick", args.brick)
    if args.stack >= 0:
        vox.set_option("stack", args.stack)

    vb = ib = None
    label = args.mesh
    bcast_ms, bcast = 0.0, {}
    warmup_ms = None
    if rank == 0:
        vb, ib, label = make_mesh(args.mesh)
        warmup_ms = vox.stats()["warmup_ms"]     # (dxv_create's one pass through every step on a four-triangle scene: the runtime's lazy parts)
        vox.InitFromArrays(vb, ib)               # upload + LBVH + candidate lists (Init work, not part of a step)
        cold_build_ms = vox.stats()["build_ms"]  # (the first build of a process behind that pass)
        cold_list_ms = vox.stats()["list_ms"]
        vox.InitFromArrays(vb, ib)               # ... the same again: the figure config.build_roofline is made from
        warm_list_ms = vox.stats()["list_ms"]
    if use_dist:
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        broadcast_scene(vox, dist, torch.device("cuda", local_rank), info=bcast)
        torch.cuda.synchronize()
        dist.barrier()
        bcast_ms = (time.perf_counter() - t0) * 1e3          # export + broadcast + checksum all-gather + import, barrier to barrier
    st0 = vox.stats()
    T, V = st0["num_tris"], st0["num_verts"]

    # N > 1: block-cyclic Z partition (blocks of 8 slices dealt round-robin): contiguous slabs leave
    # the GPUs that own empty space idle (profiles/r01: 2.5x at 8 slabs); still no collective.
    # (blocks of 8 slices; of 4 -- one layer of 4^3-voxel bricks -- from 8 ranks on: the finer deal balances the ranks better,
    # slowest rank 0.119 instead of 0.127 ms looped on one GPU: profiles/r03/rank_times_zblock.jsonl)
    zblock = 4 if world >= 8 else 8
    interleave = (world > 1 or args.interleave) and N % (zblock * world) == 0
    z0, nz = slab_range(N, rank, world)
    if interleave:
        nz = N // world

    launched = set()                             # grid sizes launched once before any warm-up (that launch allocates the frame's grid)
    untimed = [0]                                # launches made before the first warm-up step (allocations + clock spin-up)
    prepare_ms = {}

    def prepare_share(n):
        """What Init does when it is told the grid: the work queue of this rank's share of the n^3 grid, built now and kept with the
        scene (dxv_prepare_launch*).  After every Init / import, before the launches."""
        if args.no_prepare or mode != dxv.MODE_REFERENCE:
            return
        prepare_share_of(vox, n, rank, world, zblock if (world > 1 or args.interleave) else 0)
        prepare_ms[n] = vox.stats()["prepare_ms"]

    prepare_share(N)

    def timed_region(frames, steps, warmup, n=None, per_step=False, lib_events=None):
        """`steps` steps with `frames` voxelizations in flight (frames of the one context, taking the steps in turn),
        barrier + synchronize on both sides; (wall seconds, mean kernel ms, per-step ms).  A rank's share of the grid is a
        short launch whose tail -- its last long rays running alone -- does not shrink with it; the reference hides the same
        thing by keeping FrameCount = 3 grids in flight (Content/Voxelizer.h:24).  n: grid size (default: the headline's).
        per_step: an event after every step as well (per-step times; ~4 us of stream time each, so not in the headline region).
        With one voxelization in flight the library's own two events per launch are switched off for the region (option
        events: ~8 us per step, 6 % of a rank's step at 8 ranks): the region is bracketed by two events of its own.  With several
        in flight they are what says how long a launch took (lib_events=False: off all the same -- the headline region at N > 1,
        whose kernel time comes from the one-in-flight region behind it; the mean kernel ms returned is then the wall time per step)."""
        turn = [0]
        n = N if n is None else n
        inter = (world > 1 or args.interleave) and n % (zblock * world) == 0
        z0n, nzn = slab_range(n, rank, world)

        def step():
            f = turn[0] % frames
            turn[0] += 1
            if inter:
                vox.VoxelizeInterleaved(n, rank, world, zblock, mode, sync=False, frameIndex=f)
            elif nzn:
                vox.Voxelize(n, mode, z0n, nzn, sync=False, frameIndex=f)

        if n not in launched:                    # (one launch allocates the frame's grid; lists and queue exist since Init)
            launched.add(n)
            step()
            vox.SyncAll()
            # ... and the GPU out of its idle clocks: the same launch for ~0.1 s before anything is timed (a timed region of 20
            # steps is 16 ms: on a GPU that has just woken up it measured 4 % less than the same steps a second later)
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < args.spin_ms * 1e-3:
                for _ in range(8):
                    step()
                vox.SyncAll()
            untimed[0] += turn[0]
        for _ in range(max(warmup, frames)):     # every frame launches at least once before the clock starts
            step()
        vox.SyncAll()
        lib_ev = (frames != 1) if lib_events is None else bool(lib_events)
        vox.set_option("events", 1 if lib_ev else 0)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1 if per_step else 2)]
        for e in evs:                            # (torch creates the HIP event at its first record: not inside the region)
            e.record(stream)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        gc_was_on = gc.isenabled()
        gc.disable()                             # (a collector pause between the opening event and the first launch would be charged to the steps)
        t0 = time.perf_counter()
        evs[0].record(stream)
        for k in range(steps):
            step()
            if per_step:
                evs[k + 1].record(stream)        # (the launches queue back to back on this stream: event k+1 - event k = step k on the device)
        if not per_step:
            evs[1].record(stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0            # this rank's K steps; the closing barrier (tens of us of RCCL) is not part of any
        if gc_was_on:
            gc.enable()
        if use_dist:                             # rank's work: the maximum over ranks is taken by reduce_max below
            dist.barrier()
        vox.SyncAll()                            # deferred kernel status (stack overflow) is an error
        vox.set_option("events", 1)
        steps_ms = [evs[k].elapsed_time(evs[k + 1]) for k in range(steps)] if per_step else []
        if frames == 1:
            k_ms = evs[0].elapsed_time(evs[-1]) / max(steps, 1)      # avg launch duration on the kernel's stream
        elif not lib_ev:
            k_ms = dt / max(steps, 1) * 1e3
        else:                                    # overlapping launches: the library's own events around each frame's last launch
            ks = []
            for f in range(frames):
                vox.SetFrame(f)
                ks.append(vox.stats()["voxelize_ms"])
            k_ms = float(np.mean(ks))
        vox.SetFrame(0)
        return dt, k_ms, steps_ms

    def st_probe():
        vox.SyncAll()
        return vox.stats()

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather(x):
        """x of every rank, in rank order"""
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        if not use_dist:
            return [float(x)]
        out = torch.empty(world, dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(out, t)
        return [float(v) for v in out.tolist()]

    def stats_ms(xs):
        return {"median": float(np.median(xs)), "min": float(np.min(xs)), "max": float(np.max(xs))} if len(xs) else None

    # ONE voxelization in flight per GPU at every N: the same discipline on both ends of a scaling ratio (--frames overrides; the
    # figures with two and three in flight -- frames of the one context, the reference's FrameCount = 3 grids, Content/Voxelizer.h:24,
    # :110 -- are config.frames_in_flight_2/3).
    frames = max(1, min(args.frames if args.frames else 1, vox.FrameCount))
    # THE HEADLINE: the scene's work queue comes from Init (prepare_share above); every step clears the grid and writes every queued
    # brick, one dispatch dealt out by the hardware.  Nothing a step reads was left behind by another launch.
    vox.set_option("plan", 2)
    dt, step_ms_events, _ = timed_region(frames, args.steps, args.warmup, lib_events=False)
    dt_max = reduce_max(dt)
    _, k_one, per_step = timed_region(1, args.steps, 1, per_step=True)  # the same steps once more, one in flight, with an event behind every one: their spread
    if frames != 1:
        step_ms_events = k_one                   # (the dominant kernel's launch duration is a one-in-flight figure: overlapping launches share the GPU)
    st_run = st_probe()                          # of the timed rule (the extras below overwrite the launch fields)
    queued = bool(st_run["plan_bricks"]) and mode == dxv.MODE_REFERENCE
    prepared_run = queued and bool(st_run["plan_prepared"])
    # a launch that builds its queue: the queue build (the kernel in front of the brick kernel: queue + clear) on its own, a few launches with
    # the library's events
    pm = []
    if queued and not prepared_run:
        for _ in range(7):
            if interleave:
                vox.VoxelizeInterleaved(N, rank, world, zblock, mode)
            elif nz:
                vox.Voxelize(N, mode, z0, nz)
            pm.append(vox.stats()["plan_ms"])
    plan_ms = float(np.median(pm)) if pm else 0.0
    kernel_ms = step_ms_events - plan_ms         # the dominant kernel's average launch duration: the step between two events minus the kernel in front of it (none when prepared)
    # The same steps (i) with the queue built INSIDE every launch -- what a launch of a grid Init was not told does (option prepared = 0;
    # persistent waves: the launch does not know its size) -- and (ii) with the queue and the zeros of the bricks it does not run KEPT
    # from step to step (plan = 1: a static scene voxelized into the same frame again, the reference's own loop,
    # Content/Voxelizer.cpp:108-113; launched through the hardware's dispatcher once a sync has read the queue's lengths)
    kept = unprepared = None
    if queued:
        vox.set_option("prepared", 0)
        if prepared_run:
            dtu, ku, _ = timed_region(frames, args.steps, args.warmup, lib_events=False)
            dtu = reduce_max(dtu)
            stu = st_probe()
            pmu = []
            for _ in range(5):
                if interleave:
                    vox.VoxelizeInterleaved(N, rank, world, zblock, mode)
                elif nz:
                    vox.Voxelize(N, mode, z0, nz)
                pmu.append(vox.stats()["plan_ms"])
            unprepared = {"dt": dtu, "kernel_ms": ku, "waves": stu["plan_waves"], "queue_build_ms": float(np.median(pmu))}
        vox.set_option("plan", 1)
        dtk, kk, _ = timed_region(frames, args.steps, args.warmup, lib_events=False)
        dtk = reduce_max(dtk)
        kept = {"dt": dtk, "kernel_ms": kk, "waves": st_probe()["plan_waves"]}
        vox.set_option("plan", 2)
        vox.set_option("prepared", 1)
        timed_region(1, 2, 1)                                           # (back to the default for what follows)
    rank_kernel_ms = gather(step_ms_events)      # every rank's mean step between two events: an imbalance of the partition shows here
    rank_wall_ms = gather(dt / max(args.steps, 1) * 1e3)
    kmax = max(rank_kernel_ms)
    solid = vox.CountSolid() if nz else 0
    tot = torch.tensor([float(solid)], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)

    extras = {}
    if not args.no_extras:
        # the same steps with two and three voxelizations in flight per GPU (the reference keeps FrameCount = 3 grids in flight,
        # Content/Voxelizer.h:24: consecutive frames write different grids and overlap on the GPU); the headline keeps ONE in flight
        # at every N so that values at different N compare like for like
        for other in [f for f in (1, 2, 3) if f != frames]:
            dt2, k2, _ = timed_region(other, args.steps, 2)
            dt2 = reduce_max(dt2)
            extras[f"frames_in_flight_{other}"] = {"value": (N ** 3) * args.steps / dt2 / 1e6, "unit": "Mvoxels/s",
                                                   "ms_per_step": dt2 / args.steps * 1e3, "kernel_ms": reduce_max(k2)}
    if world == 1 and not args.no_extras:

        def median_ms(m, reps=5):
            vox.Voxelize(N, m)
            ts = []
            for _ in range(reps):
                vox.Voxelize(N, m)
                ts.append(vox.stats()["voxelize_ms"])
            return float(np.median(ts))

        if mode == dxv.MODE_REFERENCE:
            # The reference's own per-frame output: the R10G10B10A2_UNORM texel, float4(Normal, 1) where the ray is inside (hlsl:83-84,
            # Content/Voxelizer.cpp:65), written beside the occupancy byte.  Every solid voxel then needs its hit's interpolated normal
            # (the class bits cannot give it) and the step stores 4 more bytes per voxel -- the one variant of the path where HBM bytes
            # begin to matter.  The same steps as the headline's, texel image on.
            vox.EnableTexels(True)
            kt = min(args.steps, 200)
            dtt, ktm, _ = timed_region(1, kt, 3)
            stt = st_probe()
            vox.EnableTexels(False)
            tb = algorithmic_bytes(N, nz, T, V) + 4 * N * N * nz
            extras["texels"] = {"what": "the same steps with the reference's R10G10B10A2 texel image written beside the occupancy grid (dxv_enable_texels): "
                                        "N^3 x 4 B more stored per step, the interpolated normal fetched for every solid voxel",
                                "ms": dtt / kt * 1e3, "mvoxels_s": (N ** 3) * kt / dtt / 1e6, "steps": kt, "prepared": bool(stt["plan_prepared"]),
                                "stored_bytes": 5 * N * N * nz, "algorithmic_bytes": tb, "achieved_gbps": tb / (ktm * 1e-3) / 1e9,
                                "frac": tb / (ktm * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            timed_region(1, 2, 1)
        om = dxv.MODE_PARITY if mode == dxv.MODE_REFERENCE else dxv.MODE_REFERENCE
        other_ms = median_ms(om)                 # the second occupancy rule on the same scene, for the record
        extras["other_rule"] = {"mode": "parity" if args.mode == "reference" else "reference", "ms": other_ms,
                                "mvoxels_s": N ** 3 / other_ms / 1e3}
        if mode == dxv.MODE_REFERENCE:
            # the kernel north_star describes (LBVH walk, LDS stack, ballot vote) on the same scene
            vox.set_option("lists", 0)
            tw = median_ms(mode)
            vox.set_option("lists", 1)
            extras["tree_walk_ms"] = tw
            extras["tree_walk_mvoxels_s"] = N ** 3 / tw / 1e3
            if N == 512:
                # BASELINE.json's metric names 256^3 beside 512^3: the same scene and kernels at that grid
                def median_ms_at(n, m, reps=7):
                    vox.Voxelize(n, m)
                    ts = []
                    for _ in range(reps):
                        vox.Voxelize(n, m)
                        ts.append(vox.stats()["voxelize_ms"])
                    return float(np.median(ts))
                prepare_share(256)                                       # (Init told this grid too)
                l256 = median_ms_at(256, mode)
                p256 = vox.stats()["plan_prepared"]
                vox.set_option("prepared", 0)
                u256 = median_ms_at(256, mode)
                vox.set_option("plan", 1)
                k256 = median_ms_at(256, mode)
                vox.set_option("plan", 2)
                vox.set_option("prepared", 1)
                vox.set_option("lists", 0)
                t256 = median_ms_at(256, mode)
                vox.set_option("lists", 1)
                extras["grid_256"] = {"ms": l256, "mvoxels_s": 256 ** 3 / l256 / 1e3, "prepared": bool(p256),
                                      "unprepared_ms": u256, "unprepared_mvoxels_s": 256 ** 3 / u256 / 1e3,
                                      "kept_queue_ms": k256, "kept_queue_mvoxels_s": 256 ** 3 / k256 / 1e3,
                                      "tree_walk_ms": t256, "tree_walk_mvoxels_s": 256 ** 3 / t256 / 1e3,
                                      "queue_prepare_ms": prepare_ms.get(256),
                                      "note": "standalone launches, median of 7, the library's events around each: ms = the headline's step at this grid (queue from Init, "
                                              "grid cleared inside the launch); unprepared = queue built inside the launch (plan = 2); kept = plan 1"}
                vox.Voxelize(N, mode)                                    # (the 512^3 grid again for what follows)
            if args.mesh == "torus1m":
                # the other "1 M-triangle mesh" BASELINE.md names (no part of its grid is cleared by the partial launch)
                bvb, bib, blabel = make_mesh("bunny16")
                vox.InitFromArrays(bvb, bib)
                prepare_share(N)
                lm = median_ms(mode)
                vox.set_option("lists", 0)
                tm = median_ms(mode)
                vox.set_option("lists", 1)
                extras["bunny16"] = {"workload": f"{blabel}, {N}^3, reference predicate", "ms": lm, "mvoxels_s": N ** 3 / lm / 1e3,
                                     "tree_walk_ms": tm, "tree_walk_mvoxels_s": N ** 3 / tm / 1e3}

    if not args.no_extras and mode == dxv.MODE_REFERENCE and args.mesh == "torus1m":
        # BASELINE config 4 beside the headline at every N: dragon x9 at 1024^3, the same partition -- a rank's share is
        # eight times the headline's, so the launch tail weighs an eighth as much in the scaling it shows
        n4, k4 = 1024, max(5, min(args.steps, 10))
        if rank == 0:
            vb4, ib4, label4 = make_mesh("dragon9")
            vox.InitFromArrays(vb4, ib4)
        if use_dist:
            broadcast_scene(vox, dist, torch.device("cuda", local_rank), grid=n4)
        prepare_share(n4)
        dt4, k4ms, step4 = timed_region(1, k4, 3, n=n4, per_step=True)
        rk4 = gather(k4ms)
        dt4 = reduce_max(dt4)
        extras["config4_dragon9_1024"] = {"workload": "dragon x9 (900,000 triangles), 1024^3, reference predicate, same partition",
                                          "value": (n4 ** 3) * k4 / dt4 / 1e6, "unit": "Mvoxels/s", "steps": k4,
                                          "ms_per_step": dt4 / k4 * 1e3, "rank_kernel_ms": rk4, "step_ms": stats_ms(step4)}

    cold = None
    if rank == 0 and world == 1 and not args.no_extras and mode == dxv.MODE_REFERENCE:
        # The first call: one Init (upload, LBVH, candidate lists, the grid's work queue) and the scene's first three Voxelize calls, on a
        # context that has launched other things before (allocations of this size exist; the code is loaded)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        vox.InitFromArrays(vb, ib, gridDim=0 if args.no_prepare else N)
        torch.cuda.synchronize()
        init_ms = (time.perf_counter() - t0) * 1e3
        sti = vox.stats()
        calls = []
        for _ in range(3):
            t0 = time.perf_counter()
            vox.Voxelize(N, mode)
            calls.append({"wall_ms": (time.perf_counter() - t0) * 1e3, "events_ms": vox.stats()["voxelize_ms"]})
        cold = {"what": "InitFromArrays with the grid (upload + LBVH + lists + the grid's work queue, host wall clock around the call) and the scene's first "
                        "three synchronous Voxelize calls (wall clock, and the library's events around what the launch put into the stream); the process's "
                        "one warm-up pass (dxv_create of its first context: config.warmup_ms) is behind all of them",
                "init_wall_ms": init_ms, "init_parts_ms": {"upload": sti["upload_ms"], "lbvh": sti["build_ms"], "lists": sti["list_ms"],
                                                           "queue": 0.0 if args.no_prepare else sti["prepare_ms"]},
                "voxelize_calls": calls, "init_plus_first_voxelize_ms": init_ms + calls[0]["wall_ms"]}

    if rank == 0:
        value = (N ** 3) * args.steps / dt_max / 1e6
        bytes_launch = algorithmic_bytes(N, nz, T, V)
        achieved = bytes_launch / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes committed under profiles/ (collected and corrected as MI355X_MICROARCH.md
        # prescribes; tools/collect_evidence.py) -- only when they were taken on THESE sources: the file carries the hash of
        # csrc/ at the time, and a figure of other kernels is not reported as this run's
        traffic, traffic_note, l1 = None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{args.mesh}/{N}/{args.mode}/gpus{world}"    # (N > 1: one rank's launch, like `achieved`)
                ent = tj.get(key, {})
                if ent.get("source_hash") == source_hash():
                    traffic = ent.get("hbm_bytes_per_launch")
                    l1 = ent.get("l1")                               # the vector-L1 view of the same launch (tools/collect_evidence_r5.py)
                elif ent:
                    traffic_note = (f"profiles/traffic.json holds {ent.get('hbm_bytes_per_launch')} B for this workload, measured on sources "
                                    f"{ent.get('source_hash', '(round ' + str(ent.get('round')) + ', no hash)')}; this run's are {source_hash()}: not reported")
            except Exception:
                traffic = None
        bricks = st_run.get("plan_bricks", 0)
        scene_bytes = bytes_launch - N * N * nz                  # the read side of the algorithmic bytes
        stored = N * N * nz + 64 * bricks if queued else N * N * nz   # the grid's clear + the queued bricks' results
        kernel = (("k_voxelize_listed" if prepared_run else "k_voxelize_queue") if queued else "k_voxelize") if args.mode == "reference" else "k_parity_rows"
        unprepared_out = None
        if unprepared:
            unprepared_out = {"what": "the same steps with the work queue built INSIDE every launch (k_plan_bricks, which also clears the grid, then persistent "
                                      "waves: a launch that has just built its queue does not know its size) -- what a launch of a grid Init was not told "
                                      "does (option prepared = 0); round 5's headline",
                              "ms_per_step": unprepared["dt"] / args.steps * 1e3, "mvoxels_s": (N ** 3) * args.steps / unprepared["dt"] / 1e6,
                              "queue_build_ms": unprepared["queue_build_ms"], "persistent_waves": unprepared["waves"]}
        kept_out = None
        if kept:
            k_ms = kept["dt"] / args.steps * 1e3
            kept_out = {"what": "the same steps with the work queue and the zeros of the bricks it does not run KEPT from step to step (option plan = 1): a "
                                "static scene voxelized into the same frame again, the reference's own loop (Content/Voxelizer.cpp:108-113); the kept "
                                "queue's size is known to the host, so the hardware deals it out (k_voxelize_listed, one workgroup per brick)",
                        "ms_per_step": k_ms, "mvoxels_s": (N ** 3) * args.steps / kept["dt"] / 1e6, "kernel_ms": kept["kernel_ms"],
                        "workgroups_launched": kept["waves"], "stored_bytes_per_launch": 64 * bricks,
                        "roofline_frac_on_algorithmic_bytes": bytes_launch / (kept["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        out = {
            "metric": "solid Mvoxels/s at 512^3 (1M-tri mesh)" if (N == 512 and args.mesh == "torus1m")
                      else f"solid Mvoxels/s at {N}^3 ({label})",
            "value": value, "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label}, {T} triangles, {N}^3 grid, {args.mode} predicate, "
                                   f"one ray per voxel, " + (f"Z blocks of {zblock} slices dealt round-robin over {world} GPUs"
                                                           if interleave else f"Z-slab partition over {world} GPU(s)"),
                       "step": ("one Voxelize of a scene whose Init was told the grid: work queue from Init (dxv_prepare_launch: a pure function of lists, grid and "
                                "partition, like the lists of the scene), grid cleared and every queued brick written inside the step, one dispatch dealt out "
                                "by the hardware; nothing of the output carried from launch to launch" if prepared_run else
                                "one Voxelize with nothing carried from launch to launch (option plan = 2): the work queue built on "
                                "the device, the grid cleared and every voxel written inside the step" if queued else "one Voxelize"),
                       "grid": N, "triangles": T, "vertices": V, "mode": args.mode,
                       "slab_slices_rank0": nz, "frames_in_flight": frames,
                       "frames_in_flight_note": ("one voxelization in flight per GPU, at every N" if frames == 1 else
                                                 f"{frames} voxelizations in flight per GPU by --frames (frames of the one context: consecutive steps write different grids, "
                                                 "like the reference's FrameCount = 3); the default is one at every N"),
                       "solid_voxels": int(tot.item()),
                       "steps": args.steps, "warmup": args.warmup, "spin_ms": args.spin_ms,
                       "untimed_launches_before_warmup": untimed[0],     # one allocates the frame's grid and queue, the others spin the clocks up (--spin-ms)
                       "queue_build_ms": plan_ms, "queued_bricks": bricks,
                       "queue_prepare_ms": prepare_ms.get(N),               # in Init (dxv_prepare_launch): the queue's build on the device + one host round trip for its counts
                       "warmup_ms": warmup_ms,                              # dxv_create of the process's first context: one pass through every step on four triangles
                       "unprepared_step": unprepared_out,
                       "kept_step": kept_out,
                       "first_voxelize_after_init": cold,
                       "rccl_ranks": dist.get_world_size() if use_dist else 1, "backend": args.backend if use_dist else None,
                       "tree_height": st0["tree_height"], "stack_entries": st_run["stack_entries"],
                       "candidates": ({"structure": "direction-space lists", "texels_per_face_side": st_run["list_res"],
                                       "entries": st_run["list_entries"], "build_ms": warm_list_ms if rank == 0 else None,
                                       "build_ms_first_in_process": cold_list_ms if rank == 0 else None, "built": "in Init"}
                                      if st_run.get("list_entries") else {"structure": "LBVH walk"}),
                       "build_ms": st0["build_ms"], "build_ms_first_in_process": cold_build_ms,
                       "build_stages_ms": {k: st0[k] for k in ("prep_ms", "sort_ms", "hierarchy_ms", "refit_ms")},
                       "build_roofline": {"bound": "hbm", "bytes": build_bytes(T), "ms": st0["build_ms"],
                                          "achieved": build_bytes(T) / (st0["build_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                          "frac": build_bytes(T) / (st0["build_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                          "note": "LBVH build of this mesh (keys, radix sort in three passes of 10-bit digits, hierarchy, boxes, half-float node copy), "
                                                  "B_build of SURVEY.md 8(d)"},
                       "upload_ms": st0["upload_ms"], "scene_broadcast_ms": bcast_ms,
                       "scene_broadcast": ({"bytes": bcast.get("bytes"), "collective_ms": bcast.get("broadcast_ms"),
                                            "checksum": f"{bcast.get('checksum', 0):#018x}", "ranks_equal": len(set(bcast.get("checksums", [0]))) == 1}
                                           if bcast else None),
                       "step_ms_max_over_ranks": kmax, "rank_step_ms_events": rank_kernel_ms, "rank_ms_per_step": rank_wall_ms,
                       "rank_imbalance": max(rank_kernel_ms) / (sum(rank_kernel_ms) / len(rank_kernel_ms)) if min(rank_kernel_ms) > 0 else None,
                       "step_ms_rank0": stats_ms(per_step),
                       "work_queue": ({"queued_bricks": bricks, "workgroups_launched": st_run["plan_waves"],
                                       "launch": ("one workgroup per queued brick dealt out by the hardware; the bricks nobody runs are zeroed by workgroups of the same dispatch"
                                                  if prepared_run else "persistent waves taking bricks from the queue's heads (a launch that builds its queue does not know its size)"),
                                       "built": ("in Init (dxv_prepare_launch), once per (lists, grid, partition)" if prepared_run else
                                                 "on the device inside every step, by the kernel that also clears the grid (k_plan_bricks)")}
                                      if queued else None),
                       **extras},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, **({"traffic_note": traffic_note} if traffic_note else {}),
                         "stored_bytes_per_launch": stored,
                         "achieved_on_stored_bytes": (stored + scene_bytes) / (kernel_ms * 1e-3) / 1e9,
                         "kernel": kernel, "kernel_ms": kernel_ms, "step_ms_between_events": step_ms_events,
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "l1": l1,
                         "note": "algorithmic bytes by SURVEY.md 8(d) (grid + every tree node, index and vertex once) over the dominant kernel's average "
                                 "launch duration = the step between two events on the kernels' stream (a prepared launch is ONE dispatch: the clear's workgroups ride in it; an "
                                 "unprepared one: minus the queue build in front of it, config.queue_build_ms); the kernel is not bound by HBM bandwidth: it is a gather bound by the vector L1 / address "
                                 "units (roofline.l1: line accesses per clock and CU against the measured roof of tools/micro/l1_roof.hip; DESIGN.md section 4.2)"},
        }
    # The CPU baseline is rank 0's alone and runs AFTER the process group is gone: no rank sits in an RCCL barrier (where a watchdog
    # bites first on a new node) while rank 0 runs the oracle for seconds.
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    vox.close()
    if rank == 0:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(vb, ib, N, mode, budget_s=15.0 if world == 1 else 6.0)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
