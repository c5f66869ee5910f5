"""bench.py as the driver invokes it: `python bench.py --gpus N` with no launcher around it must start its ranks itself
(a child `python -m torch.distributed.run`, 127.0.0.1 rendezvous) from a process that has not touched the GPU, and relay
rank 0's single JSON line.  Runs here on CPU: gloo backend, --dry-run (rendezvous, barrier, reductions, the line; no
voxelizer).  The GPU work of the ranks is covered by the -m gpu suite and by bench.py itself on the GPU box."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def run_bench(*argv, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None), e.pop("MASTER_PORT", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=e, timeout=600)


def test_self_launch_two_ranks_gloo_dry_run():
    r = run_bench("--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "4", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # ONE line, rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1
    assert out["config"]["rccl_ranks"] == 2                # dist.get_world_size() inside the ranks
    assert out["config"]["rank_sum"] == 3.0                # the all-reduce saw both ranks
    # rank 0's own work behind the steps (the CPU baseline in a real run) happens after the process group is gone: no rank sits
    # in a collective meanwhile
    assert out["config"]["process_group_alive_when_rank0_finishes"] is False
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("dist.destroy_process_group()", src.index("The CPU baseline is rank 0's alone")) < src.index('out["cpu_baseline"] = cpu_baseline(')


def test_single_rank_needs_no_launcher():
    r = run_bench("--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["config"]["rccl_ranks"] == 1


def test_failed_ranks_fail_the_launcher():
    # without --dry-run the ranks need a GPU: here they fail, and so must the parent (non-zero, no result line)
    r = run_bench("--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0")
    import torch
    if torch.cuda.is_available():
        return
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_parent_does_not_touch_torch_before_launching():
    # the launcher path runs before `import torch`: importing bench must not pull it in
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; "
                        "assert 'torch' not in sys.modules and 'dxrvoxelizer_amd' not in sys.modules" % ROOT],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


import pytest


@pytest.mark.gpu
def test_two_real_ranks_on_one_gpu_through_the_launcher(grids_json):
    """The whole multi-rank path with real engines: `python bench.py --gpus 2` starts two ranks (both on cuda:0, gloo:
    RCCL refuses two ranks per GPU), rank 0 builds the LBVH and the lists, the blob is broadcast, rank 1 imports it, each
    rank voxelizes its share of the block-cyclic Z partition; the summed solid count must be the oracle's."""
    r = run_bench("--gpus", "2", "--backend", "gloo", "--same-device", "--no-cpu-baseline", "--no-extras", "--mesh", "bunny",
                  "--grid", "128", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2 and out["config"]["slab_slices_rank0"] == 64
    assert out["config"]["solid_voxels"] == grids_json["bunny/128/reference"]["solid"]
    assert out["config"]["candidates"]["structure"] == "direction-space lists" and out["value"] > 0
    # the broadcast was timed and every rank's copy of the blob checked against the source's; the no-carried-state step is in the line
    b = out["config"]["scene_broadcast"]
    assert b["bytes"] > 0 and b["collective_ms"] > 0 and b["ranks_equal"] is True and int(b["checksum"], 16) != 0
    # `value` is the step of a scene whose Init was told the grid: each rank prepared its share's queue after the import, the step
    # clears the grid and the hardware deals the bricks out; the step that builds its queue itself and the kept-queue step stand beside it
    c = out["config"]
    assert c["step"].startswith("one Voxelize of a scene whose Init was told the grid") and c["queued_bricks"] > 0
    assert c["queue_build_ms"] == 0 and c["queue_prepare_ms"] > 0 and c["frames_in_flight"] == 1
    assert c["work_queue"]["queued_bricks"] == c["queued_bricks"] and out["roofline"]["kernel"] == "k_voxelize_listed"
    u = c["unprepared_step"]
    assert u["ms_per_step"] > 0 and u["queue_build_ms"] > 0 and u["persistent_waves"] % 8 == 0
    k = c["kept_step"]
    assert k["ms_per_step"] > 0 and k["stored_bytes_per_launch"] == 64 * c["queued_bricks"]
