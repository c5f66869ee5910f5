"""Guard against silent codegen regressions of the dominant kernels (found the hard way in round 1:
one extra line made hipcc spill 36 B/lane to meet the 64-VGPR bound and cost 14 % on MI355X).
Reads the compiler's own resource remarks recorded by dxrvoxelizer_amd/build.py."""
import os

from dxrvoxelizer_amd import build


def resources():
    usage = os.path.join(build.OBJDIR, "traverse.usage")
    if not os.path.exists(usage):
        build.build(force=True)
    return build.kernel_resources("traverse")


def test_default_kernels_do_not_spill_and_keep_full_occupancy(dxvlib):
    res = resources()
    # <4x4x4 brick, 20-entry column, reference rule, no texels, WALK 2 (wide) and 1 (binary postponed-leaf)>
    default_ref = [v for k, v in res.items() if "k_voxelizeINS_5BrickILi4ELi4ELi4EEELi20ELi0ELb0ELi2E" in k]
    binary_ref = [v for k, v in res.items() if "k_voxelizeINS_5BrickILi4ELi4ELi4EEELi20ELi0ELb0ELi1E" in k]
    default_par = [v for k, v in res.items() if "k_parity_rowsILi8ELi1ELb1E" in k]   # 512-voxel runs, one row per wave, four-box nodes
    block_par = [v for k, v in res.items() if "k_parity_rowsILi8ELi2ELb1E" in k]     # ... 2 x 2 rows per wave
    # WALK 4, no ablation: direction-space lists -- launched over the brick box (k_voxelize) and, the default, by persistent waves
    # through the work queue (k_voxelize_queue: the same brick body inside a loop)
    lists_box = [v for k, v in res.items() if "k_voxelizeINS_5BrickILi4ELi4ELi4EEELi16ELi0ELb0ELi4ELi0EEE" in k]
    lists_queue = [v for k, v in res.items() if "k_voxelize_queueILb0EEE" in k]
    lists_listed = [v for k, v in res.items() if "k_voxelize_listedILb0EEE" in k]       # the same body, one workgroup per queued brick
    # (round 3: the scan loop loads its four entries from one address with immediate offsets -- 70 registers, seven waves per
    # SIMD, and 7 - 17 % faster than the 62-register loop that computed four clamped addresses; held to 64 registers the same
    # loop spills 20 bytes and loses: profiles/r03/ab_scan_loop_offsets_old_new_new64.txt.  Round 4: the persistent form keeps the
    # seven waves only because its loop holds nothing in vector registers through the body and reads the launch's parameters
    # anew for every brick -- written the obvious way it took 80 registers, 8 bytes of scratch and six waves)
    assert len(lists_box) == 1 and len(lists_queue) == 1 and len(lists_listed) == 1
    for r in lists_box + lists_queue:
        assert r["scratch"] == 0 and r["vgprs"] <= 72 and r["occupancy"] >= 7 and r["lds"] == 16 * 64 * 4
    # (round 6: the hardware-dispatched form keeps nothing of the closest hit but t and the tagged slot -- 64 registers, EIGHT waves; with the
    # texel image the hit's V, W, det, index wait in four more words of the LDS column -- 72 registers, seven waves; both without scratch)
    assert lists_listed[0]["scratch"] == 0 and lists_listed[0]["vgprs"] <= 64 and lists_listed[0]["occupancy"] == 8 and lists_listed[0]["lds"] == 16 * 64 * 4
    listed_texels = [v for k, v in res.items() if "k_voxelize_listedILb1EEE" in k]
    assert len(listed_texels) == 1
    assert listed_texels[0]["scratch"] == 0 and listed_texels[0]["vgprs"] <= 72 and listed_texels[0]["occupancy"] >= 7 and listed_texels[0]["lds"] == 20 * 64 * 4
    assert len(default_ref) == 1 and len(binary_ref) == 1 and len(default_par) == 1 and len(block_par) == 1
    assert block_par[0]["scratch"] == 0 and block_par[0]["occupancy"] >= 6
    for r in (default_ref[0], binary_ref[0]):
        assert r["scratch"] == 0 and r["vgprs"] <= 64 and r["occupancy"] == 8 and r["lds"] == 20 * 64 * 4
    assert default_par[0]["scratch"] == 0 and default_par[0]["lds"] == 256
    # no variant of the voxelize kernels may use scratch memory
    for k, v in res.items():
        if "k_voxelize" in k or "k_parity_rows" in k:
            assert v["scratch"] == 0, k
