#!/usr/bin/env python3
"""Static instruction counts of the library's kernels from the compiler's gfx950 assembly (no GPU needed).
    python tools/isa_counts.py [source.hip] [kernel name substring ...] [--asm FILE]   (default: traverse.hip, the two brick kernels)
Per kernel: vector ALU / scalar ALU / vector loads / vector stores + atomics / LDS instructions, and how many correctly rounded
divisions (v_div_fixup_f32) and square roots (v_sqrt_f32) it contains."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dxrvoxelizer_amd import build as B  # noqa: E402


def assembly(src):
    out = os.path.join(tempfile.mkdtemp(prefix="dxv_isa_"), "k.s")
    cmd = [B.hipcc()] + [f for f in B.FLAGS if not f.startswith("-W")] + ["-w", "--cuda-device-only", "-S", "-o", out, os.path.join(B.CSRC, src)]
    subprocess.check_call(cmd)
    return out


def counts(path, wanted):
    txt = open(path).read()
    res = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if wanted and not any(w in name for w in wanted):
            continue
        c = collections.Counter()
        for line in body.split("\n"):
            op = re.match(r"\s+([a-z][a-z_0-9]+)", line)
            if not op:
                continue
            op = op.group(1)
            if op.startswith("v_"):
                c["valu"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                c["vector_loads"] += 1
            elif op.startswith(("global_store", "global_atomic", "buffer_store", "flat_store", "scratch_store")):
                c["vector_stores"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            if op == "v_div_fixup_f32":
                c["divisions_f32"] += 1
            if op == "v_sqrt_f32":
                c["sqrt_f32"] += 1
            if op == "v_div_fixup_f64":
                c["divisions_f64"] += 1
        res[name] = dict(c)
    return res


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    asm = sys.argv[sys.argv.index("--asm") + 1] if "--asm" in sys.argv else None
    if asm:
        args = [a for a in args if a != asm]
    src = args[0] if args and args[0].endswith((".hip", ".cpp")) else "traverse.hip"
    wanted = [a for a in args if a != src] or ["k_voxelize_listedILb0", "k_voxelize_queueILb0"]
    import json
    for name, c in counts(asm or assembly(src), wanted).items():
        print(json.dumps({"kernel": name, **c}))
