"""Build libdxv.so (HIP kernels + C-ABI) in-tree for gfx950.

    python -m dxrvoxelizer_amd.build [--force] [--save-temps] [--ablate]

hipcc cross-compiles without a GPU; the resulting dxrvoxelizer_amd/libdxv.so travels with the
repository snapshot to the GPU box.
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libdxv.so")

SOURCES = ["dxv_api.hip", "dxv_lists.hip", "dxv_frames.hip", "dxv_blob.hip", "dxv_debug.hip", "lbvh.hip", "radix_sort.hip", "traverse.hip", "raycast.hip", "dirmap.hip",
           "obj_ingest.cpp"]
HEADERS = ["dxv_device.h", "dxv_math.h", "dxv_trace.h", "dxv_types.h", "dxv_raycast.h", "dxv_dirmap.h", "dxv_ctx.h", "dxv_policy.h", os.path.join("..", "..", "include", "dxv.h")]

# -ffp-contract=off: the arithmetic of the path has a fixed operation order; the only fused
# operations are the explicit fmaf calls in dxv_math.h (hipcc contracts by default).
# Correctly rounded f32 divide/sqrt is hipcc's default and is stated explicitly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _compile(src, extra, objdir=None):
    objdir = objdir or OBJDIR
    obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
    cmd = [hipcc()] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
    if src.endswith(".cpp"):
        cmd = [c for c in cmd if not c.startswith("--offload-arch") and not c.startswith("-fhip")]
        cmd.insert(1, "-x"), cmd.insert(2, "c++")
    if src.endswith(".hip"):
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")   # VGPR / scratch / LDS per kernel -> build/*.usage
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("compile failed: %s\n%s%s" % (" ".join(cmd), r.stdout, r.stderr))
    if src.endswith(".hip"):
        with open(os.path.join(objdir, os.path.splitext(src)[0] + ".usage"), "w") as fh:
            fh.write(r.stderr)
        return obj, "\n".join(l for l in r.stderr.splitlines() if "remark:" not in l)
    return obj, r.stderr


def build(force=False, save_temps=False, verbose=False, ablate=False, defines=(), name=None):
    """ablate=True: a second library, libdxv_ablate.so, with the timing-only variants of the lists kernel (-DDXV_ABLATE:
    option `ablate`, wrong grids by design) for tools/ablate.py; the product library never contains them."""
    lib = os.path.join(HERE, "libdxv_ablate.so") if ablate else LIB
    objdir = OBJDIR + ("_ablate" if ablate else "")
    if name:                                                    # a diagnostic variant: libdxv_<name>.so built with extra -D flags
        lib, objdir = os.path.join(HERE, f"libdxv_{name}.so"), OBJDIR + "_" + name
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in HEADERS] + [__file__]
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= _newest(deps):
        return lib
    os.makedirs(objdir, exist_ok=True)
    extra = (["-save-temps=obj"] if save_temps else []) + (["-DDXV_ABLATE"] if ablate else []) + ["-D" + d for d in defines]
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda s: _compile(s, extra, objdir), SOURCES))
    for _, err in results:
        if verbose and err.strip():
            print(err, file=sys.stderr)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [o for o, _ in results]
    # cwd = the ignored build directory: the offload bundler drops its temp files where it runs
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=objdir)
    if r.returncode:
        raise RuntimeError("link failed: %s\n%s%s" % (" ".join(cmd), r.stdout, r.stderr))
    return lib


def kernel_resources(src="traverse"):
    """{kernel name: {"vgprs": n, "scratch": bytes per lane, "lds": bytes, "occupancy": waves/SIMD}}
    from the compiler's resource remarks of the last build."""
    import re
    path = os.path.join(OBJDIR, src + ".usage")
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True, ablate="--ablate" in sys.argv))
