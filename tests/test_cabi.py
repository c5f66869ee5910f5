"""The drop-in boundary: libdxv.so loads, exports every symbol include/dxv.h declares, and fails
loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, has_gpu


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dxv.h")).read()
    return sorted(set(re.findall(r"DXV_API\s+[\w\s\*]+?\b(dxv_\w+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ("dxv_create", "dxv_destroy", "dxv_set_mesh", "dxv_build", "dxv_voxelize", "dxv_grid_download",
                 "dxv_obj_load", "dxv_scene_export", "dxv_scene_import", "dxv_last_error"):
        assert must in names
    assert len(names) >= 20


def test_library_exports_every_declared_symbol(dxvlib):
    from dxrvoxelizer_amd import _lib
    names = declared_symbols()
    for n in names:
        assert hasattr(dxvlib, n), f"libdxv.so does not export {n}"
    assert sorted(_lib.SYMBOLS) == names          # the Python binding covers the whole header


def test_every_option_the_library_takes_is_documented_in_the_header():
    """dxv_set_option's keys (csrc/dxv_api.hip) against the list in include/dxv.h: a knob nobody can read about is a bug of the boundary."""
    src = open(os.path.join(ROOT, "dxrvoxelizer_amd", "csrc", "dxv_api.hip")).read()
    keys = set(re.findall(r'strcmp\(key, "([a-z0-9]+)"\)', src))
    assert {"plan", "prepared", "prepclear", "lists", "coop", "farmap"} <= keys
    header = open(os.path.join(ROOT, "include", "dxv.h")).read()
    doc = header[header.index("Tuning knobs"):header.index("DXV_API int dxv_set_option")]
    documented = set(re.findall(r"^ \*\s+([a-z0-9]+)\s", doc, re.M)) | set(re.findall(r"\b([a-z0-9]+) 0\|", doc))
    missing = sorted(k for k in keys - {"ablate"} if k not in documented and not re.search(r"\b" + k + r"\b", doc))
    assert not missing, f"options without a line in include/dxv.h: {missing}"


def test_no_oracle_in_product():
    """The product never routes through the oracle: no source of the package mentions it and the
    library does not link it."""
    pkg = os.path.join(ROOT, "dxrvoxelizer_amd")
    for base, _, files in os.walk(pkg):
        if os.path.basename(base) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "liboracle" not in text and "import oracle" not in text and "from oracle" not in text, f
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "libdxv.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "amdhip64" in out


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure path")
def test_create_fails_loudly_without_gpu(dxvlib):
    import dxrvoxelizer_amd as dxv
    with pytest.raises(dxv.DxvError) as e:
        dxv.Voxelizer()
    assert "no CPU path" in str(e.value) or "no HIP device" in str(e.value)
    ctx = C.c_void_p()
    assert dxvlib.dxv_create(C.byref(ctx), 0) != 0 and not ctx.value


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from dxrvoxelizer_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "library_path", lambda: str(tmp_path / "libdxv.so"))
    with pytest.raises(_lib.DxvError):
        _lib.load_library()


def test_header_is_plain_c(tmp_path):
    """include/dxv.h must be consumable from C (cgo, JNI stubs, plain C hosts), not only C++."""
    import subprocess
    src = tmp_path / "use.c"
    src.write_text('#include "dxv.h"\nint main(void) { dxv_ctx* c = 0; dxv_stats s; (void)s; return dxv_create(&c, 0) == 0 ? (dxv_destroy(c), 0) : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "use.o")])
    for prog in ("voxelize_obj.cpp", "refit_loop.cpp"):
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", prog)])


def test_cpp_multi_gpu_host_compiles_and_links(tmp_path, dxvlib):
    """include/dxv_multi.hpp (one context per device, scene broadcast through the RCCL C API, slab / block-cyclic Voxelize, no
    Python) must compile as plain C++17 against the HIP and RCCL headers and link with libdxv.so + libamdhip64 + librccl;
    it runs in the -m gpu suite (tests/test_gpu_parity.py::test_cpp_multi_gpu_host)."""
    import subprocess
    rocm = "/opt/rocm"
    if not os.path.exists(os.path.join(rocm, "include", "rccl", "rccl.h")):
        pytest.skip("no RCCL headers on this machine")
    exe = tmp_path / "multi_gpu"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"),
                           os.path.join(ROOT, "tests", "cpp", "multi_gpu.cpp"), "-o", str(exe),
                           "-L" + os.path.join(ROOT, "dxrvoxelizer_amd"), "-l:libdxv.so", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lrccl",
                           "-Wl,-rpath," + os.path.join(ROOT, "dxrvoxelizer_amd"), "-Wl,-rpath," + os.path.join(rocm, "lib")])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr          # (no arguments: usage, no GPU touched)
