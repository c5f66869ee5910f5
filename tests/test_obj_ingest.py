"""N2 mesh ingest on the fast path (csrc/obj_ingest.cpp): the mmap + multi-threaded parser must
return the bytes the sequential restatement of XUSGObjLoader.cpp (oracle) returns -- for every
number spelling, and for every worker count."""
import os

import numpy as np
import pytest


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def load_with_threads(path, n):
    import dxrvoxelizer_amd as dxv
    old = os.environ.get("DXV_OBJ_THREADS")
    os.environ["DXV_OBJ_THREADS"] = str(n)
    try:
        return dxv.obj_load(path)
    finally:
        if old is None:
            del os.environ["DXV_OBJ_THREADS"]
        else:
            os.environ["DXV_OBJ_THREADS"] = old


def spellings():
    """Decimal spellings that exercise the three number paths: <= 2^24 digit strings, <= 15
    digits through double, and strtof for the rest -- including decimals that sit next to a
    binary32 rounding midpoint, where double rounding would go wrong."""
    rng = np.random.default_rng(0xD0C5)
    out = ["0", "-0", "+1", ".5", "5.", "-.25", "1e0", "1E+2", "-1.5e-3", "16777216", "16777217", "0.000001",
           "1.17549435e-38", "1e-45", "3.4028235e38", "1e39", "123456789012345678901234567890", "0.1e-30",
           "1.0000000596046448", "1.00000005960464478", "1.00000005960464477", "0x1.8p1", "1e-11", "1e11", "1e22",
           "1e23", "8.5", "00012.500", "4e-320"]
    for _ in range(1500):                                  # what exporters write
        x = rng.uniform(-100, 100)
        out += ["%.6f" % x, "%.4f" % x, "%.9g" % x, "%.8e" % x, repr(float(np.float32(x)))]
    for _ in range(1500):                                  # neighbours of binary32 midpoints
        a = np.float32(rng.uniform(-1, 1) * 10.0 ** rng.integers(-6, 7))
        b = np.nextafter(a, np.float32(np.inf))
        mid = (float(a) + float(b)) / 2                    # exact in binary64
        s = "%.40e" % mid
        mant, ex = s.split("e")
        for nd in (9, 12, 15, 17, 30):
            lo = mant[: nd + 2 if mant[0] == "-" else nd + 1]
            out.append(lo + "e" + ex)                      # truncated: on or just inside the midpoint
            out.append(lo[:-1] + str(min(9, int(lo[-1]) + 1)) + "e" + ex)
    return out


def test_number_spellings_match_strtof(orc, dxvlib, tmp_path):
    sp = spellings()
    while len(sp) % 3:
        sp.append("1")
    path = str(tmp_path / "numbers.obj")
    nv = len(sp) // 3
    with open(path, "w") as f:
        for i in range(nv):
            f.write("v %s %s %s\n" % tuple(sp[3 * i: 3 * i + 3]))
        f.write("vn 0 0 1\n")
        for i in range(0, nv - 2, 3):
            f.write("f %d//1 %d//1 %d//1\n" % (i + 1, i + 2, i + 3))
    want = orc.obj_load(path)
    for threads in (1, 4):
        got = load_with_threads(path, threads)
        ok = want[0].view(np.uint32) == got[0].view(np.uint32)
        bad = np.argwhere(~ok)
        assert bad.size == 0, [(sp[3 * r + c] if c < 3 else "nrm") for r, c in bad[:5]]
        assert same_bits(want[1], got[1]) and same_bits(want[2], got[2])


def write_torus_obj(path, nu, nv, with_vn, crlf=False, relative=False):
    """A torus as QUADS (fan triangulated by the loader); optional vn, CRLF, negative indices."""
    nl = "\r\n" if crlf else "\n"
    u = np.arange(nu) * (2 * np.pi / nu)
    v = np.arange(nv) * (2 * np.pi / nv)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    x = (0.6 + 0.3 * np.cos(vv)) * np.cos(uu)
    y = 0.3 * np.sin(vv)
    z = (0.6 + 0.3 * np.cos(vv)) * np.sin(uu)
    n = np.stack([np.cos(vv) * np.cos(uu), np.sin(vv), np.cos(vv) * np.sin(uu)], -1).reshape(-1, 3)
    p = np.stack([x, y, z], -1).reshape(-1, 3)
    V = len(p)
    with open(path, "w", newline="") as f:
        f.write("# torus %d x %d%s" % (nu, nv, nl))
        f.write("o torus" + nl)
        for a in p:
            f.write("v %.6f %.6f %.6f%s" % (a[0], a[1], a[2], nl))
        if with_vn:
            for a in n:
                f.write("vn %.4f %.4f %.4f%s" % (a[0], a[1], a[2], nl))
        f.write("g quads" + nl + "s 1" + nl)
        for i in range(nu):
            for j in range(nv):
                q = [i * nv + j, ((i + 1) % nu) * nv + j, ((i + 1) % nu) * nv + (j + 1) % nv, i * nv + (j + 1) % nv]
                if relative:
                    q = [k - V for k in q]
                else:
                    q = [k + 1 for k in q]
                if with_vn:
                    f.write("f " + " ".join("%d//%d" % (k, k) for k in q) + nl)
                else:
                    f.write("f " + " ".join("%d" % k for k in q) + nl)
        f.write("# no newline at the end")
    return V


@pytest.mark.parametrize("with_vn,crlf,relative", [(False, False, False), (True, False, False),
                                                   (False, True, True), (True, True, True)])
def test_worker_count_does_not_change_the_mesh(orc, dxvlib, tmp_path, with_vn, crlf, relative):
    path = str(tmp_path / "torus.obj")
    V = write_torus_obj(path, 300, 150, with_vn, crlf, relative)
    assert os.path.getsize(path) > 2 << 20                 # several spans even at the 256 KiB floor
    want = orc.obj_load(path)
    assert want[0].shape == (V, 6) and want[1].size == 300 * 150 * 6
    for threads in (1, 2, 7, 16):
        got = load_with_threads(path, threads)
        assert same_bits(want[0], got[0]), threads
        assert same_bits(want[1], got[1]), threads
        assert same_bits(want[2], got[2]), threads
    got = __import__("dxrvoxelizer_amd").obj_load(path)    # default worker count
    assert same_bits(want[0], got[0]) and same_bits(want[1], got[1]) and same_bits(want[2], got[2])


def test_bad_files_fail_with_codes(dxvlib, tmp_path):
    import dxrvoxelizer_amd as dxv
    with pytest.raises(dxv.DxvError):
        dxv.obj_load(str(tmp_path / "missing.obj"))
    empty = tmp_path / "empty.obj"
    empty.write_text("")
    with pytest.raises(dxv.DxvError):
        dxv.obj_load(str(empty))
    oob = tmp_path / "oob.obj"
    oob.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n")
    with pytest.raises(dxv.DxvError):
        dxv.obj_load(str(oob))
    with pytest.raises(dxv.DxvError):
        dxv.obj_load(str(tmp_path))                         # a directory
