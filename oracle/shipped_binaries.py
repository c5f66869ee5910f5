#!/usr/bin/env python3
"""What the reference's SHIPPED BINARIES say about the hot path.  TEST INFRASTRUCTURE (build container only: reads /root/reference).

The reference holds no test vectors for the path, but it ships the path compiled: Bin/DXRVoxelizer.cso (the DXIL library of
Content/Shaders/DXRVoxelizer.hlsl: raygenMain / closestHitMain / missMain), Bin/PSRayCast.cso + Bin/VSScreenQuad.cso (the
grid's consumer) and Bin/DXRVoxelizer.exe (which contains XUSG::ObjLoader).  This script reads them -- the DXIL parts are LLVM
3.7 bitcode, disassembled with the image's llvm-dis after the one data-layout token today's LLVM refuses ("i8:32") is rewritten
in place; the exe through llvm-objdump -- and writes the FACTS a restatement can be checked against to
tests/golden/shipped_binaries.json (facts, not the disassembly: no text of the reference is stored):

  * which components of the ray origin are negated, how the direction is normalised, TMin / TMax / flags / mask of TraceRay,
    when and what raygenMain stores (hlsl:44-53, :58-85);
  * index and vertex fetch, the ORDER of the barycentric interpolation, the normalisation and the comparison of
    closestHitMain, with the threshold's bits (hlsl:90-119, :132-140);
  * that missMain is empty (hlsl:145-148);
  * the display pass's texture coordinate and loop counts (PSRayCast.hlsl:118-187);
  * the loader inside the exe negating z of every `v` and `vn` record (XUSGObjLoader.cpp:190-227), the exe's default mesh, and
    the link time of the exe and the DLLs -- which era of the source the binaries belong to (the x-mirror of the screenshots,
    oracle/visual_ab.py, is older than all of them).

usage:  python oracle/shipped_binaries.py      (tests/test_shipped_binaries.py checks the oracle against the committed JSON)
"""
import datetime
import json
import os
import re
import struct
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = "/root/reference/Bin"
LLVM = "/opt/rocm/lib/llvm/bin"


def _vbr6(text):
    """bits of `text` as an unabbreviated bitstream record writes it: every character a VBR6 value, low chunk first"""
    out = []
    for ch in text.encode():
        v, chunks = ch, []
        while True:
            c = v & 0x1F
            v >>= 5
            chunks.append(c | 0x20 if v else c)
            if not v:
                break
        for c in chunks:
            out += [(c >> k) & 1 for k in range(6)]
    return out


def dxil_ll(path):
    """LLVM IR text of the DXIL part of a DXBC container"""
    d = open(path, "rb").read()
    assert d[:4] == b"DXBC", path
    nparts = struct.unpack_from("<I", d, 28)[0]
    for o in struct.unpack_from("<%dI" % nparts, d, 32):
        if d[o:o + 4] != b"DXIL":
            continue
        body = d[o + 8:o + 8 + struct.unpack_from("<I", d, o + 4)[0]]
        version, _, magic, _, boff, bsz = struct.unpack_from("<II4sIII", body, 0)
        assert magic == b"DXIL"
        bc = bytearray(body[8 + boff:8 + boff + bsz])
        # the module's data layout says i8:32 (DXIL's own), which LLVM >= 15 rejects before it reads a single function; the
        # string is VBR6 characters at no byte boundary: rewrite "i8:32" -> "i8:08" bit by bit (same length)
        bits = [(bc[i >> 3] >> (i & 7)) & 1 for i in range(len(bc) * 8)]
        pat, rep = _vbr6("i8:32"), _vbr6("i8:08")
        hits = [i for i in range(len(bits) - len(pat)) if bits[i:i + len(pat)] == pat]
        assert len(hits) == 1, hits
        bits[hits[0]:hits[0] + len(pat)] = rep
        out = bytearray(len(bc))
        for i, b in enumerate(bits):
            if b:
                out[i >> 3] |= 1 << (i & 7)
        with tempfile.TemporaryDirectory() as tmp:
            open(os.path.join(tmp, "m.bc"), "wb").write(out)
            subprocess.check_call([os.path.join(LLVM, "llvm-dis"), os.path.join(tmp, "m.bc"), "-o", os.path.join(tmp, "m.ll")])
            return open(os.path.join(tmp, "m.ll")).read(), (version >> 4) & 0xF, version & 0xF, version >> 16
    raise ValueError("no DXIL part in " + path)


def function_body(ll, name):
    m = re.search(r"define void @\"?[^\n]*" + re.escape(name) + r"[^\n]*\{\n(.*?)\n\}", ll, re.S)
    assert m, name
    return [ln.strip() for ln in m.group(1).splitlines() if ln.strip()]


def f32_bits_of_ir_double(tok):
    """IR prints a float constant as the double with the same value: back to the float's bits"""
    v = struct.unpack("<d", struct.pack("<Q", int(tok, 16)))[0] if tok.startswith("0x") else float(tok)
    return struct.unpack("<I", struct.pack("<f", v))[0]


def voxelizer_facts():
    ll, major, minor, kind = dxil_ll(os.path.join(BIN, "DXRVoxelizer.cso"))
    rg, ch, ms = function_body(ll, "raygenMain"), function_body(ll, "closestHitMain"), function_body(ll, "missMain")
    f = {"shader_model": f"lib_{major}_{minor}", "compiler": re.search(r'!\{!"(dxc [^"]+)"\}', ll).group(1)}
    # ---- raygenMain
    text = "\n".join(rg)
    f["raygen"] = r = {}
    r["index_x"] = "DispatchRaysIndex.x" if "dispatchRaysIndex.i32(i32 145, i8 0)" in text else "?"
    r["index_y_z"] = "DispatchRaysIndex.y % DispatchRaysDimensions.x, DispatchRaysIndex.y / DispatchRaysDimensions.x" \
        if re.search(r"udiv i32 %DispatchRaysIndex\d*, %DispatchRaysDimensions", text) and re.search(r"urem i32 %DispatchRaysIndex\d*, %DispatchRaysDimensions", text) else "?"
    r["origin_steps"] = [op for op, pat in (("+0.5", r"fadd fast float %\.i\d+, 5\.0+e-01"), ("/dim", r"fdiv fast float"), ("*2", r"fmul fast float %\.i\d+, 2\.0+e\+00"),
                                              ("-1", r"fadd fast float %\.i\d+, -1\.0+e\+00")) if len(re.findall(pat, text)) == 3]
    negs = re.findall(r"(%\d+) = fsub fast float -0\.0+e\+00, (%\.i\d+)", text)
    trace = re.search(r"call void @dx\.op\.traceRay[^\n]*\(i32 157, %dx\.types\.Handle %\d+, i32 (-?\d+), i32 (-?\d+), i32 (\d+), i32 (\d+), i32 (\d+), "
                      r"float (\S+), float (\S+), float (\S+), float (\S+), float (\S+), float (\S+), float (\S+), float (\S+),", text)
    flags, mask, hit_off, stride, miss_idx, ox, oy, oz, tmin, dx, dy, dz, tmax = trace.groups()
    neg_names = {n for n, _ in negs}
    r["origin_negated_components"] = [c for c, v in (("x", ox), ("y", oy), ("z", oz)) if v in neg_names]
    r["trace_ray"] = {"flags": int(flags), "instance_mask": int(mask) & 0xFF, "hit_group_offset": int(hit_off), "geometry_stride": int(stride),
                      "miss_index": int(miss_idx), "tmin": float(tmin), "tmax": float(tmax)}
    rs = re.search(r"(%\w+) = call float @dx\.op\.dot3\.f32\(i32 55, float (\S+), float (\S+), float (\S+), float \2, float \3, float \4\)\n\s*(%\w+) = call float @dx\.op\.unary\.f32\(i32 25, float \1\)", text)
    r["direction"] = "origin * rsqrt(dot3(origin, origin))" if rs and len(re.findall(r"fmul fast float [^\n]*" + re.escape(rs.group(5)), text)) == 3 and \
        {rs.group(2), rs.group(3), rs.group(4)} == {ox, oy, oz} else "?"
    r["payload_initialised_to"] = "normal 0, isInside 0" if "store <3 x float> zeroinitializer" in text and "store i32 0" in text else "?"
    st = re.search(r"icmp eq i32 (%\d+), 0\n\s*br i1 %\d+, label %(\d+), label %(\d+)", text)
    store = re.search(r"call void @dx\.op\.textureStore\.f32\(i32 67, %dx\.types\.Handle %\d+, i32 (\S+), i32 (\S+), i32 (\S+), float %\d+, float %\d+, float %\d+, float (\S+), i8 15\)", text)
    r["stores"] = {"only_when": "payload.isInside != 0" if st else "?", "value": "float4(payload.normal, 1.0)" if store and float(store.group(4)) == 1.0 else "?",
                   "at": "(index.x, index.y % dim, index.y / dim)" if store else "?"}
    # ---- closestHitMain
    text = "\n".join(ch)
    f["closest_hit"] = c = {}
    c["indices"] = "g_indexBuffers[InstanceIndex][3 * PrimitiveIndex + {0, 1, 2}]" if re.search(r"mul i32 %PrimitiveIndex, 3", text) and len(re.findall(r"bufferLoad\.i32\(i32 68", text)) == 3 else "?"
    raw = re.findall(r"rawBufferLoad\.f32\(i32 139, %dx\.types\.Handle %\d+, i32 %\d+, i32 (\d+), i8 (\d+), i32 4\)", text)
    c["vertex_fetch"] = {"loads": len(raw), "byte_offset_in_vertex": sorted({int(o) for o, _ in raw}), "component_mask": sorted({int(m) for _, m in raw})}
    c["vertex_stride_bytes"] = int(re.search(r'!"g_vertexBuffers", i32 1, i32 0, i32 -1, i32 12, i32 0, (!\d+)', ll) and
                                   re.search(re.search(r'!"g_vertexBuffers", i32 1, i32 0, i32 -1, i32 12, i32 0, (!\d+)', ll).group(1) + r" = !\{i32 1, i32 (\d+)\}", ll).group(1))
    # the interpolation: (n1 - n0) * b.x + n0, then + (n2 - n0) * b.y
    step1 = re.search(r"(%\.i\d+) = fsub fast float (%\d+), (%\d+)\n(?:[^\n]*\n){2}\s*(%\.i\d+) = fmul fast float \1, (%\d+)\n(?:[^\n]*\n){2}\s*(%\.i\d+) = fadd fast float \4, \3", text)
    c["interpolation"] = "(n0 + (n1 - n0) * b.x) + (n2 - n0) * b.y" if step1 and re.search(r"fadd fast float " + re.escape(step1.group(6)) + r", %\.i\d+", text) else "?"
    rs = re.search(r"(%\d+) = call float @dx\.op\.dot3\.f32\(i32 55, float (\S+), float (\S+), float (\S+), float \2, float \3, float \4\)\n\s*(%\w+) = call float @dx\.op\.unary\.f32\(i32 25, float \1\)", text)
    c["normalise"] = "n * rsqrt(dot3(n, n))" if rs else "?"
    cmp_ = re.search(r"(%\d+) = call float @dx\.op\.dot3\.f32\(i32 55, float %\.i\d+, float %\.i\d+, float %\.i\d+, float %WorldRayDirection, float %WorldRayDirection\d+, float %WorldRayDirection\d+\)\n"
                     r"\s*%\d+ = fcmp fast (\w+) float \1, (\S+)", text)
    c["predicate"] = {"lhs": "dot3(normalised normal, WorldRayDirection)", "compare": cmp_.group(2), "threshold_f32_bits": f"{f32_bits_of_ir_double(cmp_.group(3)):#010x}"}
    c["payload_normal"] = "the normalised interpolated normal" if "store <3 x float>" in text else "?"
    f["miss"] = {"instructions": [ln for ln in ms]}
    return f


def display_facts():
    ll, major, minor, kind = dxil_ll(os.path.join(BIN, "PSRayCast.cso"))
    body = "\n".join(function_body(ll, "main"))
    tex = re.search(r"(%\d+) = fadd fast float %\d+, 5\.0+e-01\n\s*(%\d+) = fsub fast float 5\.0+e-01, %\d+\n\s*(%\d+) = fadd fast float %\d+, 5\.0+e-01\n\s*%\d+ = call %dx\.types\.ResRet\.f32 @dx\.op\.sampleLevel\.f32\([^\n]*float \1, float \2, float \3,", body)
    loops = sorted(int(n) for n in re.findall(r"icmp ult i32 %\d+, (\d+)", body))
    chan = set(re.findall(r"extractvalue %dx\.types\.ResRet\.f32 %\d+, (\d)", body))
    vs, *_ = dxil_ll(os.path.join(BIN, "VSScreenQuad.cso"))
    return {"shader_model": f"ps_{major}_{minor}", "texcoord": "(0.5, -0.5, 0.5) * pos + 0.5" if tex else "?", "loop_trip_counts": loops,
            "grid_channel_sampled": sorted(chan), "screen_quad_y": "1 - 2 * (id & 2)" if re.search(r"fsub fast float 1\.0+e\+00, %\d+", vs) else "?"}


def exe_facts():
    exe = os.path.join(BIN, "DXRVoxelizer.exe")
    d = open(exe, "rb").read()
    pe = struct.unpack_from("<I", d, 0x3C)[0]

    def link_time(path):
        b = open(path, "rb").read()
        o = struct.unpack_from("<I", b, 0x3C)[0]
        return datetime.datetime.fromtimestamp(struct.unpack_from("<I", b, o + 8)[0], datetime.timezone.utc).strftime("%Y-%m-%d")

    nsec, optsz = struct.unpack_from("<H", d, pe + 6)[0], struct.unpack_from("<H", d, pe + 20)[0]
    base = struct.unpack_from("<Q", d, pe + 24 + 24)[0]
    secs = [struct.unpack_from("<IIII", d, pe + 24 + optsz + 40 * i + 8) for i in range(nsec)]      # vsize, va, rsize, raw

    def va_of(off):
        for vsz, va, rsz, ro in secs:
            if ro <= off < ro + rsz:
                return base + va + off - ro
    fmt = va_of(d.index(b"%f %f %f\0"))
    asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--x86-asm-syntax=intel", exe], capture_output=True, text=True).stdout.splitlines()
    sites = [i for i, ln in enumerate(asm) if f"# {fmt:#x}" in ln]
    negated = []
    for i in sites:
        # behind the fscanf_s call: load the record's THIRD float, flip its sign bit (xor with a register holding 0x80000000), store it back
        win = " ".join(asm[i:i + 8])
        m = re.search(r"vmovss\s+xmm0, dword ptr \[(\w+)( - 0x4)?\].*vxorps\s+xmm1, xmm0, xmm\d+.*vmovss\s+dword ptr \[\1( - 0x4)?\], xmm1", win)
        negated.append(bool(m))
    return {"link_date_utc": {n: link_time(os.path.join(BIN, n)) for n in ("DXRVoxelizer.exe", "XUSG.dll", "XUSGRayTracing.dll")},
            "default_mesh": "Assets/bunny.obj" if b"Assets/bunny.obj\0" in d else "?",
            "loader_scanf_sites": len(sites), "loader_negates_third_float_after_each": negated,
            "reading": "both `%f %f %f` reads of ObjLoader::importGeometry (v and vn, XUSGObjLoader.cpp:190-213) are followed by a sign flip of z: "
                       "the shipped exe has forDX = true compiled in, like today's source"}


def main():
    out = {"source": "Bin/DXRVoxelizer.cso, Bin/PSRayCast.cso, Bin/VSScreenQuad.cso, Bin/DXRVoxelizer.exe of the reference, read by oracle/shipped_binaries.py",
           "voxelizer_dxil": voxelizer_facts(), "display_dxil": display_facts(), "exe": exe_facts()}
    path = os.path.join(ROOT, "tests", "golden", "shipped_binaries.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
