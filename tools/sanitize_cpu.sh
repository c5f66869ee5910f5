#!/bin/bash
# AddressSanitizer + UBSan over everything that can run on the CPU (GPU ASan is not available on this
# pool): the product's __host__ __device__ code through tests/hostcheck, and the oracle.
set -e
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
cp tests/hostcheck/libhostcheck.so /tmp/libhostcheck_keep.so 2>/dev/null || true
cp oracle/liboracle.so /tmp/liboracle_keep.so 2>/dev/null || true
g++ -O1 -g -std=c++17 -fPIC -shared -fopenmp -ffp-contract=off -mavx2 -mfma -Wno-unknown-pragmas \
    -fsanitize=address,undefined -fno-sanitize-recover=undefined -o tests/hostcheck/libhostcheck.so tests/hostcheck/hostcheck.cpp
gcc -O1 -g -std=gnu11 -fPIC -fvisibility=hidden -ffp-contract=off -mavx2 -mfma -fopenmp \
    -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared -o oracle/liboracle.so oracle/dxv_oracle.c -lm
touch tests/hostcheck/libhostcheck.so oracle/liboracle.so
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 python -m pytest tests -x -q -m "not gpu" \
    --deselect tests/test_slabs_gloo.py::test_two_rank_gloo_slabs_equal_single || rc=$?
cp /tmp/libhostcheck_keep.so tests/hostcheck/libhostcheck.so 2>/dev/null || true
cp /tmp/liboracle_keep.so oracle/liboracle.so 2>/dev/null || true
touch tests/hostcheck/libhostcheck.so oracle/liboracle.so
# the multi-threaded OBJ ingest, built stand-alone: ASan+UBSan, then ThreadSanitizer
OBJ=/tmp/dxv_sanitize_torus.obj
python - <<'PY'
import sys; sys.path.insert(0, "tests")
import test_obj_ingest as T
T.write_torus_obj("/tmp/dxv_sanitize_torus.obj", 400, 200, True, crlf=True, relative=True)
PY
for san in address,undefined thread; do
    g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=$san -fno-sanitize-recover=all -o /tmp/dxv_obj_$$ \
        tests/cpp/obj_load_main.cpp dxrvoxelizer_amd/csrc/obj_ingest.cpp -lpthread
    for f in $OBJ tests/golden/obj/quad_poly_neg.obj tests/golden/obj/split_vn.obj; do
        DXV_OBJ_THREADS=8 /tmp/dxv_obj_$$ $f 2 > /dev/null || rc=$?
    done
    rm -f /tmp/dxv_obj_$$
done
rm -f $OBJ
exit ${rc:-0}
