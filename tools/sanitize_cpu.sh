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
exit ${rc:-0}
