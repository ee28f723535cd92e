#!/bin/bash
# AddressSanitizer + UBSan run of the host-side library (csrc/host_io.cpp: gather-inflate, sparse deflate writer) under the CPU
# tests that exercise it.  GPU sanitizers are not available on the pool; this is the CPU build only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build/asan
g++ -O1 -g -std=c++17 -fPIC -Wall -shared -fsanitize=address,undefined -fno-omit-frame-pointer \
    -o $ROOT/build/asan/libscs_io_asan.so $ROOT/sub-cortical_segmentation_amd/csrc/host_io.cpp
cd $ROOT
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
SCS_IO_LIB_PATH=$ROOT/build/asan/libscs_io_asan.so python -m pytest tests/test_host_io.py tests/test_base.py -x -q
