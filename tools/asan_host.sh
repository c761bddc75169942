#!/bin/bash
# CPU-only sanitizer pass over the host data layer (feeder, BGZF codec, packed panel): builds
# libgauss_host.so with ASan + UBSan into gauss_amd/lib/asan/ and runs the CPU feeder / farm tests on it.
# (GPU sanitizers are not available on the pool; the HIP side is covered by the parity tests.)
set -e
cd "$(dirname "$0")/.."
mkdir -p gauss_amd/lib/asan gpurun_out
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer \
    -o gauss_amd/lib/asan/libgauss_host.so gauss_amd/csrc/host/host_*.cpp gauss_amd/csrc/host/bgzf_io.cpp \
    gauss_amd/csrc/host/packed_panel.cpp -Lgauss_amd/lib -lgauss_hip -Wl,-rpath,"$(pwd)/gauss_amd/lib" -lz -lpthread -ldl
cat > gpurun_out/asan_run.py <<PY
import sys
sys.path.insert(0, "$(pwd)")
import gauss_amd.api as api
api.HOST_LIB_PATH = "$(pwd)/gauss_amd/lib/asan/libgauss_host.so"
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "not gpu", "tests/test_feeder.py", "tests/test_farm.py", "tests/test_farm_jepeg.py", "-p", "no:cacheprovider"]))
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 python gpurun_out/asan_run.py
