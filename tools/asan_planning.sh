#!/bin/bash
# Host-side sanitizer pass (no GPU): builds the host sources of libqhip with AddressSanitizer + UBSan into /tmp/qhip_asan and runs
# everything that works without a device through it — the kernel catalog's code generation and the CPU tests of the plan-only API
# (typing, lowering, code generation, hiprtc compile for gfx950). GPU AddressSanitizer is not available on this pool.
#   tools/asan_planning.sh            (from the repository root, after `make -C qurious_amd/csrc`)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/qhip_asan
mkdir -p $OUT
cd $ROOT/qurious_amd/csrc
for f in ctx table expr codegen jit agg relops filter join nlj sort projection exchange plan_api; do
  g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../../include -c $f.cpp -o $OUT/$f.o &
done
wait
g++ -shared -fsanitize=address,undefined $OUT/*.o build/kernels.o build/kernels_rel.o -L/opt/rocm/lib -lamdhip64 -lhiprtc -ldl -Wl,-rpath,/opt/rocm/lib -o $OUT/libqhip.so
cd $ROOT
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python - <<PY
import sys
import qurious_amd._ffi as f
f.LIB_PATH = "$OUT/libqhip.so"
from qurious_amd import catalog
print("catalog entries generated:", len(catalog.catalog_sources()))
import pytest
sys.exit(pytest.main(["tests/test_cabi_and_planning.py", "tests/test_projection_exprs.py", "tests/test_time_keys.py", "tests/test_rust_shim.py", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"]))
PY
