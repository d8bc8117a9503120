#!/bin/bash
# TEST INFRASTRUCTURE: run the emulator tests (the product's DEVICE sources compiled for the host) under UBSan and ASan.
#   bash tests/emu/sanitize.sh [pytest -k expression]
# LDS arrays are function-local statics here, so ASan's global redzones catch an out-of-range LDS index; host buffers are
# numpy allocations, so its heap redzones catch a kernel reading or writing outside what the caller passed.
# Last runs (round 2): 68 tests clean under both after the read-batch kernel and k_extract_runs (ASan: ~7 min because of the pilot-search cases);
# 64 (all but the pilot-search cases) clean under both on the final tree of the round (tail table, trimmed streaming kernel, minimizer scan from the left).
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
cd "$R/tests/emu"
SRC="emu_lookup.cpp emu_kernels.cpp emu_sinks.cpp emu_search.cpp emu_runtime.cpp"
g++ -O1 -g -std=c++17 -fPIC -I. -I../../lphash_amd/csrc -fsanitize=undefined -fno-sanitize-recover=undefined -fno-sanitize=alignment -shared $SRC -o /tmp/libemu_ubsan.so
g++ -O1 -g -std=c++17 -fPIC -I. -I../../lphash_amd/csrc -fsanitize=address -shared $SRC -o /tmp/libemu_asan.so
cd "$R"
K=${1:-"not nothing"}
LPHQ_EMU_LIB=/tmp/libemu_ubsan.so LD_PRELOAD=$(gcc -print-file-name=libubsan.so) UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_emu_cpu.py tests/test_runs_cpu.py -x -q -k "$K"
LPHQ_EMU_LIB=/tmp/libemu_asan.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 python -m pytest tests/test_emu_cpu.py tests/test_runs_cpu.py -x -q -k "$K"
