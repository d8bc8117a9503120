#!/bin/bash
# A/B of several builds of the library on one box:
#   gpurun -- bash tools/ab_multi.sh <tag> "<bench args>" <libA> <libB> ...      (library paths relative to the repository root)
# Every library runs the same bench.py command, three interleaved repetitions; kernel_ms is the HIP-event time of the dominant kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; ARGS=$2; shift 2
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
for rep in 1 2 3; do
  for lib in "$@"; do
    echo "== $lib: $ARGS" | tee -a "$O/ab.txt"
    LPHQ_LIB_PATH=$R/$lib timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-probe --gpu-build $ARGS 2>/dev/null | grep -o '"value": [0-9.e+]*\|"kernel_ms": [0-9.]*' | paste - - | tee -a "$O/ab.txt"
  done
done
