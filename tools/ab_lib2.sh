#!/bin/bash
# A/B of two builds of the library on one box, any two files:   gpurun -- bash tools/ab_lib2.sh <tag> <libA> <libB> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; A=$2; B=$3; shift 3
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
: > "$O/ab.txt"
for rep in 1 2 3; do
  for lib in "$A" "$B"; do
    echo "== $lib: $*" | tee -a "$O/ab.txt"
    LPHQ_LIB_PATH=$R/$lib timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-probe "$@" 2>/dev/null | grep -o '"value": [0-9.e+]*\|"kernel_ms": [0-9.]*\|"parity_vs_oracle": [a-z]*' | paste - - - | tee -a "$O/ab.txt"
  done
done
