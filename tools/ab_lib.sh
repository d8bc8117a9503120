#!/bin/bash
# A/B of two builds of the library on one box: lphash_amd/liblphq_prev.so (built from an earlier commit, see tools/README.md) against
# lphash_amd/liblphq.so.   gpurun --timeout 1200 -- bash tools/ab_lib.sh <tag> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-ab}; shift
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
run() { local label=$1 lib=$2; shift 2; echo "== $label: $*" | tee -a "$O/ab.txt"; LPHQ_LIB_PATH=$lib timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --gpu-build --no-hbm-probe "$@" 2>/dev/null | grep -o '"value": [0-9.e+]*\|"kernel_ms": [0-9.]*' | paste - - | tee -a "$O/ab.txt"; }
: > "$O/ab.txt"
for rep in 1 2; do
  run prev "$R/lphash_amd/liblphq_prev.so" "$@"
  run new "$R/lphash_amd/liblphq.so" "$@"
done
