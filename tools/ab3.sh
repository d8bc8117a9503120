#!/bin/bash
# A/B of several builds of the library on one box, with the parity leg on (every line says whether the timed output equals the oracle's):
#   gpurun -- bash tools/ab3.sh <tag> "<bench args>" <libA> <libB> ...      (library paths relative to the repository root)
# Every library runs the same bench.py command, REPS (default 3) interleaved repetitions; kernel_ms is the HIP-event time of the dominant kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; ARGS=$2; shift 2
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
for rep in $(seq 1 ${REPS:-3}); do
  for lib in "$@"; do
    echo "== $lib: $ARGS" | tee -a "$O/ab.txt"
    LPHQ_LIB_PATH=$R/$lib timeout 900 python bench.py --steps 20 --warmup 3 --cpu-sample-log2 20 --b1-records 0 --no-hbm-probe --no-secondary --gpu-build $ARGS 2>"$O/last.err" | grep -o '"value": [0-9.e+]*\|"kernel_ms": [0-9.]*\|"parity_vs_oracle": [a-z]*' | head -3 | paste - - - | tee -a "$O/ab.txt"
    [ -s "$O/last.err" ] && grep -E "ERROR|Error|error" "$O/last.err" | head -3 | tee -a "$O/ab.txt"
  done
done
