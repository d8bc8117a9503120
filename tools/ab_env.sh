#!/bin/bash
# A/B of environment settings (and optionally libraries) on one box, parity leg on:
#   gpurun -- bash tools/ab_env.sh <tag> "<bench args>" "<ENV=.. [LPHQ_LIB_PATH=rel/path]>" "<ENV=..>" ...     (REPS interleaved repetitions, default 3)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; ARGS=$2; shift 2
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
for rep in $(seq 1 ${REPS:-3}); do
  for setting in "$@"; do
    echo "== [$setting] $ARGS" | tee -a "$O/ab.txt"
    ( for kv in $setting; do case $kv in LPHQ_LIB_PATH=*) export LPHQ_LIB_PATH=$R/${kv#LPHQ_LIB_PATH=};; *) export "$kv";; esac; done
      timeout 900 python bench.py --steps 20 --warmup 3 --cpu-sample-log2 20 --b1-records 0 --no-hbm-probe --no-secondary --gpu-build $ARGS 2>"$O/last.err" | grep -o '"value": [0-9.e+]*\|"kernel_ms": [0-9.]*\|"parity_vs_oracle": [a-z]*' | head -3 | paste - - - ) | tee -a "$O/ab.txt"
    grep -E "ERROR" "$O/last.err" | head -2 | tee -a "$O/ab.txt"
  done
done
