#!/bin/bash
# A/B of the non-temporal hint on the k-mer / code streams (LPHQ_NT): gpurun --timeout 900 -- bash tools/ab_nt.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
O=$R/gpurun_out/${TAG}_nt
mkdir -p "$O"
cd "$R"
run() { local label=$1; shift; echo "== $label" | tee -a "$O/ab.txt"; env "$@" timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --gpu-build --no-hbm-probe ${BENCH_ARGS:-} 2>/dev/null | grep -o '"value": [0-9.e+]*\|"kernel_ms": [0-9.]*' | paste - - | tee -a "$O/ab.txt"; }
: > "$O/ab.txt"
for rep in 1 2; do
run "C2 random, plain" LPHQ_NT=0
run "C2 random, nt" LPHQ_NT=1
BENCH_ARGS="--mode stream" run "C3 stream, plain" LPHQ_NT=0
BENCH_ARGS="--mode stream" run "C3 stream, nt" LPHQ_NT=1
done
BENCH_ARGS="--kind u" run "C5, plain" LPHQ_NT=0
BENCH_ARGS="--kind u" run "C5, nt" LPHQ_NT=1
