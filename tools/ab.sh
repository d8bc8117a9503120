#!/bin/bash
# A/B of any number of settings on ONE box, interleaved repetitions, parity leg on (every line says whether the timed output equals the oracle's):
#   gpurun -- bash tools/ab.sh <tag> "<bench args>" "<setting>" "<setting>" ...
# A setting is a list of ENV=value words; `lib=rel/path.so` selects a library build (LPHQ_LIB_PATH; tools/build_variant.sh makes them from the
# current tree with some files taken from a git revision).  A bare `rel/path.so` is short for `lib=rel/path.so`.
#   REPS=3 (default)             repetitions
#   AB_CMD="python tests/short_reads_rate.py"   another command instead of bench.py (its last 12 lines are kept; "<bench args>" are appended)
#   AB_PARITY=0                  bench.py without the oracle leg (profiling-sized runs)
# This one script replaces the round-2/3 family (ab_lib.sh, ab_lib2.sh, ab_multi.sh, ab_env.sh, ab_reads.sh, ab3.sh).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; ARGS=$2; shift 2
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
PAR="--cpu-sample-log2 20"; [ "${AB_PARITY:-1}" = 0 ] && PAR="--no-cpu-baseline"
for rep in $(seq 1 ${REPS:-3}); do
  for setting in "$@"; do
    echo "== [$setting] $ARGS" | tee -a "$O/ab.txt"
    ( for kv in $setting; do case $kv in lib=*) export LPHQ_LIB_PATH=$R/${kv#lib=};; *.so) export LPHQ_LIB_PATH=$R/$kv;; *) export "$kv";; esac; done
      if [ -n "${AB_CMD:-}" ]; then timeout 900 $AB_CMD $ARGS 2>&1 | tail -12
      else timeout 900 python bench.py --steps 20 --warmup 3 $PAR --b1-records 0 --no-hbm-probe --no-secondary --gpu-build $ARGS 2>"$O/last.err" | grep -o '"value": [0-9.e+]*\|"kernel_ms": [0-9.]*\|"parity_vs_oracle": [a-z]*' | head -3 | paste - - -; fi ) | tee -a "$O/ab.txt"
    [ -s "$O/last.err" ] && grep -E "ERROR|Error" "$O/last.err" | head -2 | tee -a "$O/ab.txt"
  done
done
