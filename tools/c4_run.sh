#!/bin/bash
# C4 on one GPU (BASELINE.json configs[3] per-GPU work): k=63 m=20, 2 B k-mers, 128-bit k-mers; random and streaming bench lines with the
# oracle parity leg, then FETCH / WRITE / L2 counter passes (one --pmc set per run).   gpurun --timeout 2400 -- bash tools/c4_run.sh [tag] [kmers]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
KM=${2:-2000000000}
O=$R/gpurun_out/${TAG}_c4
mkdir -p "$O"
cd "$R"
step() { echo "[$(date +%H:%M:%S)] $*" | tee -a "$O/log.txt"; }
free -g | head -2 | tee -a "$O/log.txt"; nproc | tee -a "$O/log.txt"
step "random bench (builds the index: GPU pilot search)"
LPHQ_BUILD_VERBOSE=1 timeout 1500 python bench.py --k 63 --kmers $KM --steps 10 --warmup 2 --gpu-build --b1-records 0 > "$O/bench_c4_random.json" 2> "$O/bench_c4_random.err" || step "random bench failed"
step "stream bench"
timeout 900 python bench.py --k 63 --kmers $KM --mode stream --steps 10 --warmup 2 --gpu-build --b1-records 0 > "$O/bench_c4_stream.json" 2> "$O/bench_c4_stream.err" || step "stream bench failed"
grep -ho '"value": [0-9.e+]*\|"kernel_ms": [0-9.]*\|"frac": [0-9.]*' "$O"/bench_c4_*.json | paste - - - - | tee -a "$O/log.txt"
cd /tmp && export TMPDIR=/tmp
pmc() { # name mode counters...
  local name=$1 mode=$2; shift 2
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d "$O/pmc_${mode}_${name}" -o pmc -- python3 "$R/bench.py" --k 63 --kmers $KM --mode $mode --steps 3 --warmup 1 --gpu-build --no-cpu-baseline --no-hbm-probe > /dev/null 2> "$O/pmc_${mode}_${name}.err"
  db=$(find "$O/pmc_${mode}_${name}" -name '*_results.db' | head -1)
  [ -n "$db" ] && python3 "$R/tools/rocpd_pmc.py" "$db" | grep -E "query_packed|tiles_stream" | cut -c1-110 | tee -a "$O/pmc_counters.txt"
  rm -rf "$O/pmc_${mode}_${name}"
}
for mode in random stream; do
  step "pmc $mode"
  pmc fetch $mode FETCH_SIZE
  pmc write $mode WRITE_SIZE
  pmc l2 $mode TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  pmc sq $mode SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
done
step "kernel trace"
for mode in random stream; do
  timeout 600 rocprofv3 --kernel-trace --stats -d "$O/trace_$mode" -o t -- python3 "$R/bench.py" --k 63 --kmers $KM --mode $mode --steps 5 --warmup 1 --gpu-build --no-cpu-baseline --no-hbm-probe > /dev/null 2> "$O/trace_$mode.err"
  db=$(find "$O/trace_$mode" -name '*_results.db' | head -1)
  [ -n "$db" ] && python3 "$R/tools/rocpd_summary.py" "$db" "$O/c4_${mode}_kernel_stats.txt" > /dev/null && head -4 "$O/c4_${mode}_kernel_stats.txt" | cut -c1-150 | tee -a "$O/log.txt"
  rm -rf "$O/trace_$mode"
done
step done
