#!/bin/bash
# Per-phase time and instruction counts of k_query_tiles_stream on C3, from a PROFILING build (make PROFILE=1: the stop switch does
# not exist in the shipped library).  gpurun --timeout 900 -- bash tools/stream_phases.sh [tag]
# The build is restored to the normal one at the end (the GPU box is ephemeral anyway).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
O=$R/gpurun_out/${TAG}_phases
mkdir -p "$O"
cd "$R"
make -C lphash_amd/csrc clean > /dev/null; make -C lphash_amd/csrc -j8 PROFILE=1 > "$O/build.txt" 2>&1 || { echo "profile build failed"; exit 1; }
EXTRA="${STREAM_ARGS:-}"
timeout 300 python bench.py --mode stream --steps 3 --warmup 1 --no-cpu-baseline --gpu-build --no-hbm-probe $EXTRA > /dev/null 2>&1   # builds and caches the index
: > "$O/phase_times.txt"
for s in 1 2 3 4 5 0; do
  ms=$(LPHQ_PROFILE_STREAM_STOP=$s timeout 300 python bench.py --mode stream --steps 20 --warmup 3 --no-cpu-baseline --gpu-build --no-hbm-probe $EXTRA 2>/dev/null | grep -o '"kernel_ms": [0-9.]*')
  echo "stop_after $s $ms" | tee -a "$O/phase_times.txt"
done
cd /tmp && export TMPDIR=/tmp
for s in 1 2 3 4 5 0; do
  LPHQ_PROFILE_STREAM_STOP=$s timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$O/pmc_$s" -o pmc -- python3 "$R/bench.py" --mode stream --steps 2 --warmup 1 --no-cpu-baseline --gpu-build --no-hbm-probe $EXTRA > /dev/null 2>&1
  echo "== stop_after $s" | tee -a "$O/phase_counters.txt"
  db=$(find "$O/pmc_$s" -name '*_results.db' | head -1)
  [ -n "$db" ] && python3 "$R/tools/rocpd_pmc.py" "$db" | grep tiles_stream | cut -c1-200 | tee -a "$O/phase_counters.txt"
  rm -rf "$O/pmc_$s"
done
cd "$R"
make -C lphash_amd/csrc clean > /dev/null; make -C lphash_amd/csrc -j8 > /dev/null 2>&1
