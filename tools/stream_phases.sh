#!/bin/bash
# Phase profile of the streaming kernel: cumulative kernel time (HIP events) and wave-level instruction counters of the variants that return
# after phase A, B, C+D, E (tools/stream_phase_libs.sh builds them here, before the call) and of the whole kernel.
#   gpurun --timeout 1500 -- bash tools/stream_phases.sh <tag> "<bench args, e.g. --k 63 --kmers 400000000>"
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; ARGS=$2
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
: > "$O/phases.txt"
for n in 1 2 3 4 0; do
  lib=$R/tools/phase_libs/liblphq_stop$n.so; [ $n = 0 ] && lib=$R/lphash_amd/liblphq.so
  echo "== stop_after $n" | tee -a "$O/phases.txt"
  LPHQ_LIB_PATH=$lib timeout 600 python bench.py --mode stream --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-probe --gpu-build $ARGS 2>/dev/null | grep -o '"kernel_ms": [0-9.]*' | tee -a "$O/phases.txt"
done
cd /tmp && export TMPDIR=/tmp
for n in 1 2 3 4 0; do
  lib=$R/tools/phase_libs/liblphq_stop$n.so; [ $n = 0 ] && lib=$R/lphash_amd/liblphq.so
  echo "== counters, stop_after $n" | tee -a "$O/phases.txt"
  export LPHQ_LIB_PATH=$lib
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
    d=$O/pmc_$n
    timeout 600 rocprofv3 --kernel-trace --pmc $set -d $d -o pmc -- python3 $R/bench.py --mode stream --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-probe --gpu-build $ARGS > /dev/null 2> "$O/pmc_$n.err"
    db=$(find $d -name '*_results.db' | head -1)
    [ -n "$db" ] && python3 $R/tools/rocpd_pmc.py $db | grep -E "tiles_stream" | cut -c1-75 | tee -a "$O/phases.txt"
    rm -rf $d
  done
done
