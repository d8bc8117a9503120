#!/bin/bash
# Counter passes for one bench command (each --pmc set in its own run, with --kernel-trace only, as the pool requires).
# usage: tools/pmc_passes.sh <tag> <bench args...>     -> gpurun_out/pmc_<tag>.txt (per-kernel averages; the rocpd databases are deleted)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_${TAG}.txt
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT
run() { # name counters...
  local name=$1; shift
  local d=$R/gpurun_out/pmc_${TAG}_${name}
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" -d $d -o pmc -- python3 $R/bench.py "${BENCH_ARGS[@]}" --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-probe --gpu-build > /dev/null 2> $R/gpurun_out/pmc_${TAG}_${name}.err
  db=$(find $d -name '*_results.db' | head -1)
  [ -n "$db" ] && python3 $R/tools/rocpd_pmc.py $db | grep -E "lphq::k_query" | cut -c1-170 >> $OUT
  rm -rf $d
}
BENCH_ARGS=("$@")
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS
# (round 5: a pass over the TCC_EA0_RDREQ_* / _DRAM_* counters aborted rocprofv3 with signal 6 on this pool -- not collected; those counters name the TARGET of a
#  request (local memory vs GMI / IO), not whether the Infinity Cache served it, profiles/r05_nt_stream.txt)
# kernel trace of the same command (durations for the rates)
d=$R/gpurun_out/trace_${TAG}
timeout 900 rocprofv3 --kernel-trace --stats -d $d -o t -- python3 $R/bench.py "${BENCH_ARGS[@]}" --steps 10 --warmup 2 --no-cpu-baseline --no-hbm-probe --gpu-build > /dev/null 2> $R/gpurun_out/trace_${TAG}.err
db=$(find $d -name '*_results.db' | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_summary.py $db $R/gpurun_out/${TAG}_kernel_stats.txt > /dev/null
rm -rf $d
cat $OUT
