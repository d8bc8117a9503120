#!/bin/bash
# Counter passes for one bench command (each --pmc set in its own run, with --kernel-trace only).
# usage: tools/pmc_passes.sh <tag> <bench args...>     -> gpurun_out/pmc_<tag>_<set>/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_${TAG}_${name} -o pmc -- python3 $R/bench.py "${BENCH_ARGS[@]}" --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_${name}.json 2> $R/gpurun_out/pmc_${TAG}_${name}.err
}
BENCH_ARGS=("$@")
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS
python3 $R/tools/rocpd_pmc.py $R/gpurun_out/pmc_${TAG}_*/pmc_results.db
