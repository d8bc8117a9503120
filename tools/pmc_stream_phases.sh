#!/bin/bash
# Per-phase instruction counts of k_query_tiles_stream (LPHQ_PROFILE_STREAM_STOP = 1..5, 0 = all).
# Needs a PROFILING build of the library: make -C lphash_amd/csrc clean && make -C lphash_amd/csrc PROFILE=1 (the shipped
# liblphq.so has no stop switch); rebuild without PROFILE=1 afterwards.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for s in 1 2 3 4 5 0; do
  LPHQ_PROFILE_STREAM_STOP=$s rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $R/gpurun_out/pmc_phase_$s -o pmc -- python3 $R/bench.py --mode stream --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== stop_after $s"; python3 $R/tools/rocpd_pmc.py $R/gpurun_out/pmc_phase_$s/pmc_results.db | grep tiles_stream | cut -c1-70
done
