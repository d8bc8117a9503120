#!/bin/bash
# four k-mers per lane with the first hop batched (tools/packed4_exp.hip) against the product kernel, times + L2 counters.
#   gpurun --timeout 1200 -- bash tools/packed4_run.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03_packed4}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
# the cached indexes (C2 and C5), built with the GPU pilot search
timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-probe --gpu-build > /dev/null 2> "$O/idx_p.err"
timeout 600 python bench.py --kind u --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-probe --gpu-build > /dev/null 2> "$O/idx_u.err"
P=/tmp/lphq_bench_cache/synth_seed1_p_k31_m20_n100000000_b64.lph
U=/tmp/lphq_bench_cache/synth_seed1_u_k31_m20_n100000000_b64.lph
{ echo "== partitioned (C2 index)"; timeout 300 tools/packed4_exp $P p; echo "== unpartitioned (C5 index)"; timeout 300 tools/packed4_exp $U u; } 2>&1 | tee "$O/times.txt"
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3; do
  d=$O/pmc_$v
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $d -o pmc -- $R/tools/packed4_exp $P p $v > /dev/null 2> "$O/pmc_$v.err"
  db=$(find $d -name '*_results.db' | head -1)
  echo "== counters, variant $v (0 = product kernel)" | tee -a "$O/times.txt"
  [ -n "$db" ] && python3 $R/tools/rocpd_pmc.py $db | grep -E "k_query_packed|k_packed4" | cut -c1-100 | tee -a "$O/times.txt"
  rm -rf $d
done
