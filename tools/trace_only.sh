#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the five workloads (50 timed steps each, so that the first launches -- cold clocks and caches -- do not
# weigh on the average the way they do over 10): gpurun_out/<tag>_<wl>_kernel_stats.txt.   gpurun --timeout 1500 -- bash tools/trace_only.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
run() { # name, bench args...
  local name=$1; shift
  local d=$R/gpurun_out/trace_${TAG}_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d $d -o t -- python3 $R/bench.py "$@" --steps 50 --warmup 5 --no-cpu-baseline --no-hbm-probe --gpu-build > /dev/null 2> $R/gpurun_out/trace_${TAG}_$name.err
  db=$(find $d -name '*_results.db' | head -1)
  [ -n "$db" ] && python3 $R/tools/rocpd_summary.py $db $R/gpurun_out/${TAG}_${name}_kernel_stats.txt > /dev/null
  rm -rf $d
  head -3 $R/gpurun_out/${TAG}_${name}_kernel_stats.txt | cut -c1-170
}
run c2
run c3 --mode stream
run c5 --kind u
run c4r --k 63 --kmers 2000000000
run c4s --k 63 --kmers 2000000000 --mode stream
