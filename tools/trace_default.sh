#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command (headline + secondary list in one process), summarised per kernel.
#   gpurun --timeout 1200 -- bash tools/trace_default.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/trace_default_$TAG
timeout 1000 rocprofv3 --kernel-trace --stats -d $d -o t -- python3 $R/bench.py > $R/gpurun_out/${TAG}_default_bench_traced.json 2> $R/gpurun_out/${TAG}_default_bench_traced.err
db=$(find $d -name '*_results.db' | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_summary.py $db $R/gpurun_out/${TAG}_default_bench_kernel_stats.txt > /dev/null
rm -rf $d
head -14 $R/gpurun_out/${TAG}_default_bench_kernel_stats.txt | cut -c1-200
python3 -c "
import json
d=json.loads(open('$R/gpurun_out/${TAG}_default_bench_traced.json').read().strip().splitlines()[-1])
print('HIP-event kernel_ms: C2', d['kernel_ms'], [(e['name'], round(e['kernel_ms'],4)) for e in d['secondary']])"
