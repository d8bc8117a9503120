#!/bin/bash
# kernel trace of the index construction (GPU pilot search) for two settings of the tuning aids -> gpurun_out/r05_build_repair_<lookahead>_kernel_stats.txt
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT_LPH=$(mktemp /tmp/trace_build_XXXXXX.lph)
trap 'rm -f "$OUT_LPH"' EXIT
for cfg in "16384 16" "65536 64"; do set -- $cfg
export LPHQ_SEARCH_LOOKAHEAD=$1 LPHQ_SEARCH_STEPS=$2
d="$R/gpurun_out/trace_build_$1"
timeout 900 rocprofv3 --kernel-trace --stats -d "$d" -o t -- python3 $R/tools/search_sweep_child.py $R 63 20 128 20000 "$OUT_LPH" > /dev/null 2> $R/gpurun_out/trace_build_$1.err
db=$(find "$d" -name '*_results.db' | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_summary.py $db $R/gpurun_out/r05_build_repair_$1_kernel_stats.txt > /dev/null
rm -rf "$d"
echo "== lookahead $1 steps $2"; head -12 $R/gpurun_out/r05_build_repair_$1_kernel_stats.txt | cut -c1-150
done
