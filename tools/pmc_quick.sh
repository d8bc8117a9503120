#!/bin/bash
# One counter set of one bench command with a given library:  gpurun -- bash tools/pmc_quick.sh <tag> <lib> "<bench args>" [counters...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; LIB=$2; ARGS=$3; shift 3
CNT=${*:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY}
O=$R/gpurun_out/$TAG; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export LPHQ_LIB_PATH=$R/$LIB
d=$O/pmc
timeout 600 rocprofv3 --kernel-trace --pmc $CNT -d $d -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-probe --gpu-build $ARGS > /dev/null 2> "$O/pmc.err"
db=$(find $d -name '*_results.db' | head -1)
echo "== $LIB: $ARGS" | tee -a "$O/pmc.txt"
[ -n "$db" ] && python3 $R/tools/rocpd_pmc.py $db | grep -E "lphq::k_query" | cut -c1-100 | tee -a "$O/pmc.txt"
rm -rf $d
