#!/bin/bash
# Counter passes + kernel traces of the current kernels on all five BASELINE workloads (one --pmc set per run, --kernel-trace only beside it).
#   gpurun --timeout 2700 -- bash tools/round3_counters.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
cd "$R"
bash tools/pmc_passes.sh ${TAG}_c2 > /dev/null 2>&1
bash tools/pmc_passes.sh ${TAG}_c3 --mode stream > /dev/null 2>&1
bash tools/pmc_passes.sh ${TAG}_c5 --kind u > /dev/null 2>&1
bash tools/c4_run.sh ${TAG} > /dev/null 2>&1
for t in c2 c3 c5; do echo "== $t"; cat gpurun_out/pmc_${TAG}_$t.txt; head -5 gpurun_out/${TAG}_${t}_kernel_stats.txt | cut -c1-160; done
cat gpurun_out/${TAG}_c4/log.txt
