#!/bin/bash
# The round's closing evidence in one GPU call: counter passes + traces of the shipped kernels (pmc_traffic.json with the tree's sha, put in place ON
# THE BOX so that the bench line that follows carries `traffic`), then the -m gpu suite, smoke and the default bench line.
#   gpurun --timeout 4500 -- bash tools/round4_final.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
cd "$R"
bash tools/round4_counters.sh $TAG > gpurun_out/${TAG}_counters.log 2>&1
[ -s gpurun_out/pmc_traffic_${TAG}.json ] && cp gpurun_out/pmc_traffic_${TAG}.json profiles/pmc_traffic.json
bash tools/trace_only.sh $TAG > gpurun_out/${TAG}_trace.log 2>&1
bash tools/round3_open.sh ${TAG}_final
