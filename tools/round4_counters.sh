#!/bin/bash
# Round-4 evidence for profiles/: counter passes (one --pmc set per run, --kernel-trace only beside it) + kernel traces of the shipped kernels on the
# five BASELINE workloads, then profiles-ready files and a pmc_traffic.json whose entries carry the sha of the device sources they were taken on.
#   gpurun --timeout 3000 -- bash tools/round4_counters.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
cd "$R"
C4="--k 63 --kmers 2000000000"
bash tools/pmc_passes.sh ${TAG}_c2 > /dev/null 2>&1
bash tools/pmc_passes.sh ${TAG}_c3 --mode stream > /dev/null 2>&1
bash tools/pmc_passes.sh ${TAG}_c5 --kind u > /dev/null 2>&1
bash tools/pmc_passes.sh ${TAG}_c4r $C4 > /dev/null 2>&1
bash tools/pmc_passes.sh ${TAG}_c4s $C4 --mode stream > /dev/null 2>&1
for t in c2 c3 c5 c4r c4s; do echo "== $t"; cat gpurun_out/pmc_${TAG}_$t.txt; head -5 gpurun_out/${TAG}_${t}_kernel_stats.txt | cut -c1-160; done
python3 tools/pmc_to_json.py $TAG gpurun_out > gpurun_out/pmc_traffic_${TAG}.json
cat gpurun_out/pmc_traffic_${TAG}.json
