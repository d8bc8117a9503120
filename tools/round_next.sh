#!/bin/bash
# Second GPU call of a round, AFTER tools/round_open.sh is green: the SURVEY 8(f) "next" code on hardware, each step under its own
# short timeout so that a hang cannot hold the box.
#   gpurun --timeout 1800 -- bash tools/round_next.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
O=$R/gpurun_out/${TAG}_next
mkdir -p "$O"
cd "$R"
step() { echo "[$(date +%H:%M:%S)] $*" | tee -a "$O/log.txt"; }
step "1. next-tier gpu tests"
timeout 900 python -m pytest tests/test_zz_gpu_next.py -m gpu -x -q > "$O/pytest_gpu_next.txt" 2>&1
echo "exit code $?" >> "$O/pytest_gpu_next.txt"
tail -3 "$O/pytest_gpu_next.txt" | tee -a "$O/log.txt"
step "2. index build: GPU pilot search, 1 vs 16 rounds per host read-back; CPU search for the byte comparison"
for rps in 1 16; do
    LPHQ_BUILD_VERBOSE=1 LPHQ_SEARCH_ROUNDS_PER_SYNC=$rps timeout 600 python tools/gpu_build_time.py 1000 31 > "$O/build_rps$rps.txt" 2>&1 || step "build (rounds per sync $rps) failed"
    grep -E "gpu-search build|gpu pilot search" "$O/build_rps$rps.txt" | tee -a "$O/log.txt"
done
timeout 900 python tools/gpu_build_time.py 1000 31 --cpu > "$O/build_cpu_compare.txt" 2>&1 || step "cpu comparison build failed"
tail -1 "$O/build_cpu_compare.txt" | tee -a "$O/log.txt"
step "done"
