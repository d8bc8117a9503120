#!/bin/bash
# A/B of two builds of the library on read batches (tests/short_reads_rate.py): lphash_amd/liblphq_prev.so against liblphq.so.
#   gpurun --timeout 1200 -- bash tools/ab_reads.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-ab_reads}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
: > "$O/ab.txt"
for rep in 1 2; do
  for lib in liblphq_prev.so liblphq.so; do
    echo "== $lib" | tee -a "$O/ab.txt"
    LPHQ_LIB_PATH=$R/lphash_amd/$lib timeout 900 python tests/short_reads_rate.py 2>&1 | tail -12 | tee -a "$O/ab.txt"
  done
done
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a "$O/ab.txt"
