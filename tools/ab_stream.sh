#!/bin/bash
# A/B of the streaming kernels on C3 (and the k=63 set): gpurun --timeout 900 -- bash tools/ab_stream.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
O=$R/gpurun_out/${TAG}_ab
mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
run() { local label=$1; shift; echo "== $label" | tee -a "$O/ab.txt"; env "$@" timeout 400 python bench.py --mode stream --steps 30 --warmup 5 --no-cpu-baseline --gpu-build --no-hbm-probe ${BENCH_ARGS:-} 2>/dev/null | grep -o '"value": [0-9.e+]*\|"kernel_ms": [0-9.]*' | paste - - | tee -a "$O/ab.txt"; }
: > "$O/ab.txt"
run "old kernel" LPHQ_STREAM_OLD=1
run "v4" LPHQ_STREAM_OLD=0
run "old kernel (again)" LPHQ_STREAM_OLD=1
run "v4 (again)" LPHQ_STREAM_OLD=0
run "v4 block 128" LPHQ_STREAM_OLD=0 LPHQ_STREAM_BLOCK=128
run "v4 block 512" LPHQ_STREAM_OLD=0 LPHQ_STREAM_BLOCK=512
BENCH_ARGS="--shuffle-records" run "v4 shuffled records" LPHQ_STREAM_OLD=0
BENCH_ARGS="--k 63 --kmers 20000000" run "old k63 20M" LPHQ_STREAM_OLD=1
BENCH_ARGS="--k 63 --kmers 20000000" run "v4 k63 20M" LPHQ_STREAM_OLD=0
