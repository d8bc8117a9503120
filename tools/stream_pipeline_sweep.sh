#!/bin/bash
# Round-2 measurement of the opt-in software-pipelined streaming kernel (lphash_amd/csrc/kernels_stream_pipe.cuh) against
# the default k_query_tiles_stream on C3 (all windows of the 100 M-k-mer index records, streaming).
#   gpurun --timeout 1500 -- bash tools/stream_pipeline_sweep.sh
# Writes gpurun_out/stream_pipe_sweep.txt (one bench JSON line per configuration; "kernel_ms" is the HIP-event time).
# The first run builds and caches the index (/tmp/lphq_bench_cache), the others reuse it.  Every run is bounded by `timeout`.
set -u
mkdir -p gpurun_out
OUT=gpurun_out/stream_pipe_sweep.txt
: > "$OUT"
run() {  # label, env assignments...
    local label=$1; shift
    echo "== $label" >> "$OUT"
    env "$@" timeout 600 python bench.py --mode stream --steps 20 --warmup 3 --no-cpu-baseline >> "$OUT" 2>&1 || echo "FAILED ($?)" >> "$OUT"
}
run "default kernel" LPHQ_STREAM_PIPELINE=0
for early in 1 0; do
    for wgs in 3 4 5 6 8; do   # residency is capped by registers (early: 5-6 / CU) and LDS (8 / CU)
        run "pipelined wgs_per_cu=$wgs early_block=$early" LPHQ_STREAM_PIPELINE=1 LPHQ_STREAM_PIPE_WGS=$wgs LPHQ_STREAM_PIPE_EARLY=$early
    done
done
grep -E "^==|kernel_ms" "$OUT" | sed -E 's/.*"value": ([0-9.e+]+).*"kernel_ms": ([0-9.]+).*/  value \1  kernel_ms \2/' | tee gpurun_out/stream_pipe_sweep_summary.txt
