#!/bin/bash
# First GPU call of a round: re-establish hardware evidence on the current tree in ONE bounded gpurun call.
#   gpurun --timeout 3000 -- bash tools/round_open.sh [tag]        (tag defaults to r02; output in gpurun_out/<tag>/)
# Steps (each under its own `timeout`, failures are recorded and do not stop the rest):
#   1. python -m pytest tests -m gpu -x -q
#   2. bench.py: C2 random (headline line, with cpu_baseline), C3 streaming default, C3 streaming pipelined, C5, the two
#      secondary workloads of SURVEY 8(d) (50 % non-member probes; shuffled records) and the HBM copy/read/write probe
#   3. rocprofv3 --kernel-trace --stats of the C2 and C3 commands -> kernel summaries (tools/rocpd_summary.py)
# QUERY PATH ONLY (SURVEY 8 a-e).  Construction code and opt-in kernels run in a separate, later call: tools/round_next.sh.
# Copy what is to be judged from gpurun_out/<tag>/ into profiles/ afterwards.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
step() { echo "[$(date +%H:%M:%S)] $*" | tee -a "$O/log.txt"; }

step "1. gpu test suite"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$O/pytest_gpu.txt" 2>&1
echo "exit code $?" >> "$O/pytest_gpu.txt"
tail -3 "$O/pytest_gpu.txt" | tee -a "$O/log.txt"

step "2. bench lines"
timeout 900 python bench.py > "$O/bench_c2_random.json" 2> "$O/bench_c2_random.err" || step "C2 bench failed"
timeout 600 python bench.py --mode stream --no-cpu-baseline > "$O/bench_c3_stream.json" 2> "$O/bench_c3_stream.err" || step "C3 bench failed"
timeout 600 python bench.py --kind u --no-cpu-baseline > "$O/bench_c5_unpartitioned.json" 2> "$O/bench_c5_unpartitioned.err" || step "C5 bench failed"
timeout 600 python bench.py --nonmember-frac 0.5 --no-cpu-baseline > "$O/bench_c2_random_half_nonmembers.json" 2> "$O/bench_c2_random_half_nonmembers.err" || step "C2 non-member bench failed"
timeout 600 python bench.py --mode stream --shuffle-records --no-cpu-baseline > "$O/bench_c3_stream_shuffled.json" 2> "$O/bench_c3_stream_shuffled.err" || step "C3 shuffled bench failed"
timeout 300 python tools/hbm_copy_probe.py > "$O/hbm_copy_probe.json" 2> "$O/hbm_copy_probe.err" || step "HBM probe failed"
grep -ho '"value": [0-9.e+]*\|"kernel_ms": [0-9.]*' "$O"/bench_*.json | paste - - | tee -a "$O/log.txt"
cat "$O/hbm_copy_probe.json" | tee -a "$O/log.txt"

step "3. kernel traces"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_c2" -o c2 -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$O/prof_c2.json" 2> "$O/prof_c2.err"
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_c3" -o c3 -- python3 "$R/bench.py" --mode stream --steps 20 --warmup 3 --no-cpu-baseline > "$O/prof_c3.json" 2> "$O/prof_c3.err"
cd "$R"
for t in c2 c3; do
    db=$(find "$O/prof_$t" -name '*_results.db' | head -1)
    [ -n "$db" ] && python3 tools/rocpd_summary.py "$db" "$O/${t}_kernel_stats.txt" > /dev/null && head -4 "$O/${t}_kernel_stats.txt" | cut -c1-150 | tee -a "$O/log.txt"
    rm -rf "$O/prof_$t"   # the rocpd databases are large; the summaries are what is kept
done

step "done"
