#!/bin/bash
# The round's closing evidence in one GPU call (round 5): counter passes of the shipped kernels on the five BASELINE workloads, kernel traces over 55 launches,
# pmc_traffic.json (with the traces' averages and the tree's sha) put in place ON THE BOX so that the bench lines that follow carry `traffic`, `bound` and
# `kernel_us_rocprof`; then the -m gpu suite, smoke, the default bench line, the same with --dist-single (RCCL path + the strong-scaling pass with the one rank)
# and a kernel trace of the default command.          gpurun --timeout 5400 -- bash tools/round5_final.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r05}
cd "$R"
step() { echo "[$(date +%H:%M:%S)] $*" | tee -a gpurun_out/${TAG}_final.log; }
if [ "${COUNTERS:-0}" = 1 ]; then  # ~15 GPU-minutes per workload (six rocprofv3 --pmc runs each): only when the device sources changed
  step counters
  bash tools/round4_counters.sh $TAG > gpurun_out/${TAG}_counters.log 2>&1
fi
step traces
bash tools/trace_only.sh $TAG > gpurun_out/${TAG}_trace.log 2>&1
if [ "${COUNTERS:-0}" = 1 ]; then
  python3 tools/pmc_to_json.py $TAG gpurun_out > gpurun_out/pmc_traffic_${TAG}.json 2>> gpurun_out/${TAG}_final.log
  python3 -c "import json; d=json.load(open('gpurun_out/pmc_traffic_${TAG}.json')); assert len(d) >= 6, d" && cp gpurun_out/pmc_traffic_${TAG}.json profiles/pmc_traffic.json
fi
step "pytest -m gpu"
python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/${TAG}_final_pytest_gpu.txt 2>&1; tail -3 gpurun_out/${TAG}_final_pytest_gpu.txt | tee -a gpurun_out/${TAG}_final.log
step smoke
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/${TAG}_final.log
step "default bench"
python bench.py > gpurun_out/${TAG}_final_bench_default.json 2> gpurun_out/${TAG}_final_bench_default.log; echo "rc=$?" | tee -a gpurun_out/${TAG}_final.log
step "default bench, --dist-single (RCCL with one rank; strong pass = the whole fixed job on one GPU)"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 python bench.py --dist-single > gpurun_out/${TAG}_final_dist_single_bench.json 2> gpurun_out/${TAG}_final_dist_single_bench.log; echo "rc=$?" | tee -a gpurun_out/${TAG}_final.log
step "repeat-rich index sets (not BASELINE configurations): the fallback path under load"
for a in "" "--mode stream" "--kind u" "--k 63 --kmers 400000000" "--k 63 --kmers 400000000 --mode stream"; do
  python bench.py --index-set diverged --gpu-build --no-secondary --b1-records 0 --cpu-sample-log2 20 $a 2>> gpurun_out/${TAG}_final_diverged.log | tee -a gpurun_out/${TAG}_final_diverged.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:150], '|', round(d['value']/1e9,2), 'G/s kernel', round(d['kernel_ms'],4), 'ms frac', round(d['roofline']['frac'],4), 'parity', d['parity_vs_oracle'])" | tee -a gpurun_out/${TAG}_final.log
done
step "trace of the default command"
bash tools/trace_default.sh $TAG > gpurun_out/${TAG}_trace_default.log 2>&1
step done
python3 - <<'PY' | tee -a gpurun_out/r05_final.log
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_final_*bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    rows = [("C2", d)] + [(e["name"], e) for e in d.get("secondary", [])]
    print(f)
    for n, e in rows:
        r = e["roofline"]
        print(f"  {n:16} {e['value']/1e9:8.2f} G/s kernel {e['kernel_ms']:.4f} ms frac {r['frac']:.4f} wall {r['frac_from_ms_per_step']:.4f} rocprof {r.get('frac_from_rocprof')} bound {r['bound']} parity {e['parity_vs_oracle']} pinned {e.get('index_digest_pinned')}"
              + (f" | strong {e['strong']['value']/1e9:.2f} G/s" if e.get("strong") else ""))
    print("  errors:", d.get("secondary_errors"), "wall", d.get("bench_wall_s"))
PY
