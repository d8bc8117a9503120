#!/bin/bash
# The round's closing evidence in one GPU call (round 6): counter passes and kernel traces of the shipped kernels on the five BASELINE workloads AND the three
# sink workloads (C3-runs, C4-stream-runs, C4-stream-gather1; the device sources changed, so every entry of pmc_traffic.json is re-taken), pmc_traffic.json put in
# place ON THE BOX so that the bench lines that follow carry `traffic`; the ceiling probe (profiles/gather_ceiling.json); then the -m gpu suite, smoke, the default
# bench line and the same with --dist-single (RCCL path + the strong-scaling pass with the one rank).
#     gpurun --timeout 5400 -- bash tools/round6_final.sh [tag]          COUNTERS=0 skips the counter passes (~10 GPU-minutes per workload)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r06}
cd "$R"
mkdir -p gpurun_out
step() { echo "[$(date +%H:%M:%S)] $*" | tee -a gpurun_out/${TAG}_final.log; }
C4="--k 63 --kmers 2000000000"
WLS=("c2|" "c3|--mode stream" "c5|--kind u" "c4r|$C4" "c4s|$C4 --mode stream" "c3runs|--mode stream --sink runs" "c4sruns|$C4 --mode stream --sink runs" "c4sg1|$C4 --mode stream --sink gather1" "c2nm|--nonmember-frac 0.5" "c3sh|--mode stream --shuffle-records")
step "ceiling probe"
python bench.py --probe-ceiling > gpurun_out/${TAG}_probe_ceiling.json 2> gpurun_out/${TAG}_probe_ceiling.log && cp gpurun_out/gather_ceiling.json profiles/gather_ceiling.json
if [ "${COUNTERS:-1}" = 1 ]; then
  step counters
  for w in "${WLS[@]}"; do
    name=${w%%|*}; args=${w#*|}
    bash tools/pmc_passes.sh ${TAG}_$name $args > /dev/null 2>&1
    step "counters $name done"
  done
else
  step traces
  (cd /tmp && export TMPDIR=/tmp
   for w in "${WLS[@]}"; do
     name=${w%%|*}; args=${w#*|}
     d=$R/gpurun_out/trace_${TAG}_$name
     timeout 900 rocprofv3 --kernel-trace --stats -d $d -o t -- python3 $R/bench.py $args --steps 50 --warmup 5 --no-cpu-baseline --no-hbm-probe --gpu-build > /dev/null 2> $R/gpurun_out/trace_${TAG}_$name.err
     db=$(find $d -name '*_results.db' | head -1)
     [ -n "$db" ] && python3 $R/tools/rocpd_summary.py $db $R/gpurun_out/${TAG}_${name}_kernel_stats.txt > /dev/null
     rm -rf $d
   done)
fi
for w in "${WLS[@]}"; do name=${w%%|*}; echo "== $name"; cat gpurun_out/pmc_${TAG}_$name.txt 2>/dev/null; head -4 gpurun_out/${TAG}_${name}_kernel_stats.txt 2>/dev/null | cut -c1-170; done > gpurun_out/${TAG}_counters.log 2>&1
if [ "${COUNTERS:-1}" = 1 ]; then
  python3 tools/pmc_to_json.py $TAG gpurun_out > gpurun_out/pmc_traffic_${TAG}.json 2>> gpurun_out/${TAG}_final.log
  python3 -c "import json; d=json.load(open('gpurun_out/pmc_traffic_${TAG}.json')); assert len(d) >= 11, sorted(d)" && cp gpurun_out/pmc_traffic_${TAG}.json profiles/pmc_traffic.json
fi
step "pytest -m gpu"
python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/${TAG}_final_pytest_gpu.txt 2>&1; tail -3 gpurun_out/${TAG}_final_pytest_gpu.txt | tee -a gpurun_out/${TAG}_final.log
step smoke
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/${TAG}_final.log
step "default bench"
python bench.py > gpurun_out/${TAG}_final_bench_default.json 2> gpurun_out/${TAG}_final_bench_default.log; echo "rc=$?" | tee -a gpurun_out/${TAG}_final.log
tail -12 gpurun_out/${TAG}_final_bench_default.log | tee -a gpurun_out/${TAG}_final.log
step "default bench, --dist-single (RCCL with one rank; strong pass = the whole fixed job on one GPU)"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 python bench.py --dist-single > gpurun_out/${TAG}_final_dist_single_bench.json 2> gpurun_out/${TAG}_final_dist_single_bench.log; echo "rc=$?" | tee -a gpurun_out/${TAG}_final.log
tail -12 gpurun_out/${TAG}_final_dist_single_bench.log | tee -a gpurun_out/${TAG}_final.log
python3 - <<'PY' | tee -a gpurun_out/${TAG}_final.log
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_final_*bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    rows = [("C2", d)] + [(e["name"], e) for e in d.get("secondary", [])]
    print(f)
    for n, e in rows:
        r = e["roofline"]
        s = e.get("strong")
        print(f"  {n:18} {e['value']/1e9:8.2f} G/s kernel {e['kernel_ms']:.4f} ms frac {r['frac']:.4f} rocprof {r.get('frac_from_rocprof')} chain {r.get('frac_of_gather_chain')} traffic {r.get('traffic')} bound {r['bound']} parity {e['parity_vs_oracle']}"
              + (f" | strong {s['value']/1e9:.2f} G/s, {s.get('passes')} passes, {s['ms_per_step']:.2f} ms/step, empty step {s.get('empty_step_ms')}" if s else ""))
    print("  errors:", d.get("secondary_errors"), "wall", d.get("bench_wall_s"))
PY
step done
