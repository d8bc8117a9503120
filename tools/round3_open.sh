#!/bin/bash
# First GPU call of round 3: the whole -m gpu suite (now with C4 at 2 B k-mers), smoke, and the default bench line with its `secondary`
# list, exactly as the driver runs it.   gpurun --timeout 2400 -- bash tools/round3_open.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03_open}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
step() { echo "[$(date +%H:%M:%S)] $*" | tee -a "$O/log.txt"; }
free -g | head -2 | tee -a "$O/log.txt"; nproc | tee -a "$O/log.txt"
if [ "${SKIP_TESTS:-0}" = 1 ]; then step "tests skipped"; else
step "1. pytest -m gpu"
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=8 > "$O/pytest_gpu.txt" 2>&1
echo "exit code $?" >> "$O/pytest_gpu.txt"
tail -14 "$O/pytest_gpu.txt" | tee -a "$O/log.txt"
step "2. smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a "$O/log.txt"
fi
step "3. default bench line (headline C2 + secondary C3 / C5 / C4-random / C4-stream), the driver's command"
T0=$(date +%s); timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "exit code $?" | tee -a "$O/log.txt"
echo "bench wall $(( $(date +%s) - T0 )) s" | tee -a "$O/log.txt"; tail -3 "$O/bench_default.err" | cut -c1-300 | tee -a "$O/log.txt"
python3 - "$O/bench_default.json" <<'P' | tee -a "$O/log.txt"
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["config"]["workload"][:40], "%.2f G/s" % (d["value"] / 1e9), "kernel_ms %.3f" % d["kernel_ms"], "frac %.3f" % d["roofline"]["frac"], "parity", d["parity_vs_oracle"], "wall", round(d["bench_wall_s"], 1))
for e in d.get("secondary", []):
    print(e["name"], "%.2f G/s" % (e["value"] / 1e9), "kernel_ms %.3f" % e["kernel_ms"], "frac %.3f" % e["roofline"]["frac"], "valu %.2f" % e.get("roofline_valu", {}).get("frac", float("nan")), "parity", e["parity_vs_oracle"], "cpu %.2f M/s" % (e["cpu_baseline"]["value"] / 1e6), "wall", round(e["wall_s"], 1))
print("errors", d.get("secondary_errors"))
P
step done
