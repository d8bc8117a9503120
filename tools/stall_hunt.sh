#!/bin/bash
# tools/stall_hunt.sh TAG N -- the default bench line N times; per run, every workload whose slowest timed step is 1.3x its median, with what the host spent
# queueing each step and whether Python's collector ran inside the timed region (the C4-stream-runs step of some default runs carries a host gap)
set -u
tag=${1:-r06}; n=${2:-3}
mkdir -p gpurun_out
for i in $(seq 1 "$n"); do
  python bench.py > "gpurun_out/${tag}_stall_$i.json" 2> "gpurun_out/${tag}_stall_$i.err"
  python - "gpurun_out/${tag}_stall_$i.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
for s in [d] + d.get("secondary", []):
    st, hq = s.get("kernel_ms_steps"), s.get("host_queue_ms_steps")
    if not st:
        continue
    med = sorted(st)[len(st) // 2]
    flag = "STALL" if max(st) > 1.3 * med else "ok"
    print(f"{sys.argv[1]} {s.get('name', 'C2'):<20} {flag:<6} value {s['value'] / 1e9:8.2f} G  steps {st}  host {hq}  gc {s.get('python_gc_in_timed_region')}")
PY
done
