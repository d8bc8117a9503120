#!/bin/bash
# Timeline of `lphash query-p` on the bench's single-member .fa.gz (LPHQ_CLI_TRACE: per-batch next() / retire / queue times; LPHQ_PGZ_TRACE: the
# decoder's spans), several repetitions and a few settings.   gpurun --timeout 900 -- bash tools/cli_trace.sh
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; mkdir -p gpurun_out
python - <<'PY' 2>&1 | tee gpurun_out/cli_trace.txt
import gzip, os, subprocess, sys
sys.path.insert(0, ".")
import lphash_amd as L
from lphash_amd.synth import synthetic_records
flat, recs = synthetic_records(1, 300, 100030)
with gzip.open("/tmp/c2_300.fa.gz", "wb", compresslevel=6) as f:
    for i, r in enumerate(recs):
        f.write(b">s%d\n" % i); f.write(r); f.write(b"\n")
idx = "/tmp/c2_300.lph"
L.build(recs, 31, 20, idx, kmer_bits=64, threads=32, device=0)
def throttled():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except OSError:
        return (0, 0)
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
def run(env, extra=()):
    a = throttled()
    r = subprocess.run(["lphash_amd/lphash", "query-p", "-i", idx, "-q", "/tmp/c2_300.fa.gz", "--kmer-bits", "64", *extra], capture_output=True, text=True, env=dict(os.environ, **env))
    b = throttled()
    return r.stdout.strip().split(",")[2:] + [f"throttled +{b[0] - a[0]} periods, +{(b[1] - a[1]) / 1000:.1f} ms"], r.stderr
for rep in range(10):
    print("default (decoder: usable cpus, parser: 16)", run({})[0][1:3], "| --threads 16", run({}, ["--threads", "16"])[0][1:3], "| --threads 32", run({}, ["--threads", "32"])[0][1:3], "| HSA_ENABLE_INTERRUPT=0", run({"HSA_ENABLE_INTERRUPT": "0"})[0][1:3])
for rep in range(2):
    out, err = run({"LPHQ_CLI_TRACE": "1", "LPHQ_READER_TRACE": "1"})
    print(out); print("\n".join(l for l in err.splitlines() if l.startswith(("[cli]", "[reader]"))))
out, err = run({"LPHQ_CLI_TRACE": "1", "LPHQ_PGZ_TRACE": "1"})
print(out); print(err[-6000:])
PY
