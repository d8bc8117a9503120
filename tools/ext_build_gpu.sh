#!/bin/bash
# Construction with a memory limit on the GPU box: the build tests of the -m gpu suite, a few minutes of construction fuzz (batch splits and limits),
# then a C4-sized input (2 G bases, k = 63, m = 20) built in memory and with --max-memory 4 / 1: wall time, peak RSS, same file.
#   gpurun --timeout 2400 -- bash tools/ext_build_gpu.sh [tag] [bases]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04_ext_build}
BASES=${2:-2000000000}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout 900 python -m pytest tests/test_zz_gpu_next.py -x -q -m gpu -k "build" > "$O/pytest.txt" 2>&1; tail -3 "$O/pytest.txt"
timeout 400 python tests/gpu_build_fuzz.py 240 > "$O/fuzz.txt" 2>&1; tail -2 "$O/fuzz.txt"
fi
T=${THREADS:-16}
W=/tmp/extb; mkdir -p $W
python3 - $BASES $W/in.fa <<'P'
import sys, numpy as np
n, path = int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(1)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
with open(path, "wb") as f:
    per = 4_000_000
    for i in range(n // per):
        f.write(b">r%d\n" % i); f.write(lut[rng.integers(0, 4, per, dtype=np.uint8)].tobytes()); f.write(b"\n")
P
ls -la $W/in.fa | tee "$O/builds.txt"
run() { # label, args...
  local label=$1; shift
  python3 - "$label" "$@" <<'P' | tee -a "$O/builds.txt"
import subprocess, sys, resource, time
t = time.time()
r = subprocess.run(sys.argv[2:], capture_output=True, text=True)
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
print("%-28s rc %d  wall %6.1f s  peak RSS %7.0f MB  %s" % (sys.argv[1], r.returncode, time.time() - t, ru.ru_maxrss / 1024, r.stdout.strip()[-160:]))
if r.returncode: print(r.stderr[-600:])
open("/tmp/extb/%s.err" % sys.argv[1].replace(" ", "_"), "w").write(r.stderr)
P
}
E=$R/lphash_amd/lphash
export LPHQ_BUILD_VERBOSE=1
run "in memory"        $E build-p -i $W/in.fa -k 63 -m 20 -o $W/a.lph -t $T --verbose
run "max-memory 4"     $E build-p -i $W/in.fa -k 63 -m 20 -o $W/b.lph -d $W/tmp --max-memory 4 -t $T --verbose
run "max-memory 1"     $E build-p -i $W/in.fa -k 63 -m 20 -o $W/c.lph -d $W/tmp --max-memory 1 -t $T --verbose
cmp $W/a.lph $W/b.lph && cmp $W/a.lph $W/c.lph && echo "same file ($(stat -c %s $W/a.lph) bytes)" | tee -a "$O/builds.txt"
for f in in_memory max-memory_1; do echo "== $f" >> "$O/builds.txt"; grep -h "lphq_build\]" /tmp/extb/$f.err | grep -v "gpu extraction:\|  pthash:" | tail -24 >> "$O/builds.txt"; done
ls $W/tmp | wc -l | tee -a "$O/builds.txt"
