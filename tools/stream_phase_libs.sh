#!/bin/bash
# Builds profiling variants of the library whose streaming kernel returns after phase A, B, C+D or E (tools/phase_libs/liblphq_stop<N>.so;
# the product sources are not touched: the early exits are spliced into a temporary copy at the phase markers of kernels_device.cuh).
#   [PHASE_FILE=stream8_device.cuh] bash tools/stream_phase_libs.sh            then on the GPU box:  LPHQ_LIB_PATH=tools/phase_libs/liblphq_stop2.so python bench.py --mode stream ...
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/lphq_phase.XXXXXX)
mkdir -p "$R/tools/phase_libs"
for n in 1 2 3 4; do
  rm -rf "$T/csrc"; mkdir -p "$T/x/y"; cp -r "$R/lphash_amd/csrc" "$T/x/y/csrc"; mkdir -p "$T/x/include"; cp "$R/include/lphq.h" "$T/x/include/"
  D=$T/x/y/csrc
  python3 - "$D/${PHASE_FILE:-kernels_device.cuh}" $n <<'P'
import sys
p, n = sys.argv[1], int(sys.argv[2])
s = open(p).read()
marks = ["    // ---- B: hashes", "    // ---- C + D: leftmost", "    // ---- E: one lookup", "    // ---- F: the lane"]
m = marks[n - 1]
assert s.count(m) == 1, m
s = s.replace(m, "    if (tile_base != ~0ull) return;  // profiling variant: stop after phase %d\n%s" % (n, m))
open(p, "w").write(s)
P
  make -C "$D" -j8 OUT="$R/tools/phase_libs/liblphq_stop$n.so" CLI="$T/lphash_cli_$n" "$R/tools/phase_libs/liblphq_stop$n.so" > "$T/make$n.log" 2>&1 || { tail -5 "$T/make$n.log"; exit 1; }
  rm -rf "$T/x"
done
ls -la "$R/tools/phase_libs"
rm -rf "$T"
