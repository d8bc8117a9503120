#!/bin/bash
# One step from a recursive clone of the reference to pinned parity (README.md beside this file has the same steps by hand):
#     git clone --recursive https://github.com/jermp/lphash ~/lphash
#     bash tests/golden/ref/make_cases.sh ~/lphash
#     python -m pytest tests/test_host_cpu.py -k reference_built -q && python -m pytest tests/test_gpu_parity.py -m gpu -k reference_built -q
# Builds the reference twice (kmer_t = uint64_t and __uint128_t, include/compile_constants.tpd:1-2; scripts/experiments.sh:76), builds
# dump_codes.cpp against each, and writes three case directories next to this script: the reference's own build-p / build-u on its bundled
# unitigs (BASELINE.json configs[0]: k=31 m=20; README.md:183: k=63 m=17), a 200-kbase slice of its bundled query genome, and the codes the
# reference computes for it.  Needs cmake, a C++17 compiler, zlib; no GPU.  NOT run in the build container of this repository: the
# reference tree there has an empty external/pthash.
set -euo pipefail
REF=$(cd "${1:?usage: make_cases.sh <recursive clone of jermp/lphash>}" && pwd)
HERE=$(cd "$(dirname "$0")" && pwd)
[ -f "$REF/external/pthash/include/pthash.hpp" ] || { echo "external/pthash is empty: clone with --recursive" >&2; exit 2; }
W=$(mktemp -d)
trap 'rm -rf "$W"' EXIT
SRCS="src/constants.cpp src/quartet_wtree.cpp src/minimizer.cpp src/partitioned_mphf.cpp src/unpartitioned_mphf.cpp src/mphf_utils.cpp"
cp "$REF/include/compile_constants.tpd" "$W/tpd.orig"
restore() { cp "$W/tpd.orig" "$REF/include/compile_constants.tpd"; }
trap 'restore; rm -rf "$W"' EXIT
for B in 128 64; do
  if [ $B = 64 ]; then printf '// typedef __uint128_t kmer_t;\ntypedef uint64_t kmer_t;\n' > "$REF/include/compile_constants.tpd"
  else printf 'typedef __uint128_t kmer_t;\n// typedef uint64_t kmer_t;\n' > "$REF/include/compile_constants.tpd"; fi
  cmake -S "$REF" -B "$W/build$B" -DCMAKE_BUILD_TYPE=Release > "$W/cmake$B.log"
  cmake --build "$W/build$B" --target lphash -j > "$W/make$B.log"
  (cd "$REF" && g++ -std=c++17 -O3 -march=native -pthread -I. "$HERE/dump_codes.cpp" $SRCS -lz -o "$W/dump_codes$B")
done
restore
# (no `zcat | head -c`: head leaves early, zcat dies of SIGPIPE and `set -o pipefail` would end the script here)
python3 -c 'import gzip, sys; sys.stdout.buffer.write(gzip.open(sys.argv[1]).read(200100))' "$REF/data/queries/salmonella_enterica.fasta.gz" | gzip > "$W/q.fa.gz"
case_dir() { # name tool bits k m input
  local d="$HERE/$1" tool=$2 bits=$3 k=$4 m=$5 in=$6 kind
  mkdir -p "$d"
  "$W/build$bits/lphash" build-$tool -i "$REF/data/unitigs_stitched/$in" -k $k -m $m -o "$d/index.lph" > "$d/build_stats.csv"
  cp "$REF/data/unitigs_stitched/$in" "$d/input.fa.gz"
  cp "$W/q.fa.gz" "$d/queries.fa.gz"
  "$W/dump_codes$bits" $tool "$d/index.lph" "$d/queries.fa.gz" "$d/codes_streaming.u64" "$d/codes_random.u64"
  printf '{"kind": "%s", "kmer_bits": %s, "k": %s, "m": %s, "seed": 42, "c": 3.0, "input": "%s", "queries": "queries.fa.gz"}\n' $tool $bits $k $m $in > "$d/meta.json"
  echo "wrote $d ($(stat -c %s "$d/index.lph") bytes, $(($(stat -c %s "$d/codes_random.u64") / 8)) codes per pass)"
}
case_dir p_se_k31_m20_b64 p 64 31 20 se.ust.k31.fa.gz
case_dir u_se_k31_m20_b64 u 64 31 20 se.ust.k31.fa.gz
case_dir p_se_k63_m17_b128 p 128 63 17 se.ust.k63.fa.gz
echo "now: python -m pytest tests/test_host_cpu.py -k reference_built -q   (and, on a GPU box: python -m pytest tests/test_gpu_parity.py -m gpu -k reference_built -q)"
