#!/bin/bash
# Builds lphash_amd/liblphq_<name>.so from the current tree with some csrc files taken from a git revision (A/B of kernel changes on one box with
# tools/ab3.sh / pmc_quick.sh; same ABI as the current library, so the current Python side loads it):
#   tools/build_variant.sh <name> <file>@<rev> [<file>@<rev> ...]        e.g.  tools/build_variant.sh base stream8_device.cuh@HEAD
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
T=$(mktemp -d /tmp/lphq_variant.XXXXXX)
mkdir -p "$T/lphash_amd" "$T/include"
cp -r "$R/lphash_amd/csrc" "$T/lphash_amd/csrc"
cp "$R/include/lphq.h" "$T/include/"
rm -f "$T"/lphash_amd/csrc/*.o
for spec in "$@"; do
  f=${spec%@*}; rev=${spec#*@}
  git -C "$R" show "$rev:lphash_amd/csrc/$f" > "$T/lphash_amd/csrc/$f"
done
make -C "$T/lphash_amd/csrc" -j8 ../liblphq.so > "$T/build.log" 2>&1 || { tail -20 "$T/build.log"; exit 1; }
cp "$T/lphash_amd/liblphq.so" "$R/lphash_amd/liblphq_$NAME.so"
rm -rf "$T"
echo "built lphash_amd/liblphq_$NAME.so"
