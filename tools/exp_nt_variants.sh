#!/bin/bash
# Round-5 experiment: non-temporal hints on the loads of the eight-per-lane streaming kernel, one library per variant, built from a patched
# COPY of the current sources (the product sources are not touched):   tools/exp_nt_variants.sh   ->  lphash_amd/liblphq_{ntb,ntu,ntbu}.so
#   ntb   the bases of a tile (phase A: one 8-byte + one 4-byte load per lane) with `nt`
#   ntu   the fused-block line of a lookup (uni_load: six / eight 16-byte loads of one 128-B line) with `nt`
#   ntbu  both -- with the code stores (`nt` since round 2) every stream but the pilot words is then marked non-temporal
# A/B: gpurun -- bash tools/ab.sh r05_nt "--k 63 --kmers 2000000000 --mode stream" lphash_amd/liblphq.so lphash_amd/liblphq_ntb.so lphash_amd/liblphq_ntu.so lphash_amd/liblphq_ntbu.so
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
build() { # name, python patch
  local name=$1 T
  T=$(mktemp -d /tmp/lphq_nt.XXXXXX)
  mkdir -p "$T/lphash_amd" "$T/include"
  cp -r "$R/lphash_amd/csrc" "$T/lphash_amd/csrc"; cp "$R/include/lphq.h" "$T/include/"; rm -f "$T"/lphash_amd/csrc/*.o
  python3 - "$T/lphash_amd/csrc" "$name" <<'PY'
import sys
d, name = sys.argv[1], sys.argv[2]
if "b" in name[2:]:
    p = d + "/stream8_device.cuh"; s = open(p).read()
    a = "        const uint2 v = *(const uint2*)(seq_al + t0 + 8ull * tid);  // (t0 and seq_al are 16-byte aligned)\n        xs[0] = v.x;\n        xs[1] = v.y;\n"
    assert a in s
    s = s.replace(a, "        const uint64_t v = __builtin_nontemporal_load((const uint64_t*)(seq_al + t0 + 8ull * tid));\n        xs[0] = (uint32_t)v;\n        xs[1] = (uint32_t)(v >> 32);\n")
    b = "        if (live2) xs[2] = *(const uint32_t*)(seq_al + t0 + 4ull * dd[2]);"
    assert b in s
    s = s.replace(b, "        if (live2) xs[2] = __builtin_nontemporal_load((const uint32_t*)(seq_al + t0 + 4ull * dd[2]));")
    c = "            if (have[r]) xs[r] = *(const uint32_t*)(seq_al + a);"
    assert c in s
    s = s.replace(c, "            if (have[r]) xs[r] = __builtin_nontemporal_load((const uint32_t*)(seq_al + a));")
    open(p, "w").write(s)
if "u" in name[2:]:
    p = d + "/device_query.cuh"; s = open(p).read()
    a = "    r.b01 = p[0];\n    r.b23 = p[1];\n#pragma unroll\n    for (uint32_t i = 0; i < UniRegs<P>::NV; ++i) r.v[i] = p[2 + i];\n"
    assert a in s
    s = s.replace(a, "    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));\n    const u32x4* q = (const u32x4*)p;\n"
                     "    auto ld = [&](uint32_t i) { const u32x4 t = __builtin_nontemporal_load(q + i); return make_uint4(t.x, t.y, t.z, t.w); };\n"
                     "    r.b01 = ld(0);\n    r.b23 = ld(1);\n#pragma unroll\n    for (uint32_t i = 0; i < UniRegs<P>::NV; ++i) r.v[i] = ld(2 + i);\n")
    open(p, "w").write(s)
PY
  make -C "$T/lphash_amd/csrc" -j8 ../liblphq.so > "$T/build.log" 2>&1 || { tail -20 "$T/build.log"; exit 1; }
  cp "$T/lphash_amd/liblphq.so" "$R/lphash_amd/liblphq_$name.so"
  rm -rf "$T"; echo "built lphash_amd/liblphq_$name.so"
}
for v in ntb ntu ntbu; do build $v; done
