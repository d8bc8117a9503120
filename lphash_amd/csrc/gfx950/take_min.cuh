// gfx950/take_min.cuh -- "keep the smaller hash and where it was", the inner step of every minimizer scan, written for the CDNA3/4 ISA.
//
//     if (h < mh) { mh = h; mp = p; }          (strict: the value seen first wins ties -- the LEFTMOST minimum when scanning left to right,
//                                               as `<=` from the right end does in include/mphf_utils.hpp:130 and the strict `>` comparisons
//                                               of include/partitioned_mphf.hpp:122,150-163)
//
// The compiler turns the C++ form into v_cmp_lt_u64 + three v_cndmask_b32 (+ a v_mov_b32 when p is uniform: an SGPR operand next to the
// implicit VCC would be two constant-bus reads) = 4-5 VALU instructions.  Here the lanes that take the new value are made the EXEC mask for
// two moves (v_mov_b64 is one instruction on gfx940+): 3 VALU + 2 SALU.  The random kernel at k=63 is bound by VALU issue (44 of these per
// lookup, profiles/r03_isa_murmur.txt), the scalar unit is idle beside it.
// The block ends with `s_nop 0`: the compiler's hazard recognizer does not see the two VALU writes inside an inline-assembly block, and a DPP or
// v_readlane consumer of mh / mp scheduled right behind it needs two wait states after a VALU write, of which `s_mov_b64 exec` is only one (today's
// callers feed ordinary VALU instructions, for which nothing is needed; the nop issues on the scalar port of a VALU-bound kernel).
// tests/test_isa_budget_cpu.py checks that the compiled minimizer loop really contains this sequence.
// Included as <gfx950/take_min.cuh>; tests/emu/gfx950/take_min.cuh is the host stand-in the wave emulator compiles instead.
#pragma once
#include <stdint.h>

namespace lphq {

// p uniform over the wave (a loop counter, a compile-time position)
__device__ __forceinline__ void take_min_uniform(uint64_t& mh, uint32_t& mp, uint64_t h, uint32_t p) {
    uint64_t saved;
    asm("v_cmp_lt_u64 vcc, %[h], %[mh]\n\t"
        "s_and_saveexec_b64 %[sv], vcc\n\t"
        "v_mov_b64 %[mh], %[h]\n\t"
        "v_mov_b32 %[mp], %[p]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_nop 0"
        : [mh] "+v"(mh), [mp] "+v"(mp), [sv] "=&s"(saved)
        : [h] "v"(h), [p] "s"(p)
        : "vcc", "scc");
}
}  // namespace lphq
