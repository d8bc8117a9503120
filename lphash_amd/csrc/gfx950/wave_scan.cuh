// gfx950/wave_scan.cuh -- inclusive prefix sum of one 32-bit value per lane over the wave (64 lanes), on the DPP data path of the CDNA3/4 ISA:
// four shifted adds inside each row of 16 lanes (row_shr:1, 2, 4, 8 -- a lane without a source keeps its value), then lane 15 of a row into the
// next row (row_bcast:15, rows 1 and 3) and lane 31 into the upper half (row_bcast:31): 6 adds, no LDS traffic, no ballots.  The streaming kernel
// numbers the super-k-mer leaders of a wave with it (one scan per 8 windows of a lane, where round 3 paid 16 ballots and 16 mbcnt pairs).
// Included as <gfx950/wave_scan.cuh>; tests/emu/gfx950/wave_scan.cuh is the host stand-in the wave emulator compiles instead.
#pragma once
#include <stdint.h>

namespace lphq {

__device__ __forceinline__ uint32_t wave_inclusive_add(uint32_t x) {
    // __builtin_amdgcn_update_dpp(old, src, dpp_ctrl, row_mask, bank_mask, bound_ctrl): lanes whose source lies outside the row (or whose row is
    // masked out) get `old` = 0, i.e. add nothing
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return x;
}

}  // namespace lphq
