// TEST INFRASTRUCTURE ONLY: host stand-in for lphash_amd/csrc/gfx950/shift_in.cuh (two gfx950 instructions there), found first on the wave
// emulator's include path.  Same contract: mask = 2 * mask + (a != b).
#pragma once
#include <stdint.h>

namespace lphq {
static inline void shift_in_ne(uint32_t& mask, uint32_t a, uint32_t b) { mask = 2u * mask + (a != b ? 1u : 0u); }
}  // namespace lphq
