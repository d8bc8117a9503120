// TEST INFRASTRUCTURE ONLY: host stand-in for lphash_amd/csrc/gfx950/wave_scan.cuh (DPP adds there), found first on the wave emulator's include
// path.  Same contract: the inclusive prefix sum of x over the 64 lanes of the wave.
#pragma once
#include <stdint.h>

namespace lphq {
static inline uint32_t wave_inclusive_add(uint32_t x) {
    const unsigned lane = threadIdx.x & 63u;
    uint32_t acc = 0;
    for (unsigned s = 0; s < 64; ++s) {  // (a shuffle is a collective of the wave: every lane asks for every source)
        const uint32_t v = (uint32_t)emu::shfl((uint64_t)x, (int)s);
        if (s <= lane) acc += v;
    }
    return acc;
}
}  // namespace lphq
