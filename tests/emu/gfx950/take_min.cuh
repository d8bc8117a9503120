// TEST INFRASTRUCTURE ONLY: host stand-in for lphash_amd/csrc/gfx950/take_min.cuh (inline gfx950 assembly there), found first on the wave
// emulator's include path.  Same contract: if (h < mh) { mh = h; mp = p; } -- strict, the value seen first wins ties.
#pragma once
#include <stdint.h>

namespace lphq {
static inline void take_min_uniform(uint64_t& mh, uint32_t& mp, uint64_t h, uint32_t p) {
    if (h < mh) {
        mh = h;
        mp = p;
    }
}
}  // namespace lphq
