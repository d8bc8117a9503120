// TEST INFRASTRUCTURE ONLY: host stand-in for lphash_amd/csrc/gfx950/vgpr_allocation.cuh (one gfx950 instruction there), found first on the wave emulator's
// include path.  The emulator has no register file to size.
#pragma once

namespace lphq {
static inline void allocate_64_vgprs() {}
}  // namespace lphq
