// gfx950/vgpr_allocation.cuh -- make a kernel's VGPR allocation 64 registers whatever the compiler ends up needing (one v_mov to v63 at the call).
// Why a kernel would want that: stream8_device.cuh, the payload sink (profiles/r06_transposed_gather_attempt.txt).  Costs nothing where LDS or the launch bounds
// already hold a SIMD at 8 waves.  Included as <gfx950/vgpr_allocation.cuh>; tests/emu/gfx950/vgpr_allocation.cuh is the host stand-in (nothing to do there).
#pragma once

namespace lphq {

__device__ __forceinline__ void allocate_64_vgprs() { asm volatile("v_mov_b32 v63, 0" ::: "v63"); }

}  // namespace lphq
