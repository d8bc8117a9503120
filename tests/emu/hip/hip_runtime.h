// TEST INFRASTRUCTURE ONLY: the handful of HIP names lphash_amd/csrc/device_query.cuh needs, so that the very same device
// functions can be compiled for the host (g++) and checked against the oracle without a GPU (tests/test_emu_cpu.py).
// Not a product path: nothing under lphash_amd/ includes or links this.
#pragma once
#include <cstdint>
#define __device__
#define __host__
#define __forceinline__ inline
struct uint4 {
    uint32_t x, y, z, w;
};
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
