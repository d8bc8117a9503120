// TEST INFRASTRUCTURE ONLY: a host stand-in for the few HIP names the product's device sources use, so that the very same
// device functions (lphash_amd/csrc/device_query.cuh) and kernels (kernels_device.cuh) can be compiled with g++ and checked
// against the oracle without a GPU (tests/test_emu_cpu.py).  Workgroups are emulated by cooperative fibers (emu_runtime.hpp):
// 64-lane waves, __syncthreads, __ballot/__any/__shfl, LDS = function-local statics, one workgroup at a time.
// Not a product path: nothing under lphash_amd/ includes or links this.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../emu_runtime.hpp"

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
// clang's cache-hint builtins (g++ has none): plain accesses on the host
template <typename T>
static inline T __builtin_nontemporal_load(const T* p) { return *p; }
template <typename T>
static inline void __builtin_nontemporal_store(T v, T* p) { __builtin_memcpy((void*)p, &v, sizeof(T)); }  // (p may be under-aligned for T)

struct uint4 {
    uint32_t x, y, z, w;
};
struct uint2 {
    uint32_t x, y;
};
struct ulonglong2 {
    unsigned long long x, y;
};
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }

static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
// v_perm_b32: result byte i = byte sel_i of the 8 bytes {b (0..3), a (4..7)}; selector 0x0C yields 0x00 (the other special selectors are not used)
static inline uint32_t __builtin_amdgcn_perm(uint32_t a, uint32_t b, uint32_t sel) {
    const uint64_t src = ((uint64_t)a << 32) | b;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t s = (sel >> (8 * i)) & 0xFFu;
        const uint32_t byte = s < 8u ? (uint32_t)(src >> (8 * s)) & 0xFFu : 0u;
        r |= byte << (8 * i);
    }
    return r;
}
// v_mbcnt_lo / v_mbcnt_hi: bits of the mask below the calling lane (low / high half of the wave), added to `acc`
static inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t mask, uint32_t acc) {
    const uint32_t lane = emu::thread_idx().x & 63u;
    return acc + (uint32_t)__builtin_popcount(mask & (lane >= 32u ? ~0u : ((1u << lane) - 1u)));
}
static inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t mask, uint32_t acc) {
    const uint32_t lane = emu::thread_idx().x & 63u;
    return acc + (lane > 32u ? (uint32_t)__builtin_popcount(mask & ((1u << (lane - 32u)) - 1u)) : 0u);
}
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint64_t min(uint64_t a, uint64_t b) { return a < b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }

#define threadIdx (emu::thread_idx())
#define blockIdx (emu::g_block_idx)
#define gridDim (emu::g_grid_dim)

static inline void __syncthreads() { emu::sync_block(); }
static inline int __syncthreads_or(int pred) { return emu::sync_block_or(pred != 0); }
static inline unsigned long long __ballot(int pred) { return emu::ballot(pred != 0); }
static inline int __any(int pred) { return emu::ballot(pred != 0) != 0; }
template <typename T>
static inline T __shfl(T v, int src, int = 64) { return (T)emu::shfl((uint64_t)v, src); }
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int = 64) { return (T)emu::shfl_down((uint64_t)v, delta); }
// one OS thread runs all fibers, so plain read-modify-write is atomic
static inline unsigned int atomicAdd(unsigned int* p, unsigned int v) { unsigned int o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
static inline unsigned int atomicOr(unsigned int* p, unsigned int v) { unsigned int o = *p; *p = o | v; return o; }
static inline unsigned int atomicMin(unsigned int* p, unsigned int v) { unsigned int o = *p; *p = v < o ? v : o; return o; }
