// Scratch: two dependent gathers (small table 4 B -> big table, 16 B of a 128-B element) with the BIG table in memory of another type,
// to see whether a table that is not cached in L2 leaves the small one resident there (tools/microbench3.hip is the cached baseline).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench4.hip -o tools/microbench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t splitmix(uint64_t x) { x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31); }
__global__ void k_fill_tab(uint64_t* tab, size_t nwords) { size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x; for (size_t i = t; i < nwords; i += st) tab[i] = splitmix(i); }
__global__ __launch_bounds__(256) void k_two(const uint32_t* __restrict__ small, uint32_t nsmall, const u32x4* __restrict__ big, uint32_t nbig, uint64_t* __restrict__ out, size_t n) {
    size_t t = blockIdx.x * (size_t)256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (size_t i = t; i < n; i += st) {
        const uint32_t a = small[(uint32_t)(splitmix(i) % nsmall)];
        const u32x4 v = big[(size_t)((a ^ (uint32_t)splitmix(i + 7)) % nbig) * 8];
        __builtin_nontemporal_store((uint64_t)(v.x + v.y), out + i);
    }
}
// the same with the 128-B element fetched by 8 lanes in ONE request (lane l of a group of 8 reads piece l), owner = lane 0 of the group: 1/8 of the items per pass
__global__ __launch_bounds__(256) void k_two_coop(const uint32_t* __restrict__ small, uint32_t nsmall, const u32x4* __restrict__ big, uint32_t nbig, uint64_t* __restrict__ out, size_t n) {
    size_t t = blockIdx.x * (size_t)256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    const uint32_t lane = threadIdx.x & 63u;
    for (size_t i = t; i < n; i += st) {
        const uint32_t a = small[(uint32_t)(splitmix(i) % nsmall)];
        const uint32_t e = (a ^ (uint32_t)splitmix(i + 7)) % nbig;
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {  // round r serves the items of lanes 8r .. 8r+7: lane l reads piece (l & 7) of the element of lane 8r + (l >> 3)
            const uint32_t er = __shfl(e, 8 * r + (int)(lane >> 3), 64);
            const u32x4 v = big[(size_t)er * 8 + (lane & 7u)];
            const uint32_t part = v.x + v.y;
            // owner lane 8r + j gets the piece-0 value of group j: lane 8j holds it
            const uint32_t got = __shfl(part, (int)((lane & 7u) * 8u), 64);
            if ((lane >> 3) == (uint32_t)r) acc = got;
        }
        __builtin_nontemporal_store((uint64_t)acc, out + i);
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
    const size_t n = 1ull << 27;
    uint64_t* out; CK(hipMalloc(&out, n * 8));
    const size_t sb = 36 * (1 << 20) / 10, bb = (size_t)30 << 20;
    uint64_t* s; CK(hipMalloc(&s, sb));
    k_fill_tab<<<2048, 256>>>(s, sb / 8);
    const uint32_t ns = sb / 4, nb = bb / 128;
    struct { const char* name; unsigned flags; } kinds[] = {{"default (cached)", hipDeviceMallocDefault}, {"uncached", hipDeviceMallocUncached}, {"fine-grained", hipDeviceMallocFinegrained}};
    for (auto& kd : kinds) {
        uint64_t* b = nullptr;
        hipError_t e = hipExtMallocWithFlags((void**)&b, bb, kd.flags);
        if (e != hipSuccess) { printf("big table %-18s: allocation failed (%s)\n", kd.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        k_fill_tab<<<2048, 256>>>(b, bb / 8); CK(hipDeviceSynchronize());
        const float m1 = timeit([&] { k_two<<<524288, 256>>>((uint32_t*)s, ns, (u32x4*)b, nb, out, n); }, 3);
        const float m2 = timeit([&] { k_two_coop<<<524288, 256>>>((uint32_t*)s, ns, (u32x4*)b, nb, out, n); }, 3);
        printf("small 3.6 MB -> big 30 MB %-18s: one 16-B load per lane %.1f G items/s | 8 lanes per 128-B element %.1f G items/s\n", kd.name, n / m1 * 1e-6, n / m2 * 1e-6);
        CK(hipFree(b));
    }
    return 0;
}
