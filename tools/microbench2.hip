// Gather granularity + L2 pollution experiments (scratch).  hipcc --offload-arch=gfx950 -O3 tools/microbench2.hip -o tools/microbench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t splitmix(uint64_t x) { x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31); }
__global__ void k_fill_idx(uint32_t* idx, size_t n, uint32_t nelem, uint64_t seed) { size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x; for (size_t i = t; i < n; i += st) idx[i] = (uint32_t)(splitmix(i ^ seed) % nelem); }
__global__ void k_fill_tab(uint64_t* tab, size_t nwords) { size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x; for (size_t i = t; i < nwords; i += st) tab[i] = splitmix(i); }

// elements of ELEM bytes (128-B aligned table); read CH 16-B chunks starting at chunk offset OFF of the element
template <int CH, bool NT_IDX, bool NT_OUT, bool NT_TAB>
__global__ void k_gather(const uint32_t* __restrict__ idx, const uint4* __restrict__ table, uint64_t* __restrict__ out, size_t n, uint32_t elem16, uint32_t off16) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (size_t i = t; i < n; i += st) {
        uint32_t j = NT_IDX ? __builtin_nontemporal_load(idx + i) : idx[i];
        const uint4* p = table + (size_t)j * elem16 + off16;
        uint64_t acc = 0;
#pragma unroll
        for (int w = 0; w < CH; ++w) { u32x4 v = NT_TAB ? __builtin_nontemporal_load((const u32x4*)(p + w)) : *(const u32x4*)(p + w); acc += v.x + v.y + v.z + v.w; }
        if (NT_OUT) __builtin_nontemporal_store(acc, out + i); else out[i] = acc;
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
    uint32_t* idx; CK(hipMalloc(&idx, n * 4));
    uint64_t* out; CK(hipMalloc(&out, n * 8));
    int blocks = 4096, threads = 256;
    for (size_t smb : {3, 6, 24, 96, 1024}) {
        size_t bytes = smb << 20; uint64_t* tab; CK(hipMalloc(&tab, bytes));
        k_fill_tab<<<2048, 256>>>(tab, bytes / 8);
        uint32_t nelem = (uint32_t)(bytes / 128);
        k_fill_idx<<<2048, 256>>>(idx, n, nelem, smb); CK(hipDeviceSynchronize());
        const uint4* T = (const uint4*)tab;
        float a16 = timeit([&] { k_gather<1, false, false, false><<<blocks, threads>>>(idx, T, out, n, 8, 0); }, 3);
        float a64 = timeit([&] { k_gather<4, false, false, false><<<blocks, threads>>>(idx, T, out, n, 8, 0); }, 3);
        float a64b = timeit([&] { k_gather<4, false, false, false><<<blocks, threads>>>(idx, T, out, n, 8, 2); }, 3);  // straddles the two 64-B halves
        float a128 = timeit([&] { k_gather<8, false, false, false><<<blocks, threads>>>(idx, T, out, n, 8, 0); }, 3);
        float nt_io = timeit([&] { k_gather<4, true, true, false><<<blocks, threads>>>(idx, T, out, n, 8, 0); }, 3);
        float nt_tab = timeit([&] { k_gather<4, false, false, true><<<blocks, threads>>>(idx, T, out, n, 8, 0); }, 3);
        float nt_all = timeit([&] { k_gather<4, true, true, true><<<blocks, threads>>>(idx, T, out, n, 8, 0); }, 3);
        printf("table %5zu MB (128-B elems): 16B %.1f | 64B %.1f | 64B straddling halves %.1f | 128B %.1f | 64B nt idx+out %.1f | 64B nt table %.1f | 64B nt all %.1f  G gathers/s\n", smb,
               n / a16 * 1e-6, n / a64 * 1e-6, n / a64b * 1e-6, n / a128 * 1e-6, n / nt_io * 1e-6, n / nt_tab * 1e-6, n / nt_all * 1e-6);
        CK(hipFree(tab));
    }
    return 0;
}
