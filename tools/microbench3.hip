// Two-table gather with cache-policy variants on the big table (scratch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t splitmix(uint64_t x) { x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31); }
__global__ void k_fill_tab(uint64_t* tab, size_t nwords) { size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x; for (size_t i = t; i < nwords; i += st) tab[i] = splitmix(i); }

template <int POL> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
    u32x4 v;
    if (POL == 0) return *p;
    else if (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 6) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 7) asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// per item: dependent chain small[j1] -> big[j2]; ITEMS independent items per thread iteration for MLP
template <int POL, int ITEMS>
__global__ __launch_bounds__(256) void k_two(const uint32_t* __restrict__ small, uint32_t nsmall, const u32x4* __restrict__ big, uint32_t nbig, uint64_t* __restrict__ out, size_t n) {
    size_t t = blockIdx.x * (size_t)256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (size_t i = t; i < n; i += st * ITEMS) {
        uint32_t a[ITEMS]; u32x4 v[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) a[q] = small[(uint32_t)(splitmix(i + q * st) % nsmall)];
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) v[q] = ld<POL>(big + (size_t)((a[q] ^ (uint32_t)splitmix(i + q * st + 7)) % nbig) * 8);
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) if (i + q * st < n) out[i + q * st] = v[q].x + v[q].y;
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
template <int POL> void bench(const char* name, const uint32_t* s, uint32_t ns, const u32x4* b, uint32_t nb, uint64_t* out, size_t n) {
    float m1 = timeit([&] { k_two<POL, 1><<<4096, 256>>>(s, ns, b, nb, out, n); }, 3);
    float m2 = timeit([&] { k_two<POL, 2><<<4096, 256>>>(s, ns, b, nb, out, n); }, 3);
    printf("  big-table policy %-10s: %.1f G items/s (1 item/iter)  %.1f (2 items/iter)\n", name, n / m1 * 1e-6, n / m2 * 1e-6);
}
int main() {
    const size_t n = 1ull << 27;
    uint64_t* out; CK(hipMalloc(&out, n * 8));
    for (size_t small_mb10 : {36, 20}) for (size_t big_mb : {30, 16}) {
        size_t sb = small_mb10 * (1 << 20) / 10, bb = big_mb << 20;
        uint64_t *s, *b; CK(hipMalloc(&s, sb)); CK(hipMalloc(&b, bb));
        k_fill_tab<<<2048, 256>>>(s, sb / 8); k_fill_tab<<<2048, 256>>>(b, bb / 8); CK(hipDeviceSynchronize());
        printf("small %.1f MB (4-B gathers) -> big %zu MB (16-B gather from a 128-B element)\n", small_mb10 / 10.0, big_mb);
        uint32_t ns = sb / 4, nb = bb / 128;
        bench<0>("default", (uint32_t*)s, ns, (u32x4*)b, nb, out, n);
        bench<7>("asm-plain", (uint32_t*)s, ns, (u32x4*)b, nb, out, n);
        bench<1>("nt", (uint32_t*)s, ns, (u32x4*)b, nb, out, n);
        bench<2>("sc0", (uint32_t*)s, ns, (u32x4*)b, nb, out, n);
        bench<3>("sc1", (uint32_t*)s, ns, (u32x4*)b, nb, out, n);
        bench<4>("sc0 sc1", (uint32_t*)s, ns, (u32x4*)b, nb, out, n);
        bench<5>("sc1 nt", (uint32_t*)s, ns, (u32x4*)b, nb, out, n);
        bench<6>("sc0 nt", (uint32_t*)s, ns, (u32x4*)b, nb, out, n);
        CK(hipFree(s)); CK(hipFree(b));
    }
    return 0;
}
