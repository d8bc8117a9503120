// Microbenchmarks that size the design: scattered gathers (by width and table size) and
// MurmurHash64A throughput on gfx950. Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t murmur8(uint64_t v, uint64_t seed) {
    const uint64_t m = 0xc6a4a7935bd1e995ULL;
    uint64_t h = seed ^ (8 * m);
    uint64_t k = v * m; k ^= k >> 47; k *= m; h ^= k; h *= m;
    h ^= h >> 47; h *= m; h ^= h >> 47;
    return h;
}
__device__ __forceinline__ uint64_t splitmix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31);
}

// each thread: ITER murmurs chained lightly (independent streams of 4)
template <int ILP>
__global__ void k_murmur(uint64_t* out, int iters) {
    uint64_t x[ILP];
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    for (int i = 0; i < ILP; ++i) x[i] = t * ILP + i;
    for (int it = 0; it < iters; ++it)
        for (int i = 0; i < ILP; ++i) x[i] = murmur8(x[i] & 0xFFFFFFFFFFULL, 42);
    uint64_t s = 0;
    for (int i = 0; i < ILP; ++i) s ^= x[i];
    out[t] = s;
}

// gather: idx precomputed (random), each thread does `per` gathers of WIDTH bytes from its own random lines
template <int WORDS16>  // number of 16B chunks per gather (consecutive), 0 => 8B gather
__global__ void k_gather(const uint32_t* __restrict__ idx, const uint4* __restrict__ table, uint64_t* out, size_t n, uint32_t stride16) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t step = (size_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (size_t i = t; i < n; i += step) {
        uint32_t j = idx[i];
        if (WORDS16 == 0) {
            const uint64_t* p = (const uint64_t*)(table) + (size_t)j * (stride16 * 2);
            acc += *p;
        } else {
            const uint4* p = table + (size_t)j * stride16;
#pragma unroll
            for (int w = 0; w < WORDS16; ++w) { uint4 v = p[w]; acc += v.x + v.y + v.z + v.w; }
        }
    }
    out[t] = acc;
}
// dependent chain of D gathers (8B) : latency-bound pattern like the lookup chain
template <int D>
__global__ void k_chain(const uint32_t* __restrict__ idx, const uint64_t* __restrict__ table, uint64_t* out, size_t n, uint32_t nelem) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t step = (size_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (size_t i = t; i < n; i += step) {
        uint32_t j = idx[i];
#pragma unroll
        for (int d = 0; d < D; ++d) { uint64_t v = table[(size_t)j * 8]; acc += v; j = (uint32_t)((v + j * 2654435761u) % nelem); }
    }
    out[t] = acc;
}
__global__ void k_fill_idx(uint32_t* idx, size_t n, uint32_t nelem, uint64_t seed) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = t; i < n; i += step) idx[i] = (uint32_t)(splitmix(i ^ seed) % nelem);
}
__global__ void k_fill_tab(uint64_t* tab, size_t nwords) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = t; i < nwords; i += step) tab[i] = splitmix(i);
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    uint64_t* out; CK(hipMalloc(&out, 8ull << 22));
    // ---- murmur throughput
    {
        int blocks = 256 * 8, threads = 256, iters = 256;
        float ms1 = timeit([&] { k_murmur<1><<<blocks, threads>>>(out, iters); }, 5);
        float ms4 = timeit([&] { k_murmur<4><<<blocks, threads>>>(out, iters); }, 5);
        double n1 = (double)blocks * threads * iters, n4 = n1 * 4;
        printf("murmur ILP1: %.1f Ghash/s   ILP4: %.1f Ghash/s\n", n1 / ms1 * 1e-6, n4 / ms4 * 1e-6);
    }
    // ---- gathers
    const size_t n = 1ull << 27;
    uint32_t* idx; CK(hipMalloc(&idx, n * 4));
    size_t sizes_mb[] = {2, 8, 24, 64, 192, 1024, 4096};
    for (size_t smb : sizes_mb) {
        size_t bytes = smb << 20;
        uint64_t* tab; CK(hipMalloc(&tab, bytes));
        k_fill_tab<<<2048, 256>>>(tab, bytes / 8);
        uint32_t nelem = (uint32_t)(bytes / 64);  // 64B elements
        k_fill_idx<<<2048, 256>>>(idx, n, nelem, smb);
        CK(hipDeviceSynchronize());
        int blocks = 256 * 8, threads = 256;
        float g8  = timeit([&] { k_gather<0><<<blocks, threads>>>(idx, (const uint4*)tab, out, n, 4); }, 3);
        float g16 = timeit([&] { k_gather<1><<<blocks, threads>>>(idx, (const uint4*)tab, out, n, 4); }, 3);
        float g32 = timeit([&] { k_gather<2><<<blocks, threads>>>(idx, (const uint4*)tab, out, n, 4); }, 3);
        float g64 = timeit([&] { k_gather<4><<<blocks, threads>>>(idx, (const uint4*)tab, out, n, 4); }, 3);
        float c3  = timeit([&] { k_chain<3><<<blocks, threads>>>(idx, tab, out, n, nelem); }, 3);
        printf("table %5zu MB: gather 8B %.1f G/s | 16B %.1f G/s | 32B %.1f G/s | 64B %.1f G/s | chain3x8B %.1f Glookups/s\n", smb,
               n / g8 * 1e-6, n / g16 * 1e-6, n / g32 * 1e-6, n / g64 * 1e-6, n / c3 * 1e-6);
        CK(hipFree(tab));
    }
    // idx stream alone (coalesced read baseline)
    return 0;
}
