// gather_ceiling.hip -- the ceilings the bench line quotes for the random-lookup kernels, re-measured on the box the bench runs on.
//
// What a lookup of the flat image costs the memory system is lines, not bytes: one pilot line (codeword array, a few MB),
// one 128-B block (or tail) line.  This tool measures, with NO hashing and no other work, the rates a kernel of that access shape can reach:
//   lines   scattered gathers of one 128-B-aligned line per lane from a table of `big` bytes                     -> G lines/s
//   chain   the dependent chain of a lookup: a 4-B word from a table of `small` bytes selects the 128-B element of the `big` table   -> G items/s
//   append  one device-scope atomicAdd per workgroup on ONE counter (the run-encoded result's allocation), by number of workgroups   -> G atomics/s
// Round 1 measured the first two once (profiles/r01_microbench.txt, r01_microbench2.txt: tools/microbench{,2,3}.hip, removed in round 5);
// bench.py --probe-ceiling runs this binary with the section sizes of the image at hand and writes profiles/gather_ceiling.json, which
// `roofline.frac_of_gather_chain` and FABRIC_LINE_PEAK are read from -- the ceiling no longer quotes the kernel's own rate.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_ceiling.hip -o tools/gather_ceiling
//   tools/gather_ceiling <small_bytes> <big_bytes> [<small_bytes> <big_bytes> ...]        one JSON object per line on stdout
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

__device__ __forceinline__ uint64_t splitmix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
__global__ void k_fill_idx(uint32_t* idx, size_t n, uint32_t nelem, uint64_t seed) {
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += st) idx[i] = (uint32_t)(splitmix(i ^ seed) % nelem);
}
__global__ void k_fill_small(uint32_t* tab, size_t n, uint32_t nelem_big) {  // every word of the small table names an element of the big one
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += st) tab[i] = (uint32_t)(splitmix(i * 7 + 1) % nelem_big);
}
__global__ void k_fill_big(uint64_t* tab, size_t nwords) {
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nwords; i += st) tab[i] = splitmix(i);
}

// one 128-B-aligned line per lane, 16 B of it read (beyond L2 the fetch granule is the line: profiles/r01_microbench2.txt)
__global__ __launch_bounds__(256) void k_lines(const uint32_t* __restrict__ idx, const uint4* __restrict__ big, uint64_t* __restrict__ out, size_t n) {
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += st) {
        const uint4 v = big[(size_t)__builtin_nontemporal_load(idx + i) * 8];
        __builtin_nontemporal_store((uint64_t)v.x + v.y + v.z + v.w, out + i);
    }
}
// the chain of a lookup: small[idx] (4 B) -> big[that] (one line)
__global__ __launch_bounds__(256) void k_chain(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ small, const uint4* __restrict__ big, uint64_t* __restrict__ out,
                                               size_t n) {
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += st) {
        const uint32_t j = small[__builtin_nontemporal_load(idx + i)];
        const uint4 v = big[(size_t)j * 8];
        __builtin_nontemporal_store((uint64_t)v.x + v.y + v.z + v.w, out + i);
    }
}
// one atomicAdd per workgroup on counter[(blockIdx % ncnt) * 32] (128-B apart), its result stored (the reservation is consumed); `spin` rounds of
// filler arithmetic per thread stand for the tile's other work, so that the atomics arrive spread over the kernel as they would
__global__ __launch_bounds__(128) void k_append(unsigned long long* __restrict__ counter, uint32_t ncnt, uint64_t* __restrict__ out, int spin) {
    uint64_t x = blockIdx.x * 128ull + threadIdx.x;
    for (int i = 0; i < spin; ++i) x = splitmix(x);
    unsigned long long at = 0;
    if (threadIdx.x == 0) at = atomicAdd(counter + (size_t)(blockIdx.x % ncnt) * 16, 40ull + (x & 7ull));
    if (threadIdx.x == 0 || x == 0x1234567ull) out[blockIdx.x] = at + x;
}

template <typename F>
static float timeit(F f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / reps;
}

int main(int argc, char** argv) {
    if (argc < 3 || (argc - 1) % 2) {
        fprintf(stderr, "usage: gather_ceiling <small_bytes> <big_bytes> [...]\n");
        return 2;
    }
    const size_t n = 1ull << 27;
    uint32_t* idx;
    uint64_t* out;
    CK(hipMalloc(&idx, n * 4));
    CK(hipMalloc(&out, n * 8));
    const int blocks = 1 << 19, threads = 256;  // one item per thread, as k_query_packed is launched
    for (int a = 1; a + 1 < argc; a += 2) {
        const size_t small_bytes = (strtoull(argv[a], nullptr, 10) + 255) & ~255ull, big_bytes = (strtoull(argv[a + 1], nullptr, 10) + 127) & ~127ull;
        if (small_bytes < 256 || big_bytes < 128 || big_bytes / 128 > 0xFFFFFFFFull) {
            fprintf(stderr, "bad sizes\n");
            return 2;
        }
        uint32_t* small;
        uint64_t* big;
        CK(hipMalloc(&small, small_bytes));
        CK(hipMalloc(&big, big_bytes));
        const uint32_t nbig = (uint32_t)(big_bytes / 128), nsmall = (uint32_t)(small_bytes / 4);
        k_fill_big<<<4096, 256>>>(big, big_bytes / 8);
        k_fill_small<<<4096, 256>>>(small, nsmall, nbig);
        CK(hipDeviceSynchronize());
        k_fill_idx<<<4096, 256>>>(idx, n, nbig, 11);
        CK(hipDeviceSynchronize());
        const float t_lines = timeit([&] { k_lines<<<blocks, threads>>>(idx, (const uint4*)big, out, n); }, 5);
        k_fill_idx<<<4096, 256>>>(idx, n, nsmall, 13);
        CK(hipDeviceSynchronize());
        const float t_chain = timeit([&] { k_chain<<<blocks, threads>>>(idx, small, (const uint4*)big, out, n); }, 5);
        printf("{\"probe\": \"gather\", \"items\": %zu, \"small_bytes\": %zu, \"big_bytes\": %zu, \"lines_per_s\": %.4g, \"chain_items_per_s\": %.4g}\n", n, small_bytes, big_bytes,
               n / (t_lines * 1e-3), n / (t_chain * 1e-3));
        fflush(stdout);
        CK(hipFree(small));
        CK(hipFree(big));
    }
    // the append probe: workgroups of 128 threads (the streaming kernel's), one atomic each
    unsigned long long* counter;
    CK(hipMalloc(&counter, 64 * 128));
    for (int spin : {0, 64}) {
        for (uint32_t ncnt : {1u, 8u, 64u}) {
            const int wgs = 1 << 21;
            CK(hipMemset(counter, 0, 64 * 128));
            const float t = timeit([&] { k_append<<<wgs, 128>>>(counter, ncnt, out, spin); }, 3);
            printf("{\"probe\": \"append\", \"workgroups\": %d, \"counters\": %u, \"filler_rounds\": %d, \"ms\": %.4f, \"atomics_per_s\": %.4g}\n", wgs, ncnt, spin, t, wgs / (t * 1e-3));
            fflush(stdout);
        }
    }
    return 0;
}
