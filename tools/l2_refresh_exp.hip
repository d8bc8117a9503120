// Experiment (scratch tool, not part of the product): can the PTHash pilot table be KEPT in the per-XCD L2 by a companion kernel that
// re-reads it continuously while k_query_packed runs?  A timing-only experiment showed what pilot lines that always hit L2 would be worth
// (C2 7.27 -> 5.63 ms, profiles/r02_pilot_hit_upper_bound.txt); this measures whether a refresher gets any of it.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilphash_amd/csrc tools/l2_refresh_exp.hip -o tools/l2_refresh_exp -Llphash_amd -llphq -Wl,-rpath,$PWD/lphash_amd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "image_layout.h"
#include "../include/lphq.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t splitmix(uint64_t x) { x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31); }
__global__ void k_gen(uint64_t* kmers, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) kmers[i] = splitmix(i) >> 2;
}
// workgroup w (assumed on XCD w % 8) re-reads slice (w / 8) of (gridDim.x / 8) of [p, p + n16) until *stop != 0 or `sweeps` sweeps are done
__global__ __launch_bounds__(256) void k_refresh(const uint4* __restrict__ p, size_t n16, volatile const int* stop, int sweeps, uint32_t* sink) {
    const size_t per = gridDim.x / 8 ? gridDim.x / 8 : 1, me = blockIdx.x / 8;
    const size_t a = n16 * me / per, b = n16 * (me + 1) / per;
    uint32_t acc = 0;
    for (int s = 0; s < sweeps && !*stop; ++s) {
        for (size_t i = a + threadIdx.x; i < b; i += 256) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = __builtin_nontemporal_load((const u32x4*)(p + i));
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_refresh_plain(const uint4* __restrict__ p, size_t n16, volatile const int* stop, int sweeps, uint32_t* sink) {
    const size_t per = gridDim.x / 8 ? gridDim.x / 8 : 1, me = blockIdx.x / 8;
    const size_t a = n16 * me / per, b = n16 * (me + 1) / per;
    uint32_t acc = 0;
    for (int s = 0; s < sweeps && !*stop; ++s) {
        for (size_t i = a + threadIdx.x; i < b; i += 256) {
            const uint4 v = p[i];
            asm volatile("" : "+v"(acc));
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345u) *sink = acc;
}
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/lphq_bench_cache/synth_seed1_p_k31_m20_n100000000_b64.lph";
    lphq_index* idx; if (lphq_open(path, LPHQ_PARTITIONED, 64, &idx)) { printf("open: %s\n", lphq_last_error()); return 1; }
    if (lphq_attach(idx, 0)) { printf("attach: %s\n", lphq_last_error()); return 1; }
    size_t nb; lphq_image_bytes(idx, &nb); std::vector<uint8_t> himg(nb); lphq_image_copy(idx, himg.data(), nb);
    lphq_image_header H; memcpy(&H, himg.data(), sizeof H);
    const void* dimg; size_t dn; lphq_device_image(idx, &dimg, &dn);
    // pilot region: front ranks .. end of the hp tables (the sections are laid out ranks, hp, ranks, hp, free: take [off_front_ranks, off_free))
    const uint64_t lo = H.mm.off_front_ranks & ~(uint64_t)15, hi = H.mm.off_free;
    printf("pilot region: %.2f MB (image %.1f MB)\n", (hi - lo) / 1e6, nb / 1e6);
    const size_t n = 1ull << 28;
    uint64_t *k, *out; CK(hipMalloc(&k, n * 8)); CK(hipMalloc(&out, n * 8));
    k_gen<<<4096, 256>>>(k, n); CK(hipDeviceSynchronize());
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int* d_stop; uint32_t* d_sink; CK(hipMalloc(&d_stop, 4)); CK(hipMalloc(&d_sink, 4));
    int* h_flag; CK(hipHostMalloc(&h_flag, 4, hipHostMallocDefault));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timed = [&](int rep) {
        float tot = 0;
        for (int r = 0; r < rep; ++r) {
            CK(hipEventRecord(a, s1));
            if (lphq_query_packed(idx, k, n, out, s1)) { printf("query: %s\n", lphq_last_error()); exit(1); }
            CK(hipEventRecord(b, s1)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); tot += ms;
        }
        return tot / rep;
    };
    timed(2);
    printf("baseline: %.3f ms per 2^28 lookups\n", timed(5));
    for (int plain = 0; plain < 2; ++plain)
        for (int g : {8, 16, 32, 64, 128, 256}) {
            CK(hipMemset(d_stop, 0, 4));
            if (plain) k_refresh_plain<<<g, 256, 0, s2>>>((const uint4*)((const uint8_t*)dimg + lo), (hi - lo) / 16, d_stop, 1 << 30, d_sink);
            else k_refresh<<<g, 256, 0, s2>>>((const uint4*)((const uint8_t*)dimg + lo), (hi - lo) / 16, d_stop, 1 << 30, d_sink);
            const float ms = timed(5);
            *h_flag = 1;
            CK(hipMemcpyAsync(d_stop, h_flag, 4, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            printf("refresher %s, %3d workgroups: %.3f ms\n", plain ? "plain loads" : "nt loads   ", g, ms);
        }
    return 0;
}
