// Ablation of the random-batch kernel on a real index image (scratch tool, not part of the product):
//   full      = product device function query_kmer
//   hashonly  = minimizer_of only (W Murmurs + argmin)
//   lookup    = lookup_ctx + finish on precomputed (minimizer, p)
//   phf / wt  = single stages
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilphash_amd/csrc tools/ablate.hip -o tools/ablate -Llphash_amd -llphq -Wl,-rpath,$PWD/lphash_amd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "device_query.cuh"
#include "../include/lphq.h"
using namespace lphq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t splitmix(uint64_t x) { x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31); }

__global__ void k_gen(uint64_t* kmers, size_t n) {  // random (non-member) k-mers: same cost profile except type mix
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) kmers[i] = splitmix(i) >> 2;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_abl(const lphq_image_header H, const uint8_t* __restrict__ img, const uint64_t* __restrict__ kmers, const uint64_t* __restrict__ mms,
                                             size_t n, uint64_t* __restrict__ out) {
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (; i < n; i += st) {
        uint64_t lo = kmers[i];
        uint64_t r;
        if (MODE == 0) r = query_kmer<0, false, 4>(img, H, lo, 0);
        else if (MODE == 1) { uint64_t mm; uint32_t p; minimizer_of<false>(lo, 0, H.k, H.m, H.W, murmur_seedx(H.mm_seed), mm, p); r = mm ^ p; }
        else if (MODE == 2) { uint64_t mm = mms[i]; LookupCtx c = lookup_ctx<0, false, 4>(img, H, mm >> 8); r = finish_hval<false>(img, H, c, lo, 0, (uint32_t)mm & 255u); }
        else if (MODE == 3) { uint64_t mm = mms[i]; r = phf_position(img, H.mm, murmur8(mm >> 8, murmur_seedx(H.mm.seed))); }
        else if (MODE == 4) { uint64_t mm = mms[i]; uint32_t t, sz, ps; uint64_t bs; uni_lookup<4>(img, H.off_uni, H.W, (uint32_t)((mm >> 8) % H.distinct_minimizers), t, bs, sz, ps); r = t + bs + sz + ps; }
        else if (MODE == 5) { uint64_t mm = mms[i]; uint32_t h = (uint32_t)phf_position(img, H.mm, murmur8(mm >> 8, murmur_seedx(H.mm.seed))); uint32_t t, sz, ps; uint64_t bs; uni_lookup<4>(img, H.off_uni, H.W, h, t, bs, sz, ps); r = t + bs + sz + ps; }
        else { uint64_t mm; uint32_t p; minimizer_of<false>(lo, 0, H.k, H.m, H.W, murmur_seedx(H.mm_seed), mm, p); r = (mm << 8) | p; }
        out[i] = r;
    }
}
template <int MODE> float run(const lphq_image_header& H, const uint8_t* img, const uint64_t* k, const uint64_t* mm, size_t n, uint64_t* out, int grid) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_abl<MODE><<<grid, 256>>>(H, img, k, mm, n, out); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 3; ++i) k_abl<MODE><<<grid, 256>>>(H, img, k, mm, n, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 3;
}
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/lphq_bench_cache/synth_seed1_p_k31_m20_n100000000_b64.lph";
    lphq_index* idx; if (lphq_open(path, LPHQ_PARTITIONED, 64, &idx)) { printf("open: %s\n", lphq_last_error()); return 1; }
    if (lphq_attach(idx, 0)) { printf("attach: %s\n", lphq_last_error()); return 1; }
    size_t nb; lphq_image_bytes(idx, &nb); std::vector<uint8_t> himg(nb); lphq_image_copy(idx, himg.data(), nb);
    lphq_image_header H; memcpy(&H, himg.data(), sizeof H);
    const void* dimg; size_t dn; lphq_device_image(idx, &dimg, &dn);
    size_t n = 1ull << 27;
    uint64_t *k, *mm, *out; CK(hipMalloc(&k, n * 8)); CK(hipMalloc(&mm, n * 8)); CK(hipMalloc(&out, n * 8));
    k_gen<<<4096, 256>>>(k, n);
    k_abl<6><<<4096, 256>>>(H, (const uint8_t*)dimg, k, mm, n, mm);  // precompute (mm<<8|p)
    CK(hipDeviceSynchronize());
    for (int grid : {2048, 4096, 8192}) {
        printf("grid %d: full %.3f ms | hashonly %.3f | lookup(given mm) %.3f | phf %.3f | block %.3f | phf+block %.3f   (n=2^27)\n", grid,
               run<0>(H, (const uint8_t*)dimg, k, mm, n, out, grid), run<1>(H, (const uint8_t*)dimg, k, mm, n, out, grid), run<2>(H, (const uint8_t*)dimg, k, mm, n, out, grid),
               run<3>(H, (const uint8_t*)dimg, k, mm, n, out, grid), run<4>(H, (const uint8_t*)dimg, k, mm, n, out, grid), run<5>(H, (const uint8_t*)dimg, k, mm, n, out, grid));
    }
    return 0;
}
