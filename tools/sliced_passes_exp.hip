// Experiment (scratch tool, not part of the product): random lookups in BUCKET-SLICED PASSES, so that the pilot lines of a pass fit the per-XCD L2.
//   pass A  every k-mer: minimizer + PTHash key hash x; keys of DENSE buckets (60 % of the keys, front dictionary = 30 % of the codewords) are looked up at once,
//           the others are appended to one of S queues by sparse-bucket range (wave-aggregated atomics, 16 B per item)
//   pass B  one launch per queue: pilot (its slice of the back dictionary) -> position -> block -> code -> out[index]
// Timing-only upper bound for "pilot lines always hit L2": C2 7.27 -> 5.63 ms (profiles/r02_pilot_hit_upper_bound.txt).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilphash_amd/csrc tools/sliced_passes_exp.hip -o tools/sliced_passes_exp -Llphash_amd -llphq -Wl,-rpath,$PWD/lphash_amd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "device_query.cuh"
#include "../include/lphq.h"
using namespace lphq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Item { uint64_t x; uint32_t idx; uint32_t p; };

template <int S>
__global__ __launch_bounds__(256) void k_pass_a(const lphq_image_header H, const uint8_t* __restrict__ img, const uint64_t* __restrict__ kmers, size_t n,
                                                uint64_t* __restrict__ out, Item* __restrict__ queues, size_t qcap, unsigned long long* __restrict__ qn) {
    const uint64_t T = 0x9999999999999800ULL;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    uint64_t lo = live ? __builtin_nontemporal_load(kmers + i) : 0, mm;
    uint32_t p;
    minimizer_of<false>(lo, 0, H.k, H.m, H.W, murmur_seedx(H.mm_seed), mm, p);
    const uint64_t x = murmur8(mm, murmur_seedx(H.mm.seed));
    const bool dense = x < T;
    if (live && dense) {
        const uint32_t h = (uint32_t)phf_position(img, H.mm, x);
        uint32_t type, size, pos;
        uint64_t base;
        uni_lookup<4>(img, H.off_uni, H.W, h, type, base, size, pos);
        const LookupCtx c = ctx_partitioned(H, type, base, size, pos);
        __builtin_nontemporal_store(finish_hval<false>(img, H, c, lo, 0, p), out + i);
    }
    // sparse: slice by bucket range of the back dictionary
    uint32_t slice = 0;
    if (!dense) {
        const uint64_t bs = mod_u64(x, H.mm.div_sparse);
        slice = (uint32_t)(bs * S / H.mm.num_sparse);
    }
    // workgroup-private regions, no global atomics: items of slice s of workgroup b go to queues[(s * gridDim.x + b) * 256 ...], their number to cnt
    __shared__ uint32_t sm_cnt[S];
    if (threadIdx.x < S) sm_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const bool mine = live && !dense && slice == (uint32_t)s;
        const uint64_t bal = __ballot(mine);
        if (!bal) continue;
        uint32_t base = 0;
        const int leader = __ffsll((long long)bal) - 1;
        if ((int)lane == leader) base = atomicAdd(&sm_cnt[s], (uint32_t)__popcll(bal));
        base = __shfl(base, leader, 64);
        if (mine) queues[((size_t)s * gridDim.x + blockIdx.x) * 256 + base + __popcll(bal & ((1ull << lane) - 1ull))] = Item{x, (uint32_t)i, p};
    }
    __syncthreads();
    if (threadIdx.x < S) ((uint32_t*)qn)[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = sm_cnt[threadIdx.x];
}
// one workgroup per group of G regions of one slice: the regions' items are enumerated densely over the lanes
template <int G>
__global__ __launch_bounds__(256) void k_pass_b(const lphq_image_header H, const uint8_t* __restrict__ img, const uint64_t* __restrict__ kmers,
                                                uint64_t* __restrict__ out, const Item* __restrict__ queue, const uint32_t* __restrict__ cnt, uint32_t n_regions) {
    __shared__ uint32_t sm_pre[G + 1];
    const uint32_t r0 = blockIdx.x * G;
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int g = 0; g < G; ++g) {
            sm_pre[g] = acc;
            acc += (r0 + g < n_regions) ? cnt[r0 + g] : 0u;
        }
        sm_pre[G] = acc;
    }
    __syncthreads();
    const uint32_t total = sm_pre[G];
    for (uint32_t j = threadIdx.x; j < total; j += 256) {
        uint32_t g = 0;
#pragma unroll
        for (int t = 1; t < G; ++t) g += (j >= sm_pre[t]);
        const Item it = queue[(size_t)(r0 + g) * 256 + (j - sm_pre[g])];
        const uint32_t h = (uint32_t)phf_position(img, H.mm, it.x);
        uint32_t type, size, pos;
        uint64_t base;
        uni_lookup<4>(img, H.off_uni, H.W, h, type, base, size, pos);
        const LookupCtx c = ctx_partitioned(H, type, base, size, pos);
        uint64_t lo = 0;
        if (c.mode == 2u) lo = kmers[it.idx];
        __builtin_nontemporal_store(finish_hval<false>(img, H, c, lo, 0, it.p), out + it.idx);
    }
}
// XCD-SPATIAL variant of pass B: ONE launch, workgroup w serves slice w % 8 -- with the round-robin dispatch of workgroups over the 8 XCDs every XCD
// reads only its eighth of the back dictionary (0.3 MB), which then stays in that XCD's L2.
template <int G>
__global__ __launch_bounds__(256) void k_pass_b_xcd(const lphq_image_header H, const uint8_t* __restrict__ img, const uint64_t* __restrict__ kmers,
                                                    uint64_t* __restrict__ out, const Item* __restrict__ queues, const uint32_t* __restrict__ cnts, uint32_t n_regions) {
    __shared__ uint32_t sm_pre[G + 1];
    const uint32_t slice = blockIdx.x & 7u, r0 = (blockIdx.x >> 3) * G;
    const Item* queue = queues + (size_t)slice * n_regions * 256;
    const uint32_t* cnt = cnts + (size_t)slice * n_regions;
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int g = 0; g < G; ++g) {
            sm_pre[g] = acc;
            acc += (r0 + g < n_regions) ? cnt[r0 + g] : 0u;
        }
        sm_pre[G] = acc;
    }
    __syncthreads();
    const uint32_t total = sm_pre[G];
    for (uint32_t j = threadIdx.x; j < total; j += 256) {
        uint32_t g = 0;
#pragma unroll
        for (int t = 1; t < G; ++t) g += (j >= sm_pre[t]);
        const Item it = queue[(size_t)(r0 + g) * 256 + (j - sm_pre[g])];
        const uint32_t h = (uint32_t)phf_position(img, H.mm, it.x);
        uint32_t type, size, pos;
        uint64_t base;
        uni_lookup<4>(img, H.off_uni, H.W, h, type, base, size, pos);
        const LookupCtx c = ctx_partitioned(H, type, base, size, pos);
        uint64_t lo = 0;
        if (c.mode == 2u) lo = kmers[it.idx];
        __builtin_nontemporal_store(finish_hval<false>(img, H, c, lo, 0, it.p), out + it.idx);
    }
}
float run_xcd(const lphq_image_header& H, const uint8_t* dimg, const uint64_t* k, size_t n, uint64_t* out, Item* q, size_t qcap, unsigned long long* qn, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned nwg = (unsigned)((n + 255) / 256);
    constexpr int G = 32;  // ~0.4 * 256 / 8 items per region -> ~400 per group
    auto once = [&] {
        k_pass_a<8><<<nwg, 256>>>(H, dimg, k, n, out, q, qcap, qn);
        k_pass_b_xcd<G><<<8 * ((nwg + G - 1) / G), 256>>>(H, dimg, k, out, q, (const uint32_t*)qn, nwg);
    };
    once(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) once();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
template <int S>
float run(const lphq_image_header& H, const uint8_t* dimg, const uint64_t* k, size_t n, uint64_t* out, Item* q, size_t qcap, unsigned long long* qn, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned nwg = (unsigned)((n + 255) / 256);
    constexpr int G = 4 * S;  // ~0.4 * 256 / S items per region -> ~400 per group
    auto once = [&] {
        k_pass_a<S><<<nwg, 256>>>(H, dimg, k, n, out, q, qcap, qn);
        for (int s = 0; s < S; ++s) k_pass_b<G><<<(nwg + G - 1) / G, 256>>>(H, dimg, k, out, q + (size_t)s * nwg * 256, (const uint32_t*)qn + (size_t)s * nwg, nwg);
    };
    once(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) once();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
__device__ __forceinline__ uint64_t splitmix(uint64_t x) { x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL; return x ^ (x >> 31); }
__global__ void k_gen(uint64_t* kmers, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) kmers[i] = splitmix(i) >> 2;
}
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/lphq_bench_cache/synth_seed1_p_k31_m20_n100000000_b64.lph";
    lphq_index* idx; if (lphq_open(path, LPHQ_PARTITIONED, 64, &idx)) { printf("open: %s\n", lphq_last_error()); return 1; }
    if (lphq_attach(idx, 0)) { printf("attach: %s\n", lphq_last_error()); return 1; }
    size_t nb; lphq_image_bytes(idx, &nb); std::vector<uint8_t> himg(nb); lphq_image_copy(idx, himg.data(), nb);
    lphq_image_header H; memcpy(&H, himg.data(), sizeof H);
    const void* dimg; size_t dn; lphq_device_image(idx, &dimg, &dn);
    const size_t n = 1ull << 28, qcap = n;
    uint64_t *k, *out, *ref; CK(hipMalloc(&k, n * 8)); CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&ref, n * 8));
    Item* q; CK(hipMalloc(&q, 8 * qcap * sizeof(Item))); unsigned long long* qn; CK(hipMalloc(&qn, 8 * (n / 256) * 4 + 64));
    k_gen<<<4096, 256>>>(k, n); CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    lphq_query_packed(idx, k, n, ref, nullptr); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) lphq_query_packed(idx, k, n, ref, nullptr);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("product k_query_packed: %.3f ms per 2^28 lookups (non-member random k-mers)\n", ms / 5);
    auto check = [&](const char* what) {
        std::vector<uint64_t> x(1 << 22), y(1 << 22);
        CK(hipMemcpy(x.data(), out, x.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), ref, y.size() * 8, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < x.size(); ++i) bad += x[i] != y[i];
        printf("  %s: %zu of %zu sampled codes differ from the product kernel\n", what, bad, x.size());
    };
    CK(hipMemset(out, 0, n * 8)); printf("sliced, S = 1: %.3f ms\n", run<1>(H, (const uint8_t*)dimg, k, n, out, q, qcap, qn, 5)); check("S=1");
    CK(hipMemset(out, 0, n * 8)); printf("sliced, S = 2: %.3f ms\n", run<2>(H, (const uint8_t*)dimg, k, n, out, q, qcap, qn, 5)); check("S=2");
    CK(hipMemset(out, 0, n * 8)); printf("sliced, S = 3: %.3f ms\n", run<3>(H, (const uint8_t*)dimg, k, n, out, q, qcap, qn, 5)); check("S=3");
    CK(hipMemset(out, 0, n * 8)); printf("sliced, S = 4: %.3f ms\n", run<4>(H, (const uint8_t*)dimg, k, n, out, q, qcap, qn, 5)); check("S=4");
    CK(hipMemset(out, 0, n * 8)); printf("XCD-spatial, 8 slices in one launch (workgroup w -> slice w %% 8): %.3f ms\n", run_xcd(H, (const uint8_t*)dimg, k, n, out, q, qcap, qn, 5)); check("XCD-spatial");
    return 0;
}
