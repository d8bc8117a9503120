// Experiment (scratch tool, not part of the product): FOUR k-mers per lane in the random kernel, the first hop batched.
//   V1  four minimizer scans + PTHash key hashes, then the four pilot-codeword loads issued back to back; the rest of each chain (hp, position,
//       block or tail line, decode) runs one k-mer at a time
//   V2  V1 with the lane's four k-mers put in dense-bucket-first order before the loads (a 4-element sorting network on the dense flag), so that
//       the loads of step j of a wave go to the 30 % of the pilot table that 60 % of the keys use as often as possible
//   V3  every stage four wide: pilot loads, hp loads, block loads (4 x 128 B in registers), decodes
// Codes are compared with the product kernel's.  VERDICT r02 item 7: "one more attempt at the pilot-line misses, then stop".
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilphash_amd/csrc tools/packed4_exp.hip -o tools/packed4_exp -Llphash_amd -llphq -Wl,-rpath,$PWD/lphash_amd
// run (after a bench.py run has left the C2 index in the cache): tools/packed4_exp [index.lph] [u]      (u: unpartitioned file)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "device_query.cuh"
#include "../include/lphq.h"
using namespace lphq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Chain {  // one k-mer on its way
    uint64_t lo, x;
    uint32_t p, idx;
};

template <int KIND>
__device__ __forceinline__ uint64_t finish_chain(const lphq_image_header& H, const uint8_t* __restrict__ img, const Chain& c, uint64_t two, const PhfRank& rk) {
    const lphq_phf_desc& f = H.mm;
    const uint32_t pos = (uint32_t)mod_u64(c.x ^ phf_load_hp(img, f, two, rk), f.div_table);
    const uint32_t nk = (uint32_t)f.num_keys;
    const bool tail = pos >= nk;
    uint64_t tv = 0;
    if (tail) tv = ((const uint64_t*)(img + H.off_tail))[pos - nk];
    const uint32_t h = tail ? 0u : pos;
    LookupCtx ctx;
    if (KIND == 0) {
        uint32_t type, size, ps;
        uint64_t base;
        uni_lookup<4>(img, H.off_uni, H.W, h, type, base, size, ps);
        ctx = ctx_partitioned(H, type, base, size, ps);
    } else {
        uint64_t prefix;
        uint32_t size, ps;
        pair_lookup<4>(img, H.off_upair, h, prefix, size, ps);
        ctx = ctx_unpartitioned(H, prefix, size, ps);
    }
    if (tail) {
        ctx.base = tv & LPHQ_TAIL_BASE_MASK;
        ctx.mode = (uint32_t)(tv >> LPHQ_TAIL_MODE_SHIFT);
    }
    return finish_hval<false>(img, H, ctx, c.lo, 0, c.p);
}

template <int KIND, int V>
__global__ __launch_bounds__(256) void k_packed4(const lphq_image_header H, const uint8_t* __restrict__ img, const uint64_t* __restrict__ kmers, size_t n,
                                                 uint64_t* __restrict__ out) {
    const size_t quarter = n / 4;  // k-mer j of a lane: index i + j * quarter (every load and store of a wave is coalesced)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= quarter) return;
    const uint64_t T = 0x9999999999999800ULL;
    Chain c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j].idx = (uint32_t)j;
        c[j].lo = __builtin_nontemporal_load(kmers + i + j * quarter);
        uint64_t mm;
        minimizer_of<false>(c[j].lo, 0, H.k, H.m, H.W, murmur_seedx(H.mm_seed), mm, c[j].p);
        c[j].x = murmur8(mm, murmur_seedx(H.mm.seed));
    }
    if (V == 2) {  // dense buckets first: sorting network on (x >= T), stable enough for the purpose
        auto cswap = [&](int a, int b) {
            const bool sw = (c[a].x >= T) && (c[b].x < T);
            const Chain ta = c[a], tb = c[b];
            c[a].lo = sw ? tb.lo : ta.lo; c[a].x = sw ? tb.x : ta.x; c[a].p = sw ? tb.p : ta.p; c[a].idx = sw ? tb.idx : ta.idx;
            c[b].lo = sw ? ta.lo : tb.lo; c[b].x = sw ? ta.x : tb.x; c[b].p = sw ? ta.p : tb.p; c[b].idx = sw ? ta.idx : tb.idx;
        };
        cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
    }
    uint64_t two[4];
    PhfRank rk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) two[j] = phf_load_code(img, H.mm, c[j].x, rk[j]);  // the four pilot loads, back to back
    if (V != 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(finish_chain<KIND>(H, img, c[j], two[j], rk[j]), out + i + c[j].idx * quarter);
    } else {
        const lphq_phf_desc& f = H.mm;
        uint64_t hp[4], tv[4];
        uint32_t h[4];
        bool tail[4];
        UniRegs<4> ur[4];
        PairRegs pr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) hp[j] = phf_load_hp(img, f, two[j], rk[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t pos = (uint32_t)mod_u64(c[j].x ^ hp[j], f.div_table), nk = (uint32_t)f.num_keys;
            tail[j] = pos >= nk;
            tv[j] = 0;
            if (tail[j]) tv[j] = ((const uint64_t*)(img + H.off_tail))[pos - nk];
            h[j] = tail[j] ? 0u : pos;
            if (KIND == 0) uni_load<4>(img, H.off_uni, h[j], ur[j]);
            else pair_load(img, H.off_upair, h[j], pr[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            LookupCtx ctx;
            if (KIND == 0) {
                uint32_t type, size, ps;
                uint64_t base;
                uni_decode<4>(ur[j], H.W, h[j], type, base, size, ps);
                ctx = ctx_partitioned(H, type, base, size, ps);
            } else {
                uint64_t prefix;
                uint32_t size, ps;
                pair_decode<4>(pr[j], h[j], prefix, size, ps);
                ctx = ctx_unpartitioned(H, prefix, size, ps);
            }
            if (tail[j]) {
                ctx.base = tv[j] & LPHQ_TAIL_BASE_MASK;
                ctx.mode = (uint32_t)(tv[j] >> LPHQ_TAIL_MODE_SHIFT);
            }
            __builtin_nontemporal_store(finish_hval<false>(img, H, ctx, c[j].lo, 0, c[j].p), out + i + j * quarter);
        }
    }
}

__device__ uint64_t splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__global__ void k_gen(uint64_t* kmers, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) kmers[i] = splitmix(i) >> 2;
}

template <int KIND, int V>
float run(const lphq_image_header& H, const uint8_t* dimg, const uint64_t* k, size_t n, uint64_t* out, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = (int)((n / 4 + 255) / 256);
    k_packed4<KIND, V><<<grid, 256>>>(H, dimg, k, n, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) k_packed4<KIND, V><<<grid, 256>>>(H, dimg, k, n, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const bool unpart = argc > 2 && argv[2][0] == 'u';
    const char* path = argc > 1 ? argv[1] : "/tmp/lphq_bench_cache/synth_seed1_p_k31_m20_n100000000_b64.lph";
    lphq_index* idx; if (lphq_open(path, unpart ? LPHQ_UNPARTITIONED : LPHQ_PARTITIONED, 64, &idx)) { printf("open: %s\n", lphq_last_error()); return 1; }
    if (lphq_attach(idx, 0)) { printf("attach: %s\n", lphq_last_error()); return 1; }
    size_t nb; lphq_image_bytes(idx, &nb); std::vector<uint8_t> himg(nb); lphq_image_copy(idx, himg.data(), nb);
    lphq_image_header H; memcpy(&H, himg.data(), sizeof H);
    const void* dimg; size_t dn; lphq_device_image(idx, &dimg, &dn);
    const size_t n = 1ull << 28;
    uint64_t *k, *out, *ref; CK(hipMalloc(&k, n * 8)); CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&ref, n * 8));
    k_gen<<<4096, 256>>>(k, n); CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    lphq_query_packed(idx, k, n, ref, nullptr); CK(hipDeviceSynchronize());
    const int mode = argc > 3 ? atoi(argv[3]) : -1;  // one variant only (counter passes)
    if (mode < 0 || mode == 0) {
        CK(hipEventRecord(a));
        for (int r = 0; r < 10; ++r) lphq_query_packed(idx, k, n, ref, nullptr);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("product k_query_packed (one k-mer per lane): %.3f ms per 2^28 lookups (%s, random non-member k-mers)\n", ms / 10, unpart ? "unpartitioned" : "partitioned");
    }
    auto check = [&](const char* what) {
        std::vector<uint64_t> x(n), y(n);
        CK(hipMemcpy(x.data(), out, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), ref, n * 8, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += x[i] != y[i];
        printf("  %s: %zu of %zu codes differ from the product kernel\n", what, bad, n);
    };
#define VARIANT(V, label) if (mode < 0 || mode == V) { CK(hipMemset(out, 0, n * 8)); \
        printf("%s: %.3f ms\n", label, unpart ? run<1, V>(H, (const uint8_t*)dimg, k, n, out, 10) : run<0, V>(H, (const uint8_t*)dimg, k, n, out, 10)); if (mode < 0) check(label); }
    VARIANT(1, "V1 four per lane, pilot loads batched");
    VARIANT(2, "V2 the same, dense buckets first");
    VARIANT(3, "V3 every stage four wide");
    return 0;
}
