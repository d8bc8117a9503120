// search.hip -- PTHash pilot search on the GPU (index construction).
//
// PTHash v1 search: buckets in non-increasing size order; the pilot of a bucket is the FIRST
// pilot p >= 0 such that the positions (hash ^ Murmur(p, seed)) mod table_size of its keys are free and pairwise
// distinct; the positions are then taken.  That is a sequential dependency through the `taken` bitmap, and it is
// 95 % of the CPU builder's time (profiles/r01_builder_cpu_stages.txt).
//
// Here a window of consecutive buckets is searched speculatively in parallel, one wave per bucket, 64 pilots per
// step, against the committed bitmap.  Every bucket then claims its positions with atomicMin(owner[pos], index);
// the first bucket (in order) that lost a position to an earlier bucket of the window marks the end of the prefix
// that is committed this round; later buckets keep their tentative pilot and are re-validated against the grown
// bitmap next round.  A committed bucket's pilot is the first one valid against (committed bits) + (positions of
// all earlier buckets), i.e. exactly the sequential first-fit answer, so the result -- and the serialized file --
// is identical to the CPU builder's for any window size.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "search.hpp"
#include "search_rounds.hpp"

namespace lphq {
namespace {
using namespace search;

#define HIP_OK(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) {                                                   \
            err = std::string(#call) + ": " + hipGetErrorString(e_);              \
            goto done;                                                            \
        }                                                                         \
    } while (0)

}  // namespace

bool gpu_pilot_search(int device, const std::vector<uint64_t>& hs, const std::vector<uint64_t>& bstart, const std::vector<uint32_t>& order,
                      uint64_t n_nonempty, uint64_t table_size, uint64_t seed, std::vector<uint64_t>& pilots_by_bucket,
                      std::vector<uint64_t>& taken_words, std::string& err) {
    bool ok = false;
    int prev = -1;
    uint64_t* d_hs = nullptr;
    uint32_t *d_bstart = nullptr, *d_order = nullptr, *d_taken = nullptr, *d_owner = nullptr, *d_ctl = nullptr;
    uint64_t* d_pilot = nullptr;
    uint8_t* d_ready = nullptr;
    const size_t nb = bstart.size() - 1;
    const size_t taken_u32 = (table_size + 63) / 64 * 2 + 2;
    uint64_t ones_before = 0;
    std::vector<uint32_t> bstart32(bstart.begin(), bstart.end());
    std::vector<uint64_t> pilot_by_order(n_nonempty ? n_nonempty : 1, 0);
    if (table_size == 0 || table_size > 0xFFFFFFFFull || hs.size() > 0xFFFFFFFFull) {
        err = "gpu pilot search: table too large for the 32-bit position path";
        return false;
    }
    (void)hipGetDevice(&prev);
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
            err = "gpu pilot search: no such HIP device";
            return false;
        }
    }
    HIP_OK(hipSetDevice(device));
    HIP_OK(hipMalloc((void**)&d_hs, (hs.size() ? hs.size() : 1) * 8));
    HIP_OK(hipMalloc((void**)&d_bstart, bstart32.size() * 4));
    HIP_OK(hipMalloc((void**)&d_order, (order.size() ? order.size() : 1) * 4));
    HIP_OK(hipMalloc((void**)&d_taken, taken_u32 * 4));
    HIP_OK(hipMalloc((void**)&d_owner, table_size * 4));
    HIP_OK(hipMalloc((void**)&d_pilot, pilot_by_order.size() * 8));
    HIP_OK(hipMalloc((void**)&d_ctl, kCtlWords * 4));
    HIP_OK(hipMalloc((void**)&d_ready, pilot_by_order.size()));
    HIP_OK(hipMemcpy(d_hs, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_bstart, bstart32.data(), bstart32.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice));
    if (taken_words.empty()) {
        HIP_OK(hipMemset(d_taken, 0, taken_u32 * 4));
    } else {  // the table an earlier piece of this search left
        std::vector<uint32_t> t32(taken_u32, 0);
        for (size_t i = 0; i < taken_u32 && i / 2 < taken_words.size(); ++i) t32[i] = (uint32_t)(taken_words[i / 2] >> (32 * (i & 1)));
        for (uint64_t w : taken_words) ones_before += (uint64_t)__builtin_popcountll(w);
        HIP_OK(hipMemcpy(d_taken, t32.data(), taken_u32 * 4, hipMemcpyHostToDevice));
    }
    HIP_OK(hipMemset(d_owner, 0xFF, table_size * 4));
    HIP_OK(hipMemset(d_pilot, 0, pilot_by_order.size() * 8));
    HIP_OK(hipMemset(d_ready, 0, pilot_by_order.size()));
    {
        SearchArgs a;
        memset(&a, 0, sizeof a);
        a.hs = d_hs;
        a.bstart = d_bstart;
        a.order = d_order;
        a.taken = d_taken;
        a.owner = d_owner;
        a.pilot = d_pilot;
        a.ready = d_ready;
        a.ctl = d_ctl;
        a.seedx = seed ^ (8ull * LPHQ_MURMUR_M);
        a.div = divisor_u32(table_size);
        struct HipDev {  // the product backend of search_rounds.hpp: default stream, one synchronous read-back per batch of rounds
            uint32_t* d_ctl;
            bool set_ctl(const uint32_t (&c)[kCtlWords]) { return hipMemcpy(d_ctl, c, sizeof c, hipMemcpyHostToDevice) == hipSuccess; }
            bool get_ctl(uint32_t (&c)[kCtlWords]) { return hipMemcpy(c, d_ctl, sizeof c, hipMemcpyDeviceToHost) == hipSuccess && hipGetLastError() == hipSuccess; }
            void launch_search(const SearchArgs& x, int g) { k_search<<<g, 256>>>(x); }
            void launch_claim(const SearchArgs& x, int g) { k_claim<<<g, 256>>>(x); }
            void launch_validate(const SearchArgs& x, int g) { k_validate<<<g, 256>>>(x); }
            void launch_commit(const SearchArgs& x, int g) { k_commit<<<g, 256>>>(x); }
            void launch_advance(const SearchArgs& x) { k_advance<<<1, 64>>>(x); }
        } dev{d_ctl};
        const uint32_t steps = getenv("LPHQ_SEARCH_STEPS") ? (uint32_t)std::max(1, atoi(getenv("LPHQ_SEARCH_STEPS"))) : 16u;                    // tuning aids (16 steps x 256 pilots measured best)
        const uint32_t look_ahead = getenv("LPHQ_SEARCH_LOOKAHEAD") ? (uint32_t)std::max(1, atoi(getenv("LPHQ_SEARCH_LOOKAHEAD"))) : 16384u;
        const uint32_t per_sync = getenv("LPHQ_SEARCH_ROUNDS_PER_SYNC") ? (uint32_t)std::max(1, atoi(getenv("LPHQ_SEARCH_ROUNDS_PER_SYNC"))) : 16u;
        uint64_t rounds = 0;
        if (!run_rounds(dev, a, n_nonempty, look_ahead, steps, per_sync, err, &rounds)) goto done;
        if (getenv("LPHQ_BUILD_VERBOSE")) fprintf(stderr, "[lphq_build] gpu pilot search: %llu buckets in %llu rounds\n", (unsigned long long)n_nonempty, (unsigned long long)rounds);
    }
    HIP_OK(hipMemcpy(pilot_by_order.data(), d_pilot, n_nonempty * 8, hipMemcpyDeviceToHost));
    {
        std::vector<uint32_t> t32(taken_u32);
        HIP_OK(hipMemcpy(t32.data(), d_taken, taken_u32 * 4, hipMemcpyDeviceToHost));
        taken_words.assign((table_size + 63) / 64 + 1, 0);
        for (size_t i = 0; i + 1 < taken_u32 && i / 2 < taken_words.size(); ++i) taken_words[i / 2] |= (uint64_t)t32[i] << (32 * (i & 1));
    }
    {
        uint64_t ones = 0;
        for (uint64_t w : taken_words) ones += (uint64_t)__builtin_popcountll(w);
        if (ones != ones_before + hs.size()) {
            err = "gpu pilot search: internal error (occupied slots != keys)";
            goto done;
        }
    }
    pilots_by_bucket.assign(nb, 0);
    for (uint64_t oi = 0; oi < n_nonempty; ++oi) pilots_by_bucket[order[oi]] = pilot_by_order[oi];
    ok = true;
done:
    if (d_hs) (void)hipFree(d_hs);
    if (d_bstart) (void)hipFree(d_bstart);
    if (d_order) (void)hipFree(d_order);
    if (d_taken) (void)hipFree(d_taken);
    if (d_owner) (void)hipFree(d_owner);
    if (d_pilot) (void)hipFree(d_pilot);
    if (d_ctl) (void)hipFree(d_ctl);
    if (d_ready) (void)hipFree(d_ready);
    if (prev >= 0) (void)hipSetDevice(prev);
    return ok;
}

}  // namespace lphq
