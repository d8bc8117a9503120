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

#include "device_query.cuh"
#include "search.hpp"

namespace lphq {
namespace {

constexpr int kWave = 64;
constexpr uint32_t kNoOwner = 0xFFFFFFFFu;
constexpr uint64_t kPilotCap = 1ull << 28;  // unplaceable bucket (duplicate hash, or degenerate tiny table: see builder.cpp): give up here

struct SearchArgs {
    const uint64_t* hs;       // hashes grouped by bucket
    const uint32_t* bstart;   // bucket -> first hash
    const uint32_t* order;    // processing order -> bucket id
    uint32_t* taken;          // committed bitmap, table_size bits
    uint32_t* owner;          // table_size entries, kNoOwner when unclaimed
    uint64_t* pilot;          // per order index: tentative pilot (ready) or the next pilot to try (not ready)
    uint8_t* ready;           // per order index: 1 = pilot was valid against the bitmap when last checked
    uint32_t* flags;          // [0] i_star: first conflicting local index, [1] failed, [2] first not-ready local index of the commit window
    uint64_t seedx;           // murmur_seedx(seed)
    lphq_divisor div;         // table_size
    uint32_t base;            // first unfinished bucket (order index)
    uint32_t n_search;        // look-ahead window: buckets [base, base + n_search) are searched
    uint32_t n_commit;        // commit window: buckets [base, base + n_commit) may be committed this round
    uint32_t max_steps;       // search steps (64 pilots each) per bucket and round
};

__device__ __forceinline__ uint32_t position(uint64_t h, uint64_t hp, const lphq_divisor& d) { return (uint32_t)mod_u64(h ^ hp, d); }
__device__ __forceinline__ bool is_taken(const uint32_t* taken, uint32_t p) { return (taken[p >> 5] >> (p & 31u)) & 1u; }

// exact check of one pilot by one lane: all positions free and pairwise distinct
__device__ bool pilot_ok(const SearchArgs& a, const uint64_t* keys, uint32_t s, uint64_t pilot) {
    const uint64_t hp = murmur8(pilot, a.seedx);
    for (uint32_t j = 0; j < s; ++j) {
        const uint32_t pj = position(keys[j], hp, a.div);
        if (is_taken(a.taken, pj)) return false;
        for (uint32_t t = 0; t < j; ++t)
            if (position(keys[t], hp, a.div) == pj) return false;
    }
    return true;
}

// One wave per bucket of the look-ahead window.  A ready bucket only re-checks its tentative pilot against the grown
// bitmap; otherwise the scan continues from where it stopped, 64 pilots per step, at most max_steps steps per round
// (so one hard bucket never stalls a round: it started many rounds before it reaches the commit window).
__global__ __launch_bounds__(256) void k_search(SearchArgs a) {
    const uint32_t local = (blockIdx.x * 256u + threadIdx.x) / kWave, lane = threadIdx.x & (kWave - 1);
    if (local >= a.n_search) return;
    const uint32_t oi = a.base + local, b = a.order[oi];
    const uint64_t* keys = a.hs + a.bstart[b];
    const uint32_t s = a.bstart[b + 1] - a.bstart[b];
    uint64_t start = a.pilot[oi];
    if (a.ready[oi]) {  // tentative pilot: still free?
        const uint64_t hp = murmur8(start, a.seedx);
        bool hit = false;
        for (uint32_t j = lane; j < s; j += kWave) hit |= is_taken(a.taken, position(keys[j], hp, a.div));
        if (!__any(hit)) return;
    }
    for (uint32_t step = 0; step < a.max_steps; ++step) {
        const uint64_t pilot = start + lane;
        const uint64_t hp = murmur8(pilot, a.seedx);
        bool alive = true;
        for (uint32_t j = 0; j < s; j += 4) {  // 4 independent probes in flight, then one early-exit test
            uint32_t hit = 0;
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t)
                if (j + t < s) hit |= (uint32_t)is_taken(a.taken, position(keys[j + t], hp, a.div));
            alive = alive && !hit;
            if (!__any(alive)) break;
        }
        uint64_t cand = __ballot(alive);
        while (cand) {  // lowest candidate pilot first; in-bucket duplicates are rare, so check them only here
            const int l = __ffsll((long long)cand) - 1;
            bool ok = false;
            if ((int)lane == l) ok = pilot_ok(a, keys, s, pilot);
            if (__any(ok)) {
                if (lane == 0) {
                    a.pilot[oi] = start + (uint64_t)l;
                    a.ready[oi] = 1;
                }
                return;
            }
            cand &= cand - 1;
        }
        start += kWave;
        if (start >= kPilotCap) {
            if (lane == 0) a.flags[1] = 1;
            break;
        }
    }
    if (lane == 0) {  // not found yet: resume here next round
        a.pilot[oi] = start;
        a.ready[oi] = 0;
        if (local < a.n_commit) atomicMin(&a.flags[2], local);
    }
}

#define LPHQ_SEARCH_BUCKET_PROLOGUE()                                                              \
    const uint32_t local = (blockIdx.x * 256u + threadIdx.x) / kWave, lane = threadIdx.x & (kWave - 1); \
    const uint32_t ncommit = min(a.n_commit, a.flags[2]); /* stop at the first bucket still searching */ \
    if (local >= ncommit) return;                                                                  \
    const uint32_t oi = a.base + local, b = a.order[oi];                                           \
    const uint64_t* keys = a.hs + a.bstart[b];                                                     \
    const uint32_t s = a.bstart[b + 1] - a.bstart[b];                                              \
    const uint64_t hp = murmur8(a.pilot[oi], a.seedx)

// every bucket of the commit window claims its positions; the smallest local index wins a contested position
__global__ __launch_bounds__(256) void k_claim(SearchArgs a) {
    LPHQ_SEARCH_BUCKET_PROLOGUE();
    for (uint32_t j = lane; j < s; j += kWave) atomicMin(&a.owner[position(keys[j], hp, a.div)], local);
}

// first bucket (in order) that lost a position to an earlier bucket of the window
__global__ __launch_bounds__(256) void k_validate(SearchArgs a) {
    LPHQ_SEARCH_BUCKET_PROLOGUE();
    bool lost = false;
    for (uint32_t j = lane; j < s; j += kWave) lost |= a.owner[position(keys[j], hp, a.div)] < local;
    if (__any(lost) && lane == 0) atomicMin(&a.flags[0], local);
}

// buckets before i_star are final: take their positions; everybody releases the owner table
__global__ __launch_bounds__(256) void k_commit(SearchArgs a) {
    LPHQ_SEARCH_BUCKET_PROLOGUE();
    const bool final_ = local < a.flags[0];
    for (uint32_t j = lane; j < s; j += kWave) {
        const uint32_t p = position(keys[j], hp, a.div);
        a.owner[p] = kNoOwner;
        if (final_) atomicOr(&a.taken[p >> 5], 1u << (p & 31u));
    }
}

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
    uint32_t *d_bstart = nullptr, *d_order = nullptr, *d_taken = nullptr, *d_owner = nullptr, *d_flags = nullptr;
    uint64_t* d_pilot = nullptr;
    uint8_t* d_ready = nullptr;
    const size_t nb = bstart.size() - 1;
    const size_t taken_u32 = (table_size + 63) / 64 * 2 + 2;
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
    HIP_OK(hipMalloc((void**)&d_flags, 4 * 4));
    HIP_OK(hipMalloc((void**)&d_ready, pilot_by_order.size()));
    HIP_OK(hipMemcpy(d_hs, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_bstart, bstart32.data(), bstart32.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_taken, 0, taken_u32 * 4));
    HIP_OK(hipMemset(d_owner, 0xFF, table_size * 4));
    HIP_OK(hipMemset(d_pilot, 0, pilot_by_order.size() * 8));
    HIP_OK(hipMemset(d_flags, 0, 16));
    HIP_OK(hipMemset(d_ready, 0, pilot_by_order.size()));
    {
        SearchArgs a;
        a.hs = d_hs;
        a.bstart = d_bstart;
        a.order = d_order;
        a.taken = d_taken;
        a.owner = d_owner;
        a.pilot = d_pilot;
        a.ready = d_ready;
        a.flags = d_flags;
        a.seedx = seed ^ (8ull * LPHQ_MURMUR_M);
        memset(&a.div, 0, sizeof a.div);
        a.div.d64 = table_size;
        a.div.d = (uint32_t)table_size;
        a.div.shift = (uint32_t)__builtin_clz((uint32_t)table_size);
        a.div.dn = (uint32_t)table_size << a.div.shift;
        a.div.inv = (uint32_t)(~(uint64_t)0 / a.div.dn - (1ull << 32));
        // tuning aids, clamped to sane ranges (an unclamped look-ahead below the commit window once committed unsearched buckets)
        a.max_steps = getenv("LPHQ_SEARCH_STEPS") ? (uint32_t)std::min(1024, std::max(1, atoi(getenv("LPHQ_SEARCH_STEPS")))) : 32u;
        const uint32_t kLookAhead = getenv("LPHQ_SEARCH_LOOKAHEAD") ? (uint32_t)std::min(1 << 20, std::max(8192, atoi(getenv("LPHQ_SEARCH_LOOKAHEAD")))) : 16384u;
        uint64_t base = 0, rounds = 0, idle_rounds = 0;
        uint32_t window = 1024;
        while (base < n_nonempty) {
            a.base = (uint32_t)base;
            a.n_search = (uint32_t)std::min<uint64_t>(kLookAhead, n_nonempty - base);
            a.n_commit = std::min((uint32_t)std::min<uint64_t>(window, n_nonempty - base), a.n_search);  // only searched buckets may commit
            const uint32_t flags_init[3] = {0xFFFFFFFFu, 0, 0xFFFFFFFFu};
            HIP_OK(hipMemcpyAsync(d_flags, flags_init, sizeof flags_init, hipMemcpyHostToDevice, nullptr));
            k_search<<<(int)((a.n_search * (uint64_t)kWave + 255) / 256), 256>>>(a);
            const int grid = (int)((a.n_commit * (uint64_t)kWave + 255) / 256);
            k_claim<<<grid, 256>>>(a);
            k_validate<<<grid, 256>>>(a);
            k_commit<<<grid, 256>>>(a);
            uint32_t flags[3];
            HIP_OK(hipMemcpy(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost));
            if (flags[1]) {
                err = "pthash search: no pilot below 2^28 for some bucket (duplicate key in the input, or a degenerate tiny key set)";
                goto done;
            }
            const uint32_t in_window = std::min(a.n_commit, flags[2]);
            const uint32_t committed = std::min(in_window, flags[0]);
            base += committed;
            ++rounds;
            idle_rounds = committed ? 0 : idle_rounds + 1;
            if (idle_rounds > (kPilotCap / kWave) / a.max_steps + 16) {
                err = "gpu pilot search: no progress";
                goto done;
            }
            // adapt the commit window to the conflict rate only (a bucket that is still searching stops the prefix too,
            // but a smaller window would not help it): grow while conflict-free, shrink when a conflict cut it short
            const bool conflict = flags[0] < in_window;
            if (!conflict && window < 8192) window *= 2;
            else if (conflict && committed < a.n_commit / 4 && window > 256) window /= 2;
        }
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
        if (ones != hs.size()) {
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
    if (d_flags) (void)hipFree(d_flags);
    if (d_ready) (void)hipFree(d_ready);
    if (prev >= 0) (void)hipSetDevice(prev);
    return ok;
}

}  // namespace lphq
