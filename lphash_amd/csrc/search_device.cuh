// search_device.cuh -- kernels of the GPU pilot search (see search.hip for the algorithm; host side of the rounds: search_rounds.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "device_query.cuh"

namespace lphq {
namespace search {


constexpr int kWave = 64;
constexpr uint32_t kNoOwner = 0xFFFFFFFFu;
constexpr uint64_t kPilotCap = 1ull << 28;  // unplaceable bucket (duplicate hash, or degenerate tiny table: see builder.cpp): give up here

// Round control lives in device memory so that several rounds can be enqueued without a host round trip: every kernel reads
// its window from it and k_advance (one thread, last kernel of a round) moves the window on.
enum : uint32_t {
    kIStar = 0,      // first local index of the commit window that lost a position to an earlier bucket (k_validate)
    kFailed,         // a bucket ran past kPilotCap
    kNotReady,       // first local index of the commit window whose bucket is still searching (k_search)
    kBase,           // first unfinished bucket (order index)
    kNSearch,        // look-ahead window: buckets [base, base + n_search) are searched
    kNCommit,        // commit window: buckets [base, base + n_commit) may be committed this round
    kWindow,         // adaptive commit-window size
    kDone,           // base reached n_nonempty
    kRounds,         // rounds executed
    kIdle,           // consecutive rounds without a commit
    kStuck,          // kIdle exceeded kMaxIdle
    kNNonEmpty,      // (constant) buckets to place
    kLookAhead,      // (constant)
    kMaxIdle,        // (constant)
    kCtlWords = 16
};

struct SearchArgs {
    const uint64_t* hs;       // hashes grouped by bucket
    const uint32_t* bstart;   // bucket -> first hash
    const uint32_t* order;    // processing order -> bucket id
    uint32_t* taken;          // committed bitmap, table_size bits
    uint32_t* owner;          // table_size entries, kNoOwner when unclaimed
    uint64_t* pilot;          // per order index: tentative pilot (ready) or the next pilot to try (not ready)
    uint8_t* ready;           // per order index: 1 = pilot was valid against the bitmap when last checked
    uint32_t* ctl;            // kCtlWords control words (above)
    uint64_t seedx;           // murmur_seedx(seed)
    lphq_divisor div;         // table_size
    uint32_t max_steps;       // search steps (256 pilots each) per bucket and round
};
__device__ __forceinline__ bool search_over(const uint32_t* ctl) { return (ctl[kDone] | ctl[kFailed] | ctl[kStuck]) != 0u; }
// k_search itself sets kFailed, so it must not branch on it (the threads of a workgroup could disagree ahead of a barrier);
// kDone and kStuck only change between kernels
__device__ __forceinline__ bool search_stopped(const uint32_t* ctl) { return (ctl[kDone] | ctl[kStuck]) != 0u; }

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

// One workgroup (4 waves) per bucket of the look-ahead window.  A ready bucket only re-checks its tentative pilot against
// the grown bitmap, and only inside the commit window (farther ahead it is left alone until it gets there); otherwise the scan
// continues from where it stopped, 256 pilots per step, at most max_steps steps per round (so one hard bucket never stalls a
// round: it started many rounds before it reaches the commit window).  Measured on MI355X: C2 search 1.12 s with one wave per
// bucket -> 0.78 s (profiles/r01_builder_stages.txt).
__global__ __launch_bounds__(256) void k_search(SearchArgs a) {
    __shared__ uint64_t sm_found[4];
    const uint32_t local = blockIdx.x, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (search_stopped(a.ctl) || local >= a.ctl[kNSearch]) return;
    const uint32_t n_commit = a.ctl[kNCommit];
    const uint32_t oi = a.ctl[kBase] + local, b = a.order[oi];
    const uint64_t* keys = a.hs + a.bstart[b];
    const uint32_t s = a.bstart[b + 1] - a.bstart[b];
    uint64_t start = a.pilot[oi];
    if (a.ready[oi]) {  // tentative pilot: still free?  (uniform over the workgroup)
        if (local >= n_commit) return;
        const uint64_t hp = murmur8(start, a.seedx);
        bool hit = false;
        for (uint32_t j = threadIdx.x; j < s; j += 256) hit |= is_taken(a.taken, position(keys[j], hp, a.div));
        if (!__syncthreads_or(hit)) return;
    }
    for (uint32_t step = 0; step < a.max_steps; ++step) {
        const uint64_t pilot = start + threadIdx.x;
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
        uint64_t found = ~0ull;
        while (cand) {  // lowest candidate pilot of this wave first; in-bucket duplicates are rare, so check them only here
            const int l = __ffsll((long long)cand) - 1;
            bool ok = false;
            if ((int)lane == l) ok = pilot_ok(a, keys, s, pilot);
            if (__any(ok)) {
                found = start + wave * kWave + (uint64_t)l;
                break;
            }
            cand &= cand - 1;
        }
        if (lane == 0) sm_found[wave] = found;
        __syncthreads();
        const uint64_t best = min(min(sm_found[0], sm_found[1]), min(sm_found[2], sm_found[3]));
        __syncthreads();
        if (best != ~0ull) {
            if (threadIdx.x == 0) {
                a.pilot[oi] = best;
                a.ready[oi] = 1;
            }
            return;
        }
        start += 256;
        if (start >= kPilotCap) {
            if (threadIdx.x == 0) a.ctl[kFailed] = 1;
            break;
        }
    }
    if (threadIdx.x == 0) {  // not found yet: resume here next round
        a.pilot[oi] = start;
        a.ready[oi] = 0;
        if (local < n_commit) atomicMin(&a.ctl[kNotReady], local);
    }
}

#define LPHQ_SEARCH_BUCKET_PROLOGUE()                                                              \
    const uint32_t local = (blockIdx.x * 256u + threadIdx.x) / kWave, lane = threadIdx.x & (kWave - 1); \
    if (search_over(a.ctl)) return;                                                                \
    const uint32_t ncommit = min(a.ctl[kNCommit], a.ctl[kNotReady]); /* stop at the first bucket still searching */ \
    if (local >= ncommit) return;                                                                  \
    const uint32_t oi = a.ctl[kBase] + local, b = a.order[oi];                                     \
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
    if (__any(lost) && lane == 0) atomicMin(&a.ctl[kIStar], local);
}

// buckets before i_star are final: take their positions; everybody releases the owner table
__global__ __launch_bounds__(256) void k_commit(SearchArgs a) {
    LPHQ_SEARCH_BUCKET_PROLOGUE();
    const bool final_ = local < a.ctl[kIStar];
    for (uint32_t j = lane; j < s; j += kWave) {
        const uint32_t p = position(keys[j], hp, a.div);
        a.owner[p] = kNoOwner;
        if (final_) atomicOr(&a.taken[p >> 5], 1u << (p & 31u));
    }
}

// End of a round (one thread): the conflict-free, fully searched prefix of the commit window is final; move the windows on.
// The commit window adapts to the conflict rate only (a bucket that is still searching stops the prefix too, but a smaller
// window would not help it): it grows while conflict-free and shrinks when a conflict cut it short.
__global__ void k_advance(SearchArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t* c = a.ctl;
    if (search_over(c)) return;
    const uint32_t n_commit = c[kNCommit], i_star = c[kIStar];
    const uint32_t in_window = min(n_commit, c[kNotReady]);
    const uint32_t committed = min(in_window, i_star);
    const uint32_t base = c[kBase] + committed;
    uint32_t window = c[kWindow];
    c[kRounds] += 1u;
    const uint32_t idle = committed ? 0u : c[kIdle] + 1u;
    c[kIdle] = idle;
    if (idle > c[kMaxIdle]) {
        c[kStuck] = 1u;
        return;
    }
    const bool conflict = i_star < in_window;
    if (!conflict && window < 8192u) window *= 2u;
    else if (conflict && committed < n_commit / 4u && window > 256u) window /= 2u;
    c[kWindow] = window;
    c[kBase] = base;
    const uint32_t n = c[kNNonEmpty];
    if (base >= n) {
        c[kDone] = 1u;
        return;
    }
    const uint32_t rem = n - base, ns = min(c[kLookAhead], rem);
    c[kNSearch] = ns;
    c[kNCommit] = min(min(window, rem), ns);  // only searched buckets may commit
    c[kIStar] = 0xFFFFFFFFu;
    c[kNotReady] = 0xFFFFFFFFu;
}

}  // namespace search
}  // namespace lphq
