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
    kIStar = 0,      // (unused since round 5: the window is no longer cut at its first conflict)
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
    // repair passes (round 5): inside a round the commit window is driven to the sequential answer by fixed-point iteration instead of being cut at its first conflict
    kChanged,        // first local index whose pilot changed in the LAST executed pass (none: 0xFFFFFFFF)
    kChangedNow,     // ... in the pass that is running
    kConverged,      // a pass changed nothing: the remaining passes of the round do nothing
    kCut,            // first local index whose repair ran out of its step budget: the window ends there this round
    kCutNext,        // ... as found by the running pass (takes effect at the end of the pass: the window does not change under a pass's claims)
    kListCount,      // buckets queued for re-evaluation in the running pass
    kPass,           // passes executed this round
    kCtlWords = 24
};
constexpr uint32_t kRepairListCap = 8192u;  // = the largest commit window

struct SearchArgs {
    const uint64_t* hs;       // hashes grouped by bucket
    const uint32_t* bstart;   // bucket -> first hash
    const uint32_t* order;    // processing order -> bucket id
    uint32_t* taken;          // committed bitmap, table_size bits
    uint32_t* owner;          // table_size entries, kNoOwner when unclaimed
    uint64_t* pilot;          // per order index: tentative pilot (ready) or the next pilot to try (not ready)
    uint8_t* ready;           // per order index: 1 = pilot was valid against the bitmap when last checked
    uint64_t* cur;            // per order index, commit window only: the pilot of the running fixed-point iteration (starts at `pilot`, the first one valid against the bitmap)
    uint32_t* list;           // kRepairListCap local indices: the buckets a pass re-evaluates
    uint64_t* fixed;          // kRepairListCap pilots: what the re-evaluation found for list entry e
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

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Repair passes (round 5).  The sequential search gives bucket b the FIRST pilot whose positions avoid the table AND the positions of all earlier
// buckets.  Until round 5 a round committed the prefix of its window up to the first bucket that lost a position to an earlier one (~430 buckets per
// round at C4: 23 K rounds).  Now the window is driven to that answer by fixed-point iteration: every bucket starts at its base pilot (the first one valid
// against the committed table, k_search); a pass lets all buckets claim their positions (atomicMin by order), and every bucket that lost a position to an
// EARLIER one -- or that had moved off its base pilot while an earlier bucket changed in the pass before -- is re-evaluated FROM ITS BASE PILOT against
// (table + positions claimed by earlier buckets).  At a fixed point every bucket holds the first pilot that avoids the table and the positions of the earlier
// buckets as they now are: the sequential answer, by induction over the order (bucket 0 of the window never moves; once all buckets before b are final, one
// more pass makes b final).  A round runs a fixed number of passes (no host round trip); if the last executed pass still changed something, the prefix before
// its first changed bucket is at the fixed point (nothing before it changed in that pass, and what those buckets were evaluated against did not change
// either) and is what gets committed.  A bucket whose re-evaluation runs out of its step budget cuts the window at its index for this round.
// ---------------------------------------------------------------------------------------------------------------------------------------------
#define LPHQ_REPAIR_BUCKET_PROLOGUE()                                                              \
    const uint32_t local = (blockIdx.x * 256u + threadIdx.x) / kWave, lane = threadIdx.x & (kWave - 1); \
    if (search_over(a.ctl) || a.ctl[kConverged]) return;                                           \
    const uint32_t ncommit = min(min(a.ctl[kNCommit], a.ctl[kNotReady]), a.ctl[kCut]);             \
    if (local >= ncommit) return;                                                                  \
    const uint32_t oi = a.ctl[kBase] + local, b = a.order[oi];                                     \
    const uint64_t* keys = a.hs + a.bstart[b];                                                     \
    const uint32_t s = a.bstart[b + 1] - a.bstart[b]

// start of a round's iteration: every bucket of the commit window at its base pilot
__global__ __launch_bounds__(256) void k_repair_begin(SearchArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (search_over(a.ctl)) return;
    if (i == 0) {
        a.ctl[kChanged] = 0u;  // "everything may have changed": pass 0 evaluates every bucket that lost a position
        a.ctl[kChangedNow] = 0xFFFFFFFFu;
        a.ctl[kConverged] = 0u;
        a.ctl[kCut] = 0xFFFFFFFFu;
        a.ctl[kCutNext] = 0xFFFFFFFFu;
        a.ctl[kListCount] = 0u;
        a.ctl[kPass] = 0u;
    }
    const uint32_t ncommit = min(a.ctl[kNCommit], a.ctl[kNotReady]);
    if (i < ncommit) a.cur[a.ctl[kBase] + i] = a.pilot[a.ctl[kBase] + i];
}

__global__ __launch_bounds__(256) void k_repair_claim(SearchArgs a) {
    LPHQ_REPAIR_BUCKET_PROLOGUE();
    const uint64_t hp = murmur8(a.cur[oi], a.seedx);
    for (uint32_t j = lane; j < s; j += kWave) atomicMin(&a.owner[position(keys[j], hp, a.div)], local);
}

// which buckets must be re-evaluated in this pass
__global__ __launch_bounds__(256) void k_repair_list(SearchArgs a) {
    LPHQ_REPAIR_BUCKET_PROLOGUE();
    const uint32_t prev_changed = a.ctl[kChanged];  // first bucket that changed in the pass before (pass 0: 0)
    if (prev_changed >= local) return;               // nothing this bucket was evaluated against has changed: it stands (the first bucket of the window always does)
    const uint64_t cur = a.cur[oi];
    const uint64_t hp = murmur8(cur, a.seedx);
    bool lost = false;
    for (uint32_t j = lane; j < s; j += kWave) lost |= a.owner[position(keys[j], hp, a.div)] < local;
    const bool need = __any(lost) || cur != a.pilot[oi];
    if (need && lane == 0) {
        const uint32_t e = atomicAdd(&a.ctl[kListCount], 1u);
        a.list[e] = local;  // (e < window <= kRepairListCap)
    }
}

// exact check of one pilot by one lane against the table AND the claims of earlier buckets
__device__ bool pilot_ok_claims(const SearchArgs& a, const uint64_t* keys, uint32_t s, uint64_t pilot, uint32_t local) {
    const uint64_t hp = murmur8(pilot, a.seedx);
    for (uint32_t j = 0; j < s; ++j) {
        const uint32_t pj = position(keys[j], hp, a.div);
        if (is_taken(a.taken, pj) || a.owner[pj] < local) return false;
        for (uint32_t t = 0; t < j; ++t)
            if (position(keys[t], hp, a.div) == pj) return false;
    }
    return true;
}

// One workgroup per queued bucket (the grid walks over the list): the first pilot >= the base pilot whose positions are free in the table, not claimed by an
// earlier bucket, and pairwise distinct; max_steps steps of 256 pilots, else the window is cut at this bucket for the round.
__global__ __launch_bounds__(256) void k_repair_fix(SearchArgs a) {
    __shared__ uint64_t sm_found[4];
    if (search_over(a.ctl) || a.ctl[kConverged]) return;
    const uint32_t n = a.ctl[kListCount];
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
        const uint32_t local = a.list[e];
        const uint32_t oi = a.ctl[kBase] + local, b = a.order[oi];
        const uint64_t* keys = a.hs + a.bstart[b];
        const uint32_t s = a.bstart[b + 1] - a.bstart[b];
        uint64_t start = a.pilot[oi];
        uint64_t result = ~0ull;
        for (uint32_t step = 0; step < a.max_steps; ++step) {
            const uint64_t pilot = start + threadIdx.x;
            const uint64_t hp = murmur8(pilot, a.seedx);
            bool alive = true;
            for (uint32_t j = 0; j < s; j += 4) {
                uint32_t hit = 0;
#pragma unroll
                for (uint32_t t = 0; t < 4; ++t)
                    if (j + t < s) {
                        const uint32_t pj = position(keys[j + t], hp, a.div);
                        hit |= (uint32_t)(is_taken(a.taken, pj) || a.owner[pj] < local);
                    }
                alive = alive && !hit;
                if (!__any(alive)) break;
            }
            uint64_t cand = __ballot(alive);
            uint64_t found = ~0ull;
            while (cand) {
                const int l = __ffsll((long long)cand) - 1;
                bool ok = false;
                if ((int)lane == l) ok = pilot_ok_claims(a, keys, s, pilot, local);
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
                result = best;
                break;
            }
            start += 256;
            if (start >= kPilotCap) break;  // (the bucket cuts the window; if it is unplaceable k_search reports it when it reaches the front)
        }
        if (threadIdx.x == 0) {
            a.fixed[e] = result;
            if (result == ~0ull) atomicMin(&a.ctl[kCutNext], local);
        }
    }
}

// every bucket lets go of its claims (the positions of the pilots the pass started with)
__global__ __launch_bounds__(256) void k_repair_release(SearchArgs a) {
    LPHQ_REPAIR_BUCKET_PROLOGUE();
    const uint64_t hp = murmur8(a.cur[oi], a.seedx);
    for (uint32_t j = lane; j < s; j += kWave) a.owner[position(keys[j], hp, a.div)] = kNoOwner;
}

// end of a pass (one workgroup): the re-evaluated buckets take their new pilots; did anything change?
__global__ __launch_bounds__(256) void k_repair_apply(SearchArgs a) {
    if (blockIdx.x != 0 || search_over(a.ctl) || a.ctl[kConverged]) return;
    const uint32_t n = a.ctl[kListCount];
    for (uint32_t e = threadIdx.x; e < n; e += 256u) {
        const uint32_t local = a.list[e];
        const uint32_t oi = a.ctl[kBase] + local;
        const uint64_t f = a.fixed[e];
        if (f != ~0ull && f != a.cur[oi]) {
            a.cur[oi] = f;
            atomicMin(&a.ctl[kChangedNow], local);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t ch = a.ctl[kChangedNow];
        a.ctl[kChanged] = ch;
        a.ctl[kChangedNow] = 0xFFFFFFFFu;
        a.ctl[kListCount] = 0u;
        a.ctl[kPass] += 1u;
        const uint32_t cut_before = a.ctl[kCut], cut = min(cut_before, a.ctl[kCutNext]);
        a.ctl[kCut] = cut;
        // settled: nothing changed AND the window did not shrink in this pass (a bucket that ran out of budget leaves the window: the buckets before it stand
        // as they are -- they never depended on it)
        if (ch == 0xFFFFFFFFu) a.ctl[kConverged] = 1u;
    }
}

// the prefix that is at the fixed point becomes final: its positions are taken, its pilots are the answer
__global__ __launch_bounds__(256) void k_repair_commit(SearchArgs a) {
    const uint32_t local = (blockIdx.x * 256u + threadIdx.x) / kWave, lane = threadIdx.x & (kWave - 1);
    if (search_over(a.ctl)) return;
    const uint32_t limit = min(min(min(a.ctl[kNCommit], a.ctl[kNotReady]), a.ctl[kCut]), a.ctl[kChanged]);
    if (local >= limit) return;
    const uint32_t oi = a.ctl[kBase] + local, b = a.order[oi];
    const uint64_t* keys = a.hs + a.bstart[b];
    const uint32_t s = a.bstart[b + 1] - a.bstart[b];
    const uint64_t cur = a.cur[oi];
    const uint64_t hp = murmur8(cur, a.seedx);
    for (uint32_t j = lane; j < s; j += kWave) {
        const uint32_t p = position(keys[j], hp, a.div);
        atomicOr(&a.taken[p >> 5], 1u << (p & 31u));
    }
    if (lane == 0) a.pilot[oi] = cur;
}

// End of a round (one thread): the conflict-free, fully searched prefix of the commit window is final; move the windows on.
// The commit window adapts to the conflict rate only (a bucket that is still searching stops the prefix too, but a smaller
// window would not help it): it grows while conflict-free and shrinks when a conflict cut it short.
__global__ void k_advance(SearchArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t* c = a.ctl;
    if (search_over(c)) return;
    const uint32_t n_commit = c[kNCommit], i_star = min(c[kCut], c[kChanged]);  // (round 5: what cuts the committed prefix short is a repair out of budget, or passes that did not settle)
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
    c[kNotReady] = 0xFFFFFFFFu;
}

}  // namespace search
}  // namespace lphq
