// search_rounds.hpp -- host side of the rounds of the GPU pilot search, written against a small device backend so that the
// product (HIP: search.hip) and the CPU emulator of the tests (tests/emu/emu_search.cpp) run the very same control logic.
// A round is k_search, k_claim, k_validate, k_commit, k_advance; its windows live in device memory (search_device.cuh), so the
// host enqueues `rounds_per_sync` rounds back to back and looks at the control words once per batch instead of once per
// round (C2: ~9 K rounds; the per-round read-back was about half of the search's wall time).
//
// Backend Dev provides:
//   bool set_ctl(const uint32_t (&c)[kCtlWords]);  bool get_ctl(uint32_t (&c)[kCtlWords]);   // get_ctl synchronises
//   void launch_search(const search::SearchArgs&, int workgroups);  launch_claim / launch_validate / launch_commit likewise;
//   void launch_advance(const search::SearchArgs&);
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

#include "search_device.cuh"

namespace lphq {
namespace search {

inline lphq_divisor divisor_u32(uint64_t d) {  // same reciprocal as image.cpp::make_divisor, d < 2^32
    lphq_divisor r;
    memset(&r, 0, sizeof r);
    r.d64 = d;
    r.d = (uint32_t)d;
    r.shift = (uint32_t)__builtin_clz((uint32_t)d);
    r.dn = (uint32_t)d << r.shift;
    r.inv = (uint32_t)(~(uint64_t)0 / r.dn - (1ull << 32));
    return r;
}

// a: pointers + seedx + div filled in by the caller.  Returns false and sets err on failure; *rounds_out = rounds used.
template <class Dev>
bool run_rounds(Dev& dev, SearchArgs a, uint64_t n_nonempty, uint32_t look_ahead, uint32_t max_steps, uint32_t rounds_per_sync, std::string& err,
                uint64_t* rounds_out) {
    // tuning aids are clamped to sane ranges (an unclamped look-ahead below the commit window once committed unsearched buckets)
    a.max_steps = std::min<uint32_t>(1024u, std::max<uint32_t>(1u, max_steps));
    look_ahead = std::min<uint32_t>(1u << 20, std::max<uint32_t>(8192u, look_ahead));
    rounds_per_sync = std::min<uint32_t>(64u, std::max<uint32_t>(1u, rounds_per_sync));
    if (rounds_out) *rounds_out = 0;
    if (n_nonempty == 0) return true;
    if (n_nonempty > 0xFFFFFFFFull) {
        err = "gpu pilot search: too many buckets";
        return false;
    }
    uint32_t c[kCtlWords];
    memset(c, 0, sizeof c);
    c[kIStar] = c[kNotReady] = 0xFFFFFFFFu;
    c[kWindow] = 1024u;
    c[kNNonEmpty] = (uint32_t)n_nonempty;
    c[kLookAhead] = look_ahead;
    c[kMaxIdle] = (uint32_t)((kPilotCap / 256) / a.max_steps + 16);
    c[kNSearch] = (uint32_t)std::min<uint64_t>(look_ahead, n_nonempty);
    c[kNCommit] = std::min((uint32_t)std::min<uint64_t>(c[kWindow], n_nonempty), c[kNSearch]);
    if (!dev.set_ctl(c)) {
        err = "gpu pilot search: cannot initialise the round control words";
        return false;
    }
    // fixed grids (the kernels cut themselves down to the current windows): one workgroup per bucket of the look-ahead window,
    // one wave per bucket of the largest commit window
    const int search_grid = (int)std::min<uint64_t>(look_ahead, n_nonempty);
    const int commit_grid = (int)((std::min<uint64_t>(8192u, n_nonempty) * (uint64_t)kWave + 255) / 256);
    // Host-side bounds, independent of the control words the device writes: every round either commits >= 1 bucket or
    // counts towards kMaxIdle, so a correct search needs at most n_nonempty + kMaxIdle rounds (C2: ~9 K for 1.9 M buckets);
    // and a wall-clock limit (LPHQ_SEARCH_MAX_SECONDS, default 120 s; C4-scale searches take < 20 s).  On either the caller
    // (builder.cpp) finishes with the sequential CPU search.
    const uint64_t round_cap = n_nonempty + (uint64_t)c[kMaxIdle] + 64u;
    double max_seconds = 120.0;
    if (const char* e = getenv("LPHQ_SEARCH_MAX_SECONDS")) max_seconds = std::min(3600.0, std::max(1e-3, atof(e)));  // (milliseconds: the tests of the fall-back to the CPU search)
    const auto t_start = std::chrono::steady_clock::now();
    uint64_t enqueued = 0;
    for (;;) {
        for (uint32_t r = 0; r < rounds_per_sync; ++r) {
            dev.launch_search(a, search_grid);
            dev.launch_claim(a, commit_grid);
            dev.launch_validate(a, commit_grid);
            dev.launch_commit(a, commit_grid);
            dev.launch_advance(a);
        }
        if (!dev.get_ctl(c)) {
            err = "gpu pilot search: cannot read the round control words";
            return false;
        }
        if (rounds_out) *rounds_out = c[kRounds];
        if (c[kFailed]) {
            err = "pthash search: no pilot below 2^28 for some bucket (duplicate key in the input, or a degenerate tiny key set)";
            return false;
        }
        if (c[kStuck]) {
            err = "gpu pilot search: no progress";
            return false;
        }
        if (c[kDone]) return true;
        enqueued += rounds_per_sync;
        if (enqueued > round_cap) {
            err = "gpu pilot search: round cap reached without completion";
            return false;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > max_seconds) {
            err = "gpu pilot search: wall-clock limit reached";
            return false;
        }
    }
}

}  // namespace search
}  // namespace lphq
