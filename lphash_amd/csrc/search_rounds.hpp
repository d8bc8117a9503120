// search_rounds.hpp -- the round loop of the GPU pilot search, written against a small device backend so that the product
// (HIP: search.hip) and the CPU emulator of the tests (tests/emu/emu_search.cpp) run the very same control logic.
//
// Backend Dev provides:
//   bool set_flags(const uint32_t (&f)[3]);  bool get_flags(uint32_t (&f)[3]);          // [0] i_star, [1] failed, [2] first not-ready
//   void launch_search(const search::SearchArgs&, int workgroups);  launch_claim / launch_validate / launch_commit  likewise
#pragma once
#include <algorithm>
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
bool run_rounds(Dev& dev, SearchArgs a, uint64_t n_nonempty, uint32_t look_ahead, uint32_t max_steps, std::string& err, uint64_t* rounds_out) {
    // tuning aids are clamped to sane ranges (an unclamped look-ahead below the commit window once committed unsearched buckets)
    a.max_steps = std::min<uint32_t>(1024u, std::max<uint32_t>(1u, max_steps));
    look_ahead = std::min<uint32_t>(1u << 20, std::max<uint32_t>(8192u, look_ahead));
    uint64_t base = 0, rounds = 0, idle_rounds = 0;
    uint32_t window = 1024;
    while (base < n_nonempty) {
        a.base = (uint32_t)base;
        a.n_search = (uint32_t)std::min<uint64_t>(look_ahead, n_nonempty - base);
        a.n_commit = std::min((uint32_t)std::min<uint64_t>(window, n_nonempty - base), a.n_search);  // only searched buckets may commit
        const uint32_t flags_init[3] = {0xFFFFFFFFu, 0, 0xFFFFFFFFu};
        if (!dev.set_flags(flags_init)) {
            err = "gpu pilot search: cannot reset the round flags";
            return false;
        }
        dev.launch_search(a, (int)a.n_search);  // one workgroup per bucket
        const int grid = (int)((a.n_commit * (uint64_t)kWave + 255) / 256);
        dev.launch_claim(a, grid);
        dev.launch_validate(a, grid);
        dev.launch_commit(a, grid);
        uint32_t flags[3];
        if (!dev.get_flags(flags)) {
            err = "gpu pilot search: cannot read the round flags";
            return false;
        }
        if (flags[1]) {
            err = "pthash search: no pilot below 2^28 for some bucket (duplicate key in the input, or a degenerate tiny key set)";
            return false;
        }
        const uint32_t in_window = std::min(a.n_commit, flags[2]);
        const uint32_t committed = std::min(in_window, flags[0]);
        base += committed;
        ++rounds;
        idle_rounds = committed ? 0 : idle_rounds + 1;
        if (idle_rounds > (kPilotCap / 256) / a.max_steps + 16) {
            err = "gpu pilot search: no progress";
            return false;
        }
        // adapt the commit window to the conflict rate only (a bucket that is still searching stops the prefix too, but a
        // smaller window would not help it): grow while conflict-free, shrink when a conflict cut it short
        const bool conflict = flags[0] < in_window;
        if (!conflict && window < 8192) window *= 2;
        else if (conflict && committed < a.n_commit / 4 && window > 256) window /= 2;
    }
    if (rounds_out) *rounds_out = rounds;
    return true;
}

}  // namespace search
}  // namespace lphq
