// TEST INFRASTRUCTURE ONLY.  The product's GPU pilot search -- kernels (lphash_amd/csrc/search_device.cuh) AND round loop
// (search_rounds.hpp) -- run on the fiber emulator, next to a plain sequential first-fit reference (PTHash v1 search).
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "search_rounds.hpp"

using namespace lphq;
using namespace lphq::search;

namespace {
struct EmuDev {  // "device" memory is host memory, a launch is an emulated grid
    uint32_t* ctl;
    bool set_ctl(const uint32_t (&c)[kCtlWords]) { memcpy(ctl, c, sizeof c); return true; }
    bool get_ctl(uint32_t (&c)[kCtlWords]) { memcpy(c, ctl, sizeof c); return true; }
    // The product launches fixed grids and lets the kernels cut themselves down to the current windows.  An emulated workgroup
    // costs 256 fiber start-ups even when it exits at once, so the emulator launches the window's workgroups plus ONE beyond it
    // (the early-exit path still runs) -- it can, because its "device" control words are host memory.
    unsigned cut(int g, unsigned need) const {
        const bool over = (ctl[kDone] | ctl[kFailed] | ctl[kStuck]) != 0;
        return (unsigned)std::min<uint64_t>((uint64_t)g, over ? 1u : (uint64_t)need + 1u);
    }
    unsigned commit_wgs() const { return (unsigned)(((uint64_t)ctl[kNCommit] * kWave + 255) / 256); }
    void launch_search(const SearchArgs& x, int g) { emu::launch(cut(g, ctl[kNSearch]), 256, [&] { k_search(x); }); }
    void launch_claim(const SearchArgs& x, int g) { emu::launch(cut(g, commit_wgs()), 256, [&] { k_claim(x); }); }
    void launch_validate(const SearchArgs& x, int g) { emu::launch(cut(g, commit_wgs()), 256, [&] { k_validate(x); }); }
    void launch_commit(const SearchArgs& x, int g) { emu::launch(cut(g, commit_wgs()), 256, [&] { k_commit(x); }); }
    void launch_advance(const SearchArgs& x) { emu::launch(1, 64, [&] { k_advance(x); }); }
};
uint64_t murmur_u64(uint64_t v, uint64_t seed) { return murmur8(v, seed ^ (8ull * LPHQ_MURMUR_M)); }
}  // namespace

// returns 0 on success; pilots by processing order; taken = table_size-bit bitmap in u32 words
extern "C" int emu_pilot_search(const uint64_t* hs, const uint32_t* bstart, const uint32_t* order, uint64_t n_nonempty, uint64_t table_size, uint64_t seed,
                                uint32_t look_ahead, uint32_t steps, uint32_t rounds_per_sync, uint64_t* pilot_by_order, uint32_t* taken, uint64_t* rounds) {
    std::vector<uint32_t> owner(table_size, kNoOwner);
    std::vector<uint8_t> ready(n_nonempty ? n_nonempty : 1, 0);
    uint32_t ctl[kCtlWords];
    memset(ctl, 0, sizeof ctl);
    memset(taken, 0, ((table_size + 63) / 64 * 2 + 2) * 4);
    memset(pilot_by_order, 0, n_nonempty * 8);
    SearchArgs a;
    memset(&a, 0, sizeof a);
    a.hs = hs;
    a.bstart = bstart;
    a.order = order;
    a.taken = taken;
    a.owner = owner.data();
    a.pilot = pilot_by_order;
    a.ready = ready.data();
    a.ctl = ctl;
    a.seedx = seed ^ (8ull * LPHQ_MURMUR_M);
    a.div = divisor_u32(table_size);
    EmuDev dev{ctl};
    std::string err;
    if (!run_rounds(dev, a, n_nonempty, look_ahead, steps, rounds_per_sync, err, rounds)) return 1;
    for (uint32_t o : owner)
        if (o != kNoOwner) return 2;  // every round must release the owner table
    return 0;
}

// PTHash v1 search, sequential: first pilot whose positions are free and pairwise distinct, buckets in the given order
extern "C" int ref_pilot_search(const uint64_t* hs, const uint32_t* bstart, const uint32_t* order, uint64_t n_nonempty, uint64_t table_size, uint64_t seed,
                                uint64_t* pilot_by_order, uint32_t* taken) {
    memset(taken, 0, ((table_size + 63) / 64 * 2 + 2) * 4);
    std::vector<uint64_t> pos;
    for (uint64_t oi = 0; oi < n_nonempty; ++oi) {
        const uint32_t b = order[oi];
        const uint64_t* keys = hs + bstart[b];
        const uint32_t s = bstart[b + 1] - bstart[b];
        pos.resize(s);
        for (uint64_t pilot = 0;; ++pilot) {
            if (pilot >> 28) return 1;
            const uint64_t hp = murmur_u64(pilot, seed);
            bool ok = true;
            for (uint32_t j = 0; j < s && ok; ++j) {
                pos[j] = (keys[j] ^ hp) % table_size;
                ok = !((taken[pos[j] >> 5] >> (pos[j] & 31)) & 1);
                for (uint32_t t = 0; t < j && ok; ++t) ok = pos[t] != pos[j];
            }
            if (!ok) continue;
            for (uint32_t j = 0; j < s; ++j) taken[pos[j] >> 5] |= 1u << (pos[j] & 31);
            pilot_by_order[oi] = pilot;
            break;
        }
    }
    return 0;
}
