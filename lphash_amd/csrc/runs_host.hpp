// runs_host.hpp -- the host side of the run-encoded streaming result (include/lphq.h: lphq_run): how a session sizes a batch's arenas, and the expansion of
// a dense array of runs into the reference's codes on the host's threads.  No HIP in here: tests/san/san_host.cpp runs it under ASan / UBSan / TSan.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>

#include "../../include/lphq.h"
#include "image_layout.h"
#include "work_pool.hpp"

namespace lphq {

// How a streaming batch of `windows` windows over `bases` bases in n_records records is sized for the run-encoded transport.  Expected runs on random
// sequence: 2 / (W + 1) per window, + the cuts at tile and wave ends (two per tile) and at record starts; the arenas take 1.4 x that (a tile's runs all land
// in one arena, so small batches are uneven), the unasked copy back 1.15 x.  A batch that does not fit (low-complexity sequence: up to one run per window)
// is answered again with worst-case arenas (session_wait) -- slower, never wrong.
struct RunPlan {
    size_t per_arena, xper_arena, copy_runs, copy_xc, tiles;
};
inline RunPlan plan_runs(const lphq_image_header& H, uint64_t windows, uint64_t bases, size_t n_records, size_t xc_hint, bool worst_case) {
    RunPlan p;
    const uint64_t tw = (uint64_t)((1024 - (int)H.W + 1) & ~15);  // stream_tile_windows(W, 1024)
    p.tiles = (size_t)((bases + 16 + tw - 1) / tw);
    const uint64_t tiles_per_arena = (p.tiles + LPHQ_RUN_ARENAS - 1) / LPHQ_RUN_ARENAS;
    if (worst_case) {  // every window a run of its own, every window under a colliding minimizer
        p.per_arena = p.xper_arena = (size_t)std::min<uint64_t>(windows, 1024 * tiles_per_arena) + 8;
        p.copy_runs = p.copy_xc = p.per_arena * LPHQ_RUN_ARENAS;
        return p;
    }
    const uint64_t expected = 2 * windows / (H.W + 1) + 2 * p.tiles + n_records + 64;
    p.per_arena = (size_t)(expected * 14 / 10 / LPHQ_RUN_ARENAS + 1024);
    p.xper_arena = (size_t)(windows / (4 * LPHQ_RUN_ARENAS) + 1024);
    p.copy_runs = (size_t)std::min<uint64_t>(p.per_arena * LPHQ_RUN_ARENAS, expected * 115 / 100 + 1024);
    p.copy_xc = (size_t)std::min<uint64_t>(p.xper_arena * LPHQ_RUN_ARENAS, (uint64_t)xc_hint * 3 / 2 + windows / 1024 + 4096);
    return p;
}
// out[first + i] for every run, on the pool's threads (pieces of 2048 runs: ~13 K windows, ~100 KB of stores each at k = 31 / m = 20)
inline void expand_runs_host(const lphq_run* runs, size_t n, const uint64_t* xc, uint64_t* out, WorkPool* pool) {
    constexpr size_t kPiece = 2048;
    auto piece = [&](size_t pi) {
        const size_t hi = std::min(n, (pi + 1) * kPiece);
        for (size_t r = pi * kPiece; r < hi; ++r) {
            const lphq_run run = runs[r];
            uint64_t* dst = out + LPHQ_RUN_FIRST(run);
            const uint32_t len = LPHQ_RUN_LENGTH(run), kind = LPHQ_RUN_KIND(run);
            if (kind == LPHQ_RUN_UP) {
                for (uint32_t i = 0; i < len; ++i) dst[i] = run.code + i;
            } else if (kind == LPHQ_RUN_DOWN) {
                for (uint32_t i = 0; i < len; ++i) dst[i] = run.code - i;
            } else {
                memcpy(dst, xc + run.code, (size_t)len * sizeof(uint64_t));
            }
        }
    };
    const size_t pieces = (n + kPiece - 1) / kPiece;
    if (pool) pool->run(pieces, piece);
    else
        for (size_t pi = 0; pi < pieces; ++pi) piece(pi);
}

}  // namespace lphq
