// stream_sinks.h -- what the streaming kernel does with the codes of a tile (plain structs: included by the device sources and by the host launchers).
#pragma once
#include <stdint.h>

#include "../../include/lphq.h"  // lphq_run, lphq_run_counts, LPHQ_RUN_ARENAS: the run-encoded result is part of the C ABI

namespace lphq {

// What phase F does with the codes of a tile (template parameter SINK of the kernel; the lookups, phases A-E, are the same for all three):
//   kSinkCodes   8 B per window to `out` (lphq_query_records)
//   kSinkRuns    one 16-B lphq_run per run of affine codes, appended to the tile's arena (lphq_query_records_runs): what the reference's streaming
//                branch computes before it expands it (include/partitioned_mphf.hpp:131-145)
//   kSinkGather  table[code] per window (lphq_query_records_gather): the codes never reach memory
constexpr int kSinkCodes = 0, kSinkRuns = 1, kSinkGather = 2;
struct S8NoArgs {};
struct S8RunsArgs {
    lphq_run* runs;           // LPHQ_RUN_ARENAS arenas of arena_runs entries
    uint64_t* codes;          // ... of arena_codes explicit codes (windows under colliding minimizers)
    lphq_run_counts* counts;  // per arena: entries appended so far (may end above the capacity: the excess was not written)
    uint64_t arena_runs, arena_codes;
};
struct S8GatherArgs {
    const uint8_t* table;  // n_slots elements of 1 << elem_log2 bytes, in the order of the codes
    uint8_t* out;          // one element per window, at the window's index in the code space
    uint64_t n_slots, missing;
    uint32_t elem_log2;
};
}  // namespace lphq
