// TEST INFRASTRUCTURE ONLY.  The run-encoded result and the fused payload pass (stream8_device.cuh: kSinkRuns, kSinkGather; k_expand_runs, k_compact_runs)
// compiled for the host and run by the fiber emulator -- a translation unit of its own so that the emulator library builds in parallel.
#include <cstring>
#include <vector>

#include "kernels_device.cuh"  // -Itests/emu resolves <hip/hip_runtime.h> to the stub
#include "stream8_device.cuh"

using namespace lphq;

static bool header(const uint8_t* img, size_t nbytes, lphq_image_header& H) {
    if (nbytes < sizeof H) return false;
    memcpy(&H, img, sizeof H);
    return H.magic == LPHQ_IMAGE_MAGIC && H.total_bytes == nbytes;
}

// The streaming pass into the other two sinks (stream8_device.cuh: kSinkRuns, kSinkGather) as kernels.hip launches it: the eight-per-lane kernel for every W.
template <int SINK, typename Args>
static int emu_stream_sink(const uint8_t* img, size_t nbytes, const char* seq, uint64_t total, const uint64_t* seq_off, const uint64_t* out_off, uint32_t n_records,
                           const Args& sink) {
    lphq_image_header H;
    if (!header(img, nbytes, H) || H.kind >= 2) return 1;
    if (n_records == 0 || total < H.k) return 0;
    const uint64_t shift = (uint64_t)((uintptr_t)seq & 15u);
    const uint8_t* seq_al = (const uint8_t*)seq - shift;
    const uint64_t tw = stream_tile_windows(H.W, 8 * kS8Block);
    const unsigned tiles = (unsigned)((shift + total + tw - 1) / tw);
#define EMU_SINK_W8(W8V) emu::launch(tiles, kS8Block, [&] { k_query_tiles_stream8<KIND_, K128_, 6, kS8Block, false, W8V, SINK>(img, seq_al, shift, total, seq_off, out_off, n_records, nullptr, 0, sink); });
#define CALL(KIND, K128, P)                                                                                                                                       \
    do {                                                                                                                                                          \
        constexpr int KIND_ = KIND;                                                                                                                               \
        constexpr bool K128_ = K128;                                                                                                                              \
        if (P == 6) {                                                                                                                                             \
            LPHQ_S8_SWITCH_W8(H.W, EMU_SINK_W8)                                                                                                                   \
        } else emu::launch(tiles, kS8Block, [&] { k_query_tiles_stream8<KIND, K128, 4, kS8Block, true, -1, SINK>(img, seq_al, shift, total, seq_off, out_off, n_records, nullptr, 0, sink); }); \
    } while (0)
    const bool k128 = H.kmer_bits == 128, p6 = H.P == 6;
    if (H.kind == 0) {
        if (k128) { if (p6) { CALL(0, true, 6); } else { CALL(0, true, 4); } }
        else      { if (p6) { CALL(0, false, 6); } else { CALL(0, false, 4); } }
    } else {
        if (k128) { if (p6) { CALL(1, true, 6); } else { CALL(1, true, 4); } }
        else      { if (p6) { CALL(1, false, 6); } else { CALL(1, false, 4); } }
    }
#undef CALL
#undef EMU_SINK_W8
    return 0;
}

extern "C" int emu_kernel_query_records_runs(const uint8_t* img, size_t nbytes, const char* seq, uint64_t total, const uint64_t* seq_off, const uint64_t* out_off,
                                             uint32_t n_records, lphq_run* runs, uint64_t cap_runs, uint64_t* codes, uint64_t cap_codes, lphq_run_counts* counts) {
    memset(counts, 0, sizeof(lphq_run_counts) * LPHQ_RUN_ARENAS);
    const S8RunsArgs sink{runs, codes, counts, cap_runs / LPHQ_RUN_ARENAS, cap_codes / LPHQ_RUN_ARENAS};
    return emu_stream_sink<kSinkRuns>(img, nbytes, seq, total, seq_off, out_off, n_records, sink);
}

extern "C" int emu_kernel_query_records_gather(const uint8_t* img, size_t nbytes, const char* seq, uint64_t total, const uint64_t* seq_off, const uint64_t* out_off,
                                               uint32_t n_records, const void* table, uint64_t n_slots, uint32_t elem_log2, uint64_t missing, void* pout) {
    const S8GatherArgs sink{(const uint8_t*)table, (uint8_t*)pout, n_slots, missing, elem_log2};
    return emu_stream_sink<kSinkGather>(img, nbytes, seq, total, seq_off, out_off, n_records, sink);
}

extern "C" int emu_kernel_expand_runs(const lphq_run* runs, uint64_t cap_runs, const uint64_t* codes, uint64_t cap_codes, const lphq_run_counts* counts, uint64_t* out,
                                      uint64_t n_out, unsigned long long* status) {
    memset(out, 0xFF, n_out * sizeof(uint64_t));
    *status = 0;
    for (unsigned a = 0; a < LPHQ_RUN_ARENAS; ++a) {  // (the emulator's grid is one-dimensional: blockIdx.y is set by hand)
        emu::g_block_idx.y = a;
        emu::launch(3, kBlock, [&] { k_expand_runs<>(runs, cap_runs / LPHQ_RUN_ARENAS, codes, cap_codes / LPHQ_RUN_ARENAS, counts, out, n_out, status); });
    }
    emu::g_block_idx.y = 0;
    return 0;
}

extern "C" int emu_kernel_compact_runs(const lphq_run* runs, uint64_t cap_runs, const uint64_t* codes, uint64_t cap_codes, const lphq_run_counts* counts, lphq_run* dense_runs,
                                       uint64_t cap_dense_runs, uint64_t* dense_codes, uint64_t cap_dense_codes, unsigned long long* totals /* 4 */) {
    memset(totals, 0, 4 * sizeof *totals);
    for (unsigned a = 0; a < LPHQ_RUN_ARENAS; ++a) {
        emu::g_block_idx.y = a;
        emu::launch(2, kBlock, [&] {
            k_compact_runs<>(runs, cap_runs / LPHQ_RUN_ARENAS, codes, cap_codes / LPHQ_RUN_ARENAS, counts, dense_runs, cap_dense_runs, dense_codes, cap_dense_codes, totals);
        });
    }
    emu::g_block_idx.y = 0;
    return 0;
}

