// stream8.hip -- launcher of the eight-positions-per-lane streaming kernel (stream8_device.cuh).  A translation unit of its own, compiled once per function
// kind (-DLPHQ_S8_KIND=0 / 1: partitioned / unpartitioned) and per sink (-DLPHQ_S8_SINK=0 / 1 / 2: codes / runs / payload, stream8_device.cuh), because the
// wide-window variant exists once per value of (W - 8) % 8: 2 x 3 x 2 x (8 + 1) kernels that make builds in parallel instead of one long compile of kernels.hip.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "stream8_device.cuh"

namespace lphq {

template <int KIND, int SINK>
static hipError_t launch_s8(const lphq_image_header& H, const uint8_t* d_img, const uint8_t* seq_al, uint64_t shift, uint64_t total_bases, const uint64_t* d_seq_off,
                            const uint64_t* d_out_off, uint32_t n_records, uint64_t* d_out, uint64_t tile_base, int grid, hipStream_t stream,
                            const typename S8SinkArgs<SINK>::type& sink) {
#define LPHQ_S8_ARGS <<<grid, kS8Block, 0, stream>>>(d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, n_records, d_out, tile_base, sink)
    if (H.P == 6) {  // W >= 16: block decomposition, one instantiation per (W - 8) % 8
        if (H.kmer_bits == 128) {
#define LPHQ_S8_CALL(W8) k_query_tiles_stream8<KIND, true, 6, kS8Block, false, W8, SINK> LPHQ_S8_ARGS
            LPHQ_S8_SWITCH_W8(H.W, LPHQ_S8_CALL)
#undef LPHQ_S8_CALL
        } else {
#define LPHQ_S8_CALL(W8) k_query_tiles_stream8<KIND, false, 6, kS8Block, false, W8, SINK> LPHQ_S8_ARGS
            LPHQ_S8_SWITCH_W8(H.W, LPHQ_S8_CALL)
#undef LPHQ_S8_CALL
        }
    } else if (H.kmer_bits == 128) {  // W <= 15: window minima from the raw hashes (any W >= 1; the codes sink is launched from W = 8 on, kernels.hip)
        k_query_tiles_stream8<KIND, true, 4, kS8Block, true, -1, SINK> LPHQ_S8_ARGS;
    } else {
        k_query_tiles_stream8<KIND, false, 4, kS8Block, true, -1, SINK> LPHQ_S8_ARGS;
    }
#undef LPHQ_S8_ARGS
    return hipGetLastError();
}

#define LPHQ_S8_PARAMS                                                                                                                                           \
    const lphq_image_header &H, const uint8_t *d_img, const uint8_t *seq_al, uint64_t shift, uint64_t total_bases, const uint64_t *d_seq_off, const uint64_t *d_out_off, \
        uint32_t n_records
#define LPHQ_S8_TAIL uint64_t tile_base, int grid, hipStream_t stream
#define LPHQ_S8_FWD H, d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, n_records

#if LPHQ_S8_SINK == 0
#if LPHQ_S8_KIND == 0
hipError_t launch_query_tiles_stream8_p(LPHQ_S8_PARAMS, uint64_t* d_out, LPHQ_S8_TAIL) { return launch_s8<0, kSinkCodes>(LPHQ_S8_FWD, d_out, tile_base, grid, stream, S8NoArgs{}); }
#else
hipError_t launch_query_tiles_stream8_u(LPHQ_S8_PARAMS, uint64_t* d_out, LPHQ_S8_TAIL) { return launch_s8<1, kSinkCodes>(LPHQ_S8_FWD, d_out, tile_base, grid, stream, S8NoArgs{}); }
#endif
#elif LPHQ_S8_SINK == 1
#if LPHQ_S8_KIND == 0
hipError_t launch_stream8_runs_p(LPHQ_S8_PARAMS, const S8RunsArgs& sink, LPHQ_S8_TAIL) { return launch_s8<0, kSinkRuns>(LPHQ_S8_FWD, nullptr, tile_base, grid, stream, sink); }
#else
hipError_t launch_stream8_runs_u(LPHQ_S8_PARAMS, const S8RunsArgs& sink, LPHQ_S8_TAIL) { return launch_s8<1, kSinkRuns>(LPHQ_S8_FWD, nullptr, tile_base, grid, stream, sink); }
#endif
#else
#if LPHQ_S8_KIND == 0
hipError_t launch_stream8_gather_p(LPHQ_S8_PARAMS, const S8GatherArgs& sink, LPHQ_S8_TAIL) { return launch_s8<0, kSinkGather>(LPHQ_S8_FWD, nullptr, tile_base, grid, stream, sink); }
#else
hipError_t launch_stream8_gather_u(LPHQ_S8_PARAMS, const S8GatherArgs& sink, LPHQ_S8_TAIL) { return launch_s8<1, kSinkGather>(LPHQ_S8_FWD, nullptr, tile_base, grid, stream, sink); }
#endif
#endif

}  // namespace lphq
