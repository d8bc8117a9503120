// stream8.hip -- launcher of the eight-positions-per-lane streaming kernel (stream8_device.cuh).  A translation unit of its own, compiled once per function
// kind (-DLPHQ_S8_KIND=0 / 1: partitioned / unpartitioned), because the wide-window variant exists once per value of (W - 8) % 8: 2 x 2 x (8 + 1) kernels
// that make builds in parallel instead of one long compile of kernels.hip.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "stream8_device.cuh"

namespace lphq {

template <int KIND>
static hipError_t launch_s8(const lphq_image_header& H, const uint8_t* d_img, const uint8_t* seq_al, uint64_t shift, uint64_t total_bases, const uint64_t* d_seq_off,
                            const uint64_t* d_out_off, uint32_t n_records, uint64_t* d_out, uint64_t tile_base, int grid, hipStream_t stream) {
#define LPHQ_S8_ARGS <<<grid, kS8Block, 0, stream>>>(d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, n_records, d_out, tile_base)
    if (H.P == 6) {  // W >= 16: block decomposition, one instantiation per (W - 8) % 8
        if (H.kmer_bits == 128) {
#define LPHQ_S8_CALL(W8) k_query_tiles_stream8<KIND, true, 6, kS8Block, false, W8> LPHQ_S8_ARGS
            LPHQ_S8_SWITCH_W8(H.W, LPHQ_S8_CALL)
#undef LPHQ_S8_CALL
        } else {
#define LPHQ_S8_CALL(W8) k_query_tiles_stream8<KIND, false, 6, kS8Block, false, W8> LPHQ_S8_ARGS
            LPHQ_S8_SWITCH_W8(H.W, LPHQ_S8_CALL)
#undef LPHQ_S8_CALL
        }
    } else if (H.kmer_bits == 128) {
        k_query_tiles_stream8<KIND, true, 4> LPHQ_S8_ARGS;
    } else {
        k_query_tiles_stream8<KIND, false, 4> LPHQ_S8_ARGS;
    }
#undef LPHQ_S8_ARGS
    return hipGetLastError();
}

#if LPHQ_S8_KIND == 0
hipError_t launch_query_tiles_stream8_p(const lphq_image_header& H, const uint8_t* d_img, const uint8_t* seq_al, uint64_t shift, uint64_t total_bases,
                                        const uint64_t* d_seq_off, const uint64_t* d_out_off, uint32_t n_records, uint64_t* d_out, uint64_t tile_base, int grid,
                                        hipStream_t stream) {
    return launch_s8<0>(H, d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, n_records, d_out, tile_base, grid, stream);
}
#else
hipError_t launch_query_tiles_stream8_u(const lphq_image_header& H, const uint8_t* d_img, const uint8_t* seq_al, uint64_t shift, uint64_t total_bases,
                                        const uint64_t* d_seq_off, const uint64_t* d_out_off, uint32_t n_records, uint64_t* d_out, uint64_t tile_base, int grid,
                                        hipStream_t stream) {
    return launch_s8<1>(H, d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, n_records, d_out, tile_base, grid, stream);
}
#endif

}  // namespace lphq
