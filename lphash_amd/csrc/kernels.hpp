// kernels.hpp -- host-callable launchers of the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "image_layout.h"
#include "stream_sinks.h"

namespace lphq {
hipError_t launch_query_packed(const lphq_image_header& H, const uint8_t* d_img, const void* d_kmers, size_t n, uint64_t* d_out, hipStream_t stream);
hipError_t launch_pack_kmers(const lphq_image_header& H, const char* d_seq, const uint64_t* d_pos, size_t n, void* d_kmers, hipStream_t stream);
hipError_t launch_query_records(const lphq_image_header& H, const uint8_t* d_img, const char* d_seq, uint64_t total_bases, const uint64_t* d_seq_off,
                                const uint64_t* d_out_off, size_t n_records, uint64_t* d_out, int streaming, hipStream_t stream);
hipError_t launch_gather_payload(const uint64_t* d_codes, size_t n, const void* d_table, uint64_t n_slots, uint32_t elem_bytes, uint64_t missing, void* d_out,
                                 hipStream_t stream);
hipError_t launch_count_valid(const uint64_t* d_codes, size_t n, unsigned long long* d_count, hipStream_t stream);
hipError_t launch_count_bad_bases(const char* d_seq, size_t n, unsigned long long* d_count, hipStream_t stream);
// stream8.hip (one object per function kind): the eight-positions-per-lane streaming kernel over the tiles [tile_base, tile_base + grid)
hipError_t launch_query_tiles_stream8_p(const lphq_image_header& H, const uint8_t* d_img, const uint8_t* seq_al, uint64_t shift, uint64_t total_bases,
                                        const uint64_t* d_seq_off, const uint64_t* d_out_off, uint32_t n_records, uint64_t* d_out, uint64_t tile_base, int grid,
                                        hipStream_t stream);
hipError_t launch_query_tiles_stream8_u(const lphq_image_header& H, const uint8_t* d_img, const uint8_t* seq_al, uint64_t shift, uint64_t total_bases,
                                        const uint64_t* d_seq_off, const uint64_t* d_out_off, uint32_t n_records, uint64_t* d_out, uint64_t tile_base, int grid,
                                        hipStream_t stream);
// the streaming pass with the run-encoded result / fused with the payload gather (stream8.hip with -DLPHQ_S8_SINK=1 / 2: any W >= 1)
hipError_t launch_stream8_runs_p(const lphq_image_header& H, const uint8_t* d_img, const uint8_t* seq_al, uint64_t shift, uint64_t total_bases, const uint64_t* d_seq_off,
                                 const uint64_t* d_out_off, uint32_t n_records, const S8RunsArgs& sink, uint64_t tile_base, int grid, hipStream_t stream);
hipError_t launch_stream8_runs_u(const lphq_image_header& H, const uint8_t* d_img, const uint8_t* seq_al, uint64_t shift, uint64_t total_bases, const uint64_t* d_seq_off,
                                 const uint64_t* d_out_off, uint32_t n_records, const S8RunsArgs& sink, uint64_t tile_base, int grid, hipStream_t stream);
hipError_t launch_stream8_gather_p(const lphq_image_header& H, const uint8_t* d_img, const uint8_t* seq_al, uint64_t shift, uint64_t total_bases, const uint64_t* d_seq_off,
                                   const uint64_t* d_out_off, uint32_t n_records, const S8GatherArgs& sink, uint64_t tile_base, int grid, hipStream_t stream);
hipError_t launch_stream8_gather_u(const lphq_image_header& H, const uint8_t* d_img, const uint8_t* seq_al, uint64_t shift, uint64_t total_bases, const uint64_t* d_seq_off,
                                   const uint64_t* d_out_off, uint32_t n_records, const S8GatherArgs& sink, uint64_t tile_base, int grid, hipStream_t stream);
// kernels.hip: the same batch geometry as launch_query_records(streaming); d_counts is zeroed on `stream` first
hipError_t launch_query_records_runs(const lphq_image_header& H, const uint8_t* d_img, const char* d_seq, uint64_t total_bases, const uint64_t* d_seq_off,
                                     const uint64_t* d_out_off, size_t n_records, const S8RunsArgs& sink, hipStream_t stream);
hipError_t launch_query_records_gather(const lphq_image_header& H, const uint8_t* d_img, const char* d_seq, uint64_t total_bases, const uint64_t* d_seq_off,
                                       const uint64_t* d_out_off, size_t n_records, const S8GatherArgs& sink, hipStream_t stream);
hipError_t launch_expand_runs(const lphq_run* d_runs, uint64_t arena_runs, const uint64_t* d_codes, uint64_t arena_codes, const lphq_run_counts* d_counts, uint64_t* d_out,
                              uint64_t n_out, unsigned long long* d_status, hipStream_t stream);
hipError_t launch_compact_runs(const lphq_run* d_runs, uint64_t arena_runs, const uint64_t* d_codes, uint64_t arena_codes, const lphq_run_counts* d_counts,
                               lphq_run* d_dense_runs, uint64_t cap_dense_runs, uint64_t* d_dense_codes, uint64_t cap_dense_codes, unsigned long long* d_totals,
                               hipStream_t stream);
}  // namespace lphq
