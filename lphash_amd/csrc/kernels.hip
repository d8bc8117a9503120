// kernels.hip -- launchers of the gfx950 kernels (kernels_device.cuh) behind kernels.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"
#include "kernels_device.cuh"

namespace lphq {

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static inline int grid_for(size_t n, int per_block, int max_blocks) {
    size_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (size_t)max_blocks) b = max_blocks;
    return (int)b;
}

#define LPHQ_DISPATCH(KERNEL, ...)                                                     \
    do {                                                                               \
        const bool k128 = H.kmer_bits == 128;                                          \
        const bool p6 = H.P == 6;                                                      \
        if (H.kind == 3) { /* competitor harness: BBHash over k-mers */                \
            if (k128) KERNEL<3, true, 4> __VA_ARGS__; else KERNEL<3, false, 4> __VA_ARGS__; \
        } else if (H.kind == 2) { /* competitor harness: plain PTHash over k-mers */   \
            if (k128) KERNEL<2, true, 4> __VA_ARGS__; else KERNEL<2, false, 4> __VA_ARGS__; \
        } else if (H.kind == 0) {                                                      \
            if (k128) { if (p6) KERNEL<0, true, 6> __VA_ARGS__; else KERNEL<0, true, 4> __VA_ARGS__; }   \
            else      { if (p6) KERNEL<0, false, 6> __VA_ARGS__; else KERNEL<0, false, 4> __VA_ARGS__; } \
        } else {                                                                       \
            if (k128) { if (p6) KERNEL<1, true, 6> __VA_ARGS__; else KERNEL<1, true, 4> __VA_ARGS__; }   \
            else      { if (p6) KERNEL<1, false, 6> __VA_ARGS__; else KERNEL<1, false, 4> __VA_ARGS__; } \
        }                                                                              \
    } while (0)

hipError_t launch_query_packed(const lphq_image_header& H, const uint8_t* d_img, const void* d_kmers, size_t n, uint64_t* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    // one k-mer per thread measured best (sweep: profiles/r01_packed_grid_sweep.txt).  HIP requires gridDim.x * blockDim.x < 2^32,
    // so at most 2^24 - 1 workgroups of 256; larger batches use the kernel's grid-stride loop.  LPHQ_PACKED_GRID (tuning aid)
    // can only lower the limit; values < 1 are ignored.
    constexpr int kMaxBlocks = (1 << 24) - 1;
    static const int max_blocks = [] {
        const char* e = getenv("LPHQ_PACKED_GRID");
        const long v = e ? atol(e) : 0;
        return (v >= 1 && v < kMaxBlocks) ? (int)v : kMaxBlocks;
    }();
    const int grid = grid_for(n, kBlock, max_blocks);
    LPHQ_DISPATCH(k_query_packed, <<<grid, kBlock, 0, stream>>>(H, d_img, (const uint64_t*)d_kmers, n, d_out));
    return hipGetLastError();
}

hipError_t launch_pack_kmers(const lphq_image_header& H, const char* d_seq, const uint64_t* d_pos, size_t n, void* d_kmers, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int grid = grid_for(n, kBlock, 256 * 16);
    if (H.kmer_bits == 128) k_pack_kmers<true><<<grid, kBlock, 0, stream>>>(d_seq, d_pos, n, H.k, (uint64_t*)d_kmers);
    else k_pack_kmers<false><<<grid, kBlock, 0, stream>>>(d_seq, d_pos, n, H.k, (uint64_t*)d_kmers);
    return hipGetLastError();
}

hipError_t launch_count_valid(const uint64_t* d_codes, size_t n, unsigned long long* d_count, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    k_count_valid<><<<grid_for(n, kBlock * 8, 256 * 8), kBlock, 0, stream>>>(d_codes, n, d_count);
    return hipGetLastError();
}

hipError_t launch_count_bad_bases(const char* d_seq, size_t n, unsigned long long* d_count, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    k_count_bad_bases<><<<grid_for(n, kBlock * 16, 256 * 8), kBlock, 0, stream>>>((const uint8_t*)d_seq, n, d_count);
    return hipGetLastError();
}

hipError_t launch_gather_payload(const uint64_t* d_codes, size_t n, const void* d_table, uint64_t n_slots, uint32_t elem_bytes, uint64_t missing, void* d_out,
                                 hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int grid = grid_for(n, kBlock, (1 << 24) - 1);
    switch (elem_bytes) {
        case 1: k_gather_payload<uint8_t><<<grid, kBlock, 0, stream>>>(d_codes, n, (const uint8_t*)d_table, n_slots, (uint8_t)missing, (uint8_t*)d_out); break;
        case 2: k_gather_payload<uint16_t><<<grid, kBlock, 0, stream>>>(d_codes, n, (const uint16_t*)d_table, n_slots, (uint16_t)missing, (uint16_t*)d_out); break;
        case 4: k_gather_payload<uint32_t><<<grid, kBlock, 0, stream>>>(d_codes, n, (const uint32_t*)d_table, n_slots, (uint32_t)missing, (uint32_t*)d_out); break;
        case 8: k_gather_payload<uint64_t><<<grid, kBlock, 0, stream>>>(d_codes, n, (const uint64_t*)d_table, n_slots, missing, (uint64_t*)d_out); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_query_records(const lphq_image_header& H, const uint8_t* d_img, const char* d_seq, uint64_t total_bases, const uint64_t* d_seq_off,
                                const uint64_t* d_out_off, size_t n_records, uint64_t* d_out, int streaming, hipStream_t stream) {
    if (n_records == 0 || total_bases < H.k) return hipSuccess;
    const uint64_t shift = (uint64_t)((uintptr_t)d_seq & 15u);
    const uint8_t* seq_al = (const uint8_t*)d_seq - shift;
    const uint64_t span = shift + total_bases;
    // windows at least this wide take the eight-positions-per-lane kernel (measured: profiles/r03_stream8.txt).  LPHQ_STREAM8_MINW (tuning aid, read once,
    // in [1, 64]; 64 = never) moves the threshold; it changes no result
    static const uint32_t s8_min_w = [] {
        const char* e = getenv("LPHQ_STREAM8_MINW");
        const long v = e ? atol(e) : 0;
        return (uint32_t)((v >= 1 && v <= 64) ? v : kStream8MinW);
    }();
    constexpr int sblock = kStreamBlockDefault;  // 256 threads measured best in both rounds (128: -4 %, 512: -6 %, profiles/r02_stream_v4_ab.txt)
    const bool stream_kernel = streaming && H.kind < 2;  // (the competitor harness has one kernel for both values of `streaming`)
    const bool s8 = stream_kernel && H.W >= s8_min_w;
    const uint64_t tw = stream_kernel ? stream_tile_windows(H.W, 4 * sblock) : kTileW;  // (8 * kS8Block == 4 * sblock: both kernels cut the same tiles)
    static_assert(8 * 128 == 4 * sblock, "one tile geometry (stream8_device.cuh: kS8Block = 128 threads, eight positions each)");
    const uint64_t tiles = (span + tw - 1) / tw;
    if (n_records > 0xFFFFFFFFull) return hipErrorInvalidValue;
    // one workgroup per tile; HIP requires gridDim.x * blockDim.x < 2^32, so a batch of more than 2^24 - 1 tiles (~16 G bases) goes
    // out as several launches on the same stream (tiles are independent)
    const uint64_t kMaxTiles = (1u << 24) - 1;
    for (uint64_t tile_base = 0; tile_base < tiles; tile_base += kMaxTiles) {
        const int grid = (int)std::min<uint64_t>(kMaxTiles, tiles - tile_base);
        if (stream_kernel) {
            // W >= 8: eight positions per lane (stream8_device.cuh; same tile geometry, 128 threads); narrower windows, and anything below LPHQ_STREAM8_MINW in
            // A/B runs: four positions per lane
            if (s8) {
                const hipError_t e8 = H.kind == 0 ? launch_query_tiles_stream8_p(H, d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, (uint32_t)n_records, d_out, tile_base, grid, stream)
                                                  : launch_query_tiles_stream8_u(H, d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, (uint32_t)n_records, d_out, tile_base, grid, stream);
                if (e8 != hipSuccess) return e8;
                continue;
            }
#define LPHQ_STREAM_LAUNCH(KIND, K128) \
    do { \
        if (H.P != 6) k_query_tiles_stream<KIND, K128, 4, sblock, false><<<grid, sblock, 0, stream>>>(d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, (uint32_t)n_records, d_out, tile_base); \
        else if (H.W >= kBlockMinW) k_query_tiles_stream<KIND, K128, 6, sblock, true><<<grid, sblock, 0, stream>>>(d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, (uint32_t)n_records, d_out, tile_base); \
        else k_query_tiles_stream<KIND, K128, 6, sblock, false><<<grid, sblock, 0, stream>>>(d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, (uint32_t)n_records, d_out, tile_base); \
    } while (0)
            if (H.kind == 0) {
                if (H.kmer_bits == 128) LPHQ_STREAM_LAUNCH(0, true);
                else LPHQ_STREAM_LAUNCH(0, false);
            } else {
                if (H.kmer_bits == 128) LPHQ_STREAM_LAUNCH(1, true);
                else LPHQ_STREAM_LAUNCH(1, false);
            }
        } else {
            LPHQ_DISPATCH(k_query_tiles_random, <<<grid, kBlock, 0, stream>>>(d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, (uint32_t)n_records, d_out, tile_base));
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// The streaming pass into another sink: the tiles of launch_query_records(streaming), every W (the eight-per-lane kernel takes narrow windows through its
// raw-hash form; they are rare -- k - m + 1 < 8 -- and the sinks exist once)
template <typename SinkArgs, typename LaunchP, typename LaunchU>
static hipError_t launch_stream_sink(const lphq_image_header& H, const uint8_t* d_img, const char* d_seq, uint64_t total_bases, const uint64_t* d_seq_off,
                                     const uint64_t* d_out_off, size_t n_records, const SinkArgs& sink, hipStream_t stream, LaunchP lp, LaunchU lu) {
    if (H.kind >= 2 || n_records > 0xFFFFFFFFull) return hipErrorInvalidValue;
    if (n_records == 0 || total_bases < H.k) return hipSuccess;
    const uint64_t shift = (uint64_t)((uintptr_t)d_seq & 15u);
    const uint8_t* seq_al = (const uint8_t*)d_seq - shift;
    const uint64_t tw = stream_tile_windows(H.W, 8 * 128), tiles = (shift + total_bases + tw - 1) / tw;
    const uint64_t kMaxTiles = (1u << 24) - 1;
    for (uint64_t tile_base = 0; tile_base < tiles; tile_base += kMaxTiles) {
        const int grid = (int)std::min<uint64_t>(kMaxTiles, tiles - tile_base);
        const hipError_t e = H.kind == 0 ? lp(H, d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, (uint32_t)n_records, sink, tile_base, grid, stream)
                                         : lu(H, d_img, seq_al, shift, total_bases, d_seq_off, d_out_off, (uint32_t)n_records, sink, tile_base, grid, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_query_records_runs(const lphq_image_header& H, const uint8_t* d_img, const char* d_seq, uint64_t total_bases, const uint64_t* d_seq_off,
                                     const uint64_t* d_out_off, size_t n_records, const S8RunsArgs& sink, hipStream_t stream) {
    k_zero_run_counts<><<<1, 64, 0, stream>>>(sink.counts);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_stream_sink(H, d_img, d_seq, total_bases, d_seq_off, d_out_off, n_records, sink, stream, launch_stream8_runs_p, launch_stream8_runs_u);
}

hipError_t launch_query_records_gather(const lphq_image_header& H, const uint8_t* d_img, const char* d_seq, uint64_t total_bases, const uint64_t* d_seq_off,
                                       const uint64_t* d_out_off, size_t n_records, const S8GatherArgs& sink, hipStream_t stream) {
    return launch_stream_sink(H, d_img, d_seq, total_bases, d_seq_off, d_out_off, n_records, sink, stream, launch_stream8_gather_p, launch_stream8_gather_u);
}

hipError_t launch_expand_runs(const lphq_run* d_runs, uint64_t arena_runs, const uint64_t* d_codes, uint64_t arena_codes, const lphq_run_counts* d_counts, uint64_t* d_out,
                              uint64_t n_out, unsigned long long* d_status, hipStream_t stream) {
    hipError_t e = n_out ? hipMemsetAsync(d_out, 0xFF, n_out * sizeof(uint64_t), stream) : hipSuccess;  // LPHQ_INVALID: windows that belong to no run
    if (e == hipSuccess && d_status) e = hipMemsetAsync(d_status, 0, sizeof(uint64_t), stream);
    if (e != hipSuccess) return e;
    // (the grid is sized for full arenas: the counts live on the device; a workgroup whose slice of its arena is empty returns at once)
    const int gx = grid_for((size_t)std::min<uint64_t>(arena_runs, (uint64_t)1 << 31), kBlock, 1024);
    k_expand_runs<><<<dim3((unsigned)gx, LPHQ_RUN_ARENAS), kBlock, 0, stream>>>(d_runs, arena_runs, d_codes, arena_codes, d_counts, d_out, n_out, d_status);
    return hipGetLastError();
}

hipError_t launch_compact_runs(const lphq_run* d_runs, uint64_t arena_runs, const uint64_t* d_codes, uint64_t arena_codes, const lphq_run_counts* d_counts,
                               lphq_run* d_dense_runs, uint64_t cap_dense_runs, uint64_t* d_dense_codes, uint64_t cap_dense_codes, unsigned long long* d_totals,
                               hipStream_t stream) {
    const hipError_t e = hipMemsetAsync(d_totals, 0, 4 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const int gx = grid_for((size_t)std::min<uint64_t>(std::max(arena_runs, arena_codes), (uint64_t)1 << 31), kBlock * 4, 256);
    k_compact_runs<><<<dim3((unsigned)gx, LPHQ_RUN_ARENAS), kBlock, 0, stream>>>(d_runs, arena_runs, d_codes, arena_codes, d_counts, d_dense_runs, cap_dense_runs, d_dense_codes,
                                                                             cap_dense_codes, d_totals);
    return hipGetLastError();
}

}  // namespace lphq
