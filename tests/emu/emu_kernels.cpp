// TEST INFRASTRUCTURE ONLY.  The product's kernels (lphash_amd/csrc/kernels_device.cuh) compiled for the host and run by the
// fiber emulator, with the launch geometry of lphash_amd/csrc/kernels.hip restated here.
#include <cstring>
#include <vector>

#include "kernels_device.cuh"  // -Itests/emu resolves <hip/hip_runtime.h> to the stub
#include "stream8_device.cuh"
#include "extract_device.cuh"

using namespace lphq;

#define EMU_DISPATCH(CALL_B)                                                                    \
    do {                                                                                        \
        const bool k128 = H.kmer_bits == 128, p6 = H.P == 6;                                    \
        if (H.kind == 3) {                                                                      \
            if (k128) { CALL_B(3, true, 4); } else { CALL_B(3, false, 4); }                     \
        } else if (H.kind == 2) {                                                               \
            if (k128) { CALL_B(2, true, 4); } else { CALL_B(2, false, 4); }                     \
        } else if (H.kind == 0) {                                                                \
            if (k128) { if (p6) { CALL_B(0, true, 6); } else { CALL_B(0, true, 4); } }          \
            else      { if (p6) { CALL_B(0, false, 6); } else { CALL_B(0, false, 4); } }        \
        } else {                                                                                \
            if (k128) { if (p6) { CALL_B(1, true, 6); } else { CALL_B(1, true, 4); } }          \
            else      { if (p6) { CALL_B(1, false, 6); } else { CALL_B(1, false, 4); } }        \
        }                                                                                       \
    } while (0)

static bool header(const uint8_t* img, size_t nbytes, lphq_image_header& H) {
    if (nbytes < sizeof H) return false;
    memcpy(&H, img, sizeof H);
    return H.magic == LPHQ_IMAGE_MAGIC && H.total_bytes == nbytes;
}

extern "C" void emu_set_schedule(int mode, uint64_t seed) { emu::set_schedule(mode, seed); }

extern "C" int emu_kernel_query_packed(const uint8_t* img, size_t nbytes, const uint64_t* kmers, size_t n, uint64_t* out) {
    lphq_image_header H;
    if (!header(img, nbytes, H)) return 1;
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
#define CALL(KIND, K128, P) emu::launch(grid, kBlock, [&] { k_query_packed<KIND, K128, P>(H, img, kmers, n, out); })
    EMU_DISPATCH(CALL);
#undef CALL
    return 0;
}

// seq must be 16-byte aligned minus `misalign` (the caller passes base + misalign), padded with >= 64 readable bytes
extern "C" int emu_kernel_query_records(const uint8_t* img, size_t nbytes, const char* seq, uint64_t total, const uint64_t* seq_off, const uint64_t* out_off,
                                        uint32_t n_records, uint64_t* out, int streaming, int block) {
    lphq_image_header H;
    if (!header(img, nbytes, H)) return 1;
    if (n_records == 0 || total < H.k) return 0;
    const uint64_t shift = (uint64_t)((uintptr_t)seq & 15u);
    const uint8_t* seq_al = (const uint8_t*)seq - shift;
    const uint64_t span = shift + total;
    if (streaming && H.kind < 2) {
        /* eight positions per lane (stream8_device.cuh): forced by 0x40000, or -- as the launcher does -- from W = kStream8MinW on unless a form is forced */
        const bool s8 = (block & 0x40000) || (!(block & 0x30000) && H.W >= kStream8MinW);  /* (P == 4: window minima from the raw hashes, any W; P == 6: block form) */
        const bool s8_256 = s8 && (block & 0xFFFF) == 256 && (block & 0x40000);  /* forced eight-per-lane with 256 threads: 2048 positions per tile */
        const uint64_t tw = stream_tile_windows(H.W, s8 ? (s8_256 ? 8 * 256 : 8 * kS8Block) : 4 * (block & 0xFFFF));
        const unsigned tiles = (unsigned)((span + tw - 1) / tw);
#define EMU_S8_W8(W8V) emu::launch(tiles, kS8Block, [&] { k_query_tiles_stream8<KIND_, K128_, 6, kS8Block, false, W8V>(img, seq_al, shift, total, seq_off, out_off, n_records, out, 0, S8NoArgs{}); });
#define CALL(KIND, K128, P)                                                                                                                             \
    do {                                                                                                                                                \
        constexpr int KIND_ = KIND;                                                                                                                     \
        constexpr bool K128_ = K128;                                                                                                                    \
        const bool blk = (block & 0x10000) ? true : ((block & 0x20000) ? false : H.W >= kBlockMinW); /* forced either way, or as the launcher chooses */ \
        const int bs = block & 0xFFFF;                                                                                                                    \
        if (s8) {                                                                                                                                         \
            if (s8_256) emu::launch(tiles, 256, [&] { k_query_tiles_stream8<KIND, K128, P, 256>(img, seq_al, shift, total, seq_off, out_off, n_records, out, 0, S8NoArgs{}); }); \
            else if (P == 6 && H.W >= 16) { /* the wide-window variant as the launcher picks it: one instantiation per (W - 8) % 8 */                          \
                LPHQ_S8_SWITCH_W8(H.W, EMU_S8_W8)                                                                                                                \
            } else emu::launch(tiles, kS8Block, [&] { k_query_tiles_stream8<KIND, K128, P>(img, seq_al, shift, total, seq_off, out_off, n_records, out, 0, S8NoArgs{}); });  \
        } else if (H.W >= 4 && blk) {                                                                                                                            \
            if (bs == 128) emu::launch(tiles, 128, [&] { k_query_tiles_stream<KIND, K128, P, 128, true>(img, seq_al, shift, total, seq_off, out_off, n_records, out, 0); }); \
            else emu::launch(tiles, 256, [&] { k_query_tiles_stream<KIND, K128, P, 256, true>(img, seq_al, shift, total, seq_off, out_off, n_records, out, 0); });            \
        } else if (bs == 128) emu::launch(tiles, 128, [&] { k_query_tiles_stream<KIND, K128, P, 128, false>(img, seq_al, shift, total, seq_off, out_off, n_records, out, 0); }); \
        else emu::launch(tiles, 256, [&] { k_query_tiles_stream<KIND, K128, P, 256, false>(img, seq_al, shift, total, seq_off, out_off, n_records, out, 0); });               \
    } while (0)
        EMU_DISPATCH(CALL);
#undef CALL
#undef EMU_S8_W8
    } else {
        const unsigned tiles = (unsigned)((span + kTileW - 1) / kTileW);
#define CALL(KIND, K128, P) emu::launch(tiles, kBlock, [&] { k_query_tiles_random<KIND, K128, P>(img, seq_al, shift, total, seq_off, out_off, n_records, out, 0); })
        EMU_DISPATCH(CALL);
#undef CALL
    }
    return 0;
}

extern "C" int emu_kernel_pack_kmers(const char* seq, const uint64_t* pos, size_t n, uint32_t k, int k128, uint64_t* out) {
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    if (k128) emu::launch(grid, kBlock, [&] { k_pack_kmers<true>(seq, pos, n, k, out); });
    else emu::launch(grid, kBlock, [&] { k_pack_kmers<false>(seq, pos, n, k, out); });
    return 0;
}

extern "C" int emu_kernel_count_valid(const uint64_t* codes, size_t n, unsigned long long* count) {
    *count = 0;
    emu::launch((unsigned)((n + kBlock * 8 - 1) / (kBlock * 8)), kBlock, [&] { k_count_valid<>(codes, n, count); });
    return 0;
}

extern "C" int emu_kernel_gather_payload(const uint64_t* codes, size_t n, const void* table, uint64_t n_slots, int elem_bytes, uint64_t missing, void* out) {
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    if (elem_bytes == 4) emu::launch(grid, kBlock, [&] { k_gather_payload<uint32_t>(codes, n, (const uint32_t*)table, n_slots, (uint32_t)missing, (uint32_t*)out); });
    else if (elem_bytes == 8) emu::launch(grid, kBlock, [&] { k_gather_payload<uint64_t>(codes, n, (const uint64_t*)table, n_slots, missing, (uint64_t*)out); });
    else if (elem_bytes == 2) emu::launch(grid, kBlock, [&] { k_gather_payload<uint16_t>(codes, n, (const uint16_t*)table, n_slots, (uint16_t)missing, (uint16_t*)out); });
    else emu::launch(grid, kBlock, [&] { k_gather_payload<uint8_t>(codes, n, (const uint8_t*)table, n_slots, (uint8_t)missing, (uint8_t*)out); });
    return 0;
}

// k_extract_runs as extract.hip launches it.  seq: 16-byte aligned, >= 64 readable bytes on both sides.  Returns the number of runs the kernel
// produced (it may exceed cap: then only cap of them were written), or -1.
extern "C" long long emu_kernel_extract_runs(const char* seq, uint64_t total, const uint64_t* seq_off, uint32_t n_records, uint32_t k, uint32_t m, uint64_t mm_seed,
                                             uint64_t first_base, uint64_t* out_mm, uint64_t* out_val /* 2 words per run: first, p1 | size << 32 */, uint64_t cap,
                                             int force_form /* 0: as the launcher picks, 1: block minima, 2: direct */, unsigned long long* lost_windows) {
    if (((uintptr_t)seq & 15u) || total < k || n_records == 0) return -1;
    static_assert(sizeof(RunVal) == 16, "two words per run");
    const uint64_t shift = 16;
    const uint8_t* seq_al = (const uint8_t*)seq - shift;
    const uint32_t W = k - m + 1u;
    const uint32_t S = extract_tile_stride(W, 4 * kStreamBlockDefault);
    const unsigned tiles = (unsigned)((shift + total - k) / S + 1);
    unsigned long long cursor = 0;
    *lost_windows = 0;
    const bool blk = force_form ? force_form == 1 : W >= kBlockMinW;
    if (blk && W < 4) return -1;
    if (blk)
        emu::launch(tiles, kStreamBlockDefault, [&] {
            k_extract_runs<kStreamBlockDefault, true>(seq_al, shift, total, seq_off, n_records, k, m, mm_seed, first_base, out_mm, (RunVal*)out_val, &cursor, cap, 0, lost_windows);
        });
    else
        emu::launch(tiles, kStreamBlockDefault, [&] {
            k_extract_runs<kStreamBlockDefault, false>(seq_al, shift, total, seq_off, n_records, k, m, mm_seed, first_base, out_mm, (RunVal*)out_val, &cursor, cap, 0, lost_windows);
        });
    return (long long)cursor;
}

// k_group_flags / k_group_scatter as extract.hip runs them (the exclusive scan between them is the library's on the GPU, a loop here).
// val: 2 words per record (first, p1 | size << 32).  Returns the number of distinct minimizers; *n_coll, *kmers as the device path reports them.
extern "C" long long emu_kernel_classify(const uint64_t* mm_sorted, const uint64_t* val, size_t n, uint64_t* uniq_mm, uint16_t* uniq_offset_size, uint64_t* coll_first,
                                         uint32_t* coll_size, uint64_t* n_coll, unsigned long long* kmers) {
    std::vector<uint64_t> flags(n + 1, 0), slots(n + 1, 0);
    *kmers = 0;
    const unsigned grid = (unsigned)((n + 255) / 256);
    emu::launch(grid, 256, [&] { k_group_flags(mm_sorted, (const RunVal*)val, n, flags.data(), kmers); });
    uint64_t acc = 0;
    for (size_t i = 0; i < n; ++i) {
        slots[i] = acc;
        acc += flags[i];
    }
    emu::launch(grid, 256, [&] { k_group_scatter(mm_sorted, (const RunVal*)val, slots.data(), n, uniq_mm, uniq_offset_size, coll_first, coll_size); });
    *n_coll = acc >> 32;
    return (long long)(acc & 0xFFFFFFFFull);
}
