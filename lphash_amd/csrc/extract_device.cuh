// extract_device.cuh -- index construction, part 1 on the GPU: the super-k-mers of a batch of records.
//
// The reference walks every record base by base (minimizer::from_string, include/minimizer.hpp:11-170) and pushes one
// mm_record_t {minimizer, id, p1, size} whenever the leftmost-minimum m-mer of the window changes.  Position-parallel, as the
// streaming query kernel does it (kernels_device.cuh, phases A-C): a tile of window starts, one MurmurHash2_64 per m-mer
// position, the leftmost minimum of every window from LDS.  A super-k-mer is a maximal run of consecutive valid windows that
// share the POSITION of their minimizer; such a run is at most W = k - m + 1 windows long (the position must lie inside every
// window of the run).  A tile therefore computes one window before and W windows after the S window starts it owns, and emits
// every run that BEGINS at one of its own starts whole: no run is cut at a tile border, nothing has to be merged afterwards.
// Runs are appended through one global counter per tile; their order is arbitrary (the builder sorts them by minimizer next).
// Windows holding a non-ACGT base belong to no run and end the run before them (the reference restarts there,
// minimizer.hpp:150-164).  Pure integer work, wave64; no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "extract.hpp"
#include "kernels_device.cuh"

namespace lphq {

// window starts owned by one tile: TW computed windows minus one before and W after, 16-aligned (the tile origin stays aligned)
__host__ __device__ inline uint32_t extract_tile_stride(uint32_t W, int npos) { return (stream_tile_windows(W, npos) - 1u - W) & ~15u; }

// shift: seq_al + shift is the first base of the batch; shift >= 16 and seq_al 16-byte aligned, with the bytes from seq_al on readable
// (the caller owns the buffer: 64 spare bytes on both sides).  Tile t owns the window starts with shifted coordinate t*S + 1 .. t*S + S.
template <int BLOCK, bool BLKMIN>
__global__ __launch_bounds__(BLOCK) void k_extract_runs(const uint8_t* __restrict__ seq_al, uint64_t shift, uint64_t total, const uint64_t* __restrict__ seq_off,
                                                          uint32_t n_records, uint32_t k, uint32_t m, uint64_t mm_seed,
                                                          uint64_t first_base /* global coordinate of the batch's first base */, uint64_t* __restrict__ out_mm,
                                                          RunVal* __restrict__ out_val, unsigned long long* __restrict__ cursor, uint64_t cap,
                                                          uint64_t tile_base /* first tile of this launch */,
                                                          unsigned long long* __restrict__ lost_windows /* += valid windows the reference leaves out of every run */) {
    constexpr int kPos = 4 * BLOCK, NW = tile_words(kPos);
    constexpr int NBADW = (NW + 8) / 2 + 2;
    constexpr int NDW = (NW + 1) * 4;
    static_assert(NDW > BLOCK && NDW <= 2 * BLOCK, "two sequence dwords per lane at most");
    __shared__ uint64_t sm_hash[hash_slots<BLOCK>()];
    __shared__ uint32_t sm_words[NW + 8];
    __shared__ uint32_t sm_bad[NBADW];
    __shared__ uint16_t sm_code[kPos + 4];    // minimizer position of every window of the tile; >= 0xFFFE: no window / holds a non-ACGT base
    __shared__ uint32_t sm_bound[kPos / 32 + 3];  // bit i: window i is the first of a run, or no window
    __shared__ uint8_t sm_off[hash_row<BLOCK>() + 3];
    __shared__ uint32_t sm_rec[2], sm_count, sm_anybad;
    __shared__ unsigned long long sm_gbase;
    __shared__ uint64_t sm_roff[2];
    __shared__ uint64_t sm_rtab[2 * kRecCap + 1];
    const uint32_t tid = threadIdx.x;
    const uint32_t W = k - m + 1u;
    const uint32_t TW = stream_tile_windows(W, kPos), S = extract_tile_stride(W, kPos);
    const uint64_t t0 = (tile_base + blockIdx.x) * S;  // shifted coordinate of the tile's window 0 (the look-behind window)

    // ---- A: bases -> 2-bit words + bad-base bits in LDS; records under the tile ----
    uint32_t xs[2] = {0u, 0u};
    bool have[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t d = tid + (uint32_t)r * BLOCK;
        const uint64_t a = t0 + (uint64_t)d * 4u;
        have[r] = d < (uint32_t)NDW && a < shift + total && a + 4u > shift;
        if (have[r]) xs[r] = *(const uint32_t*)(seq_al + a);
    }
    if (tid < 64) {
        const uint64_t gfirst = t0 > shift ? t0 - shift : 0;
        uint64_t glast = t0 + TW + 64 > shift ? t0 + TW + 64 - shift : 0;
        if (glast >= total) glast = total ? total - 1 : 0;
        const uint32_t r = find_record_wave(seq_off, n_records, gfirst);
        const uint64_t s0 = seq_off[r], e0 = seq_off[r + 1];
        const uint32_t r2 = glast < e0 ? r : find_record_wave(seq_off, n_records, glast);
        if (tid == 0) {
            sm_rec[0] = r;
            sm_rec[1] = r2;
            sm_roff[0] = s0;
            sm_roff[1] = e0;
            sm_count = 0;
            sm_anybad = 0;
        }
    }
    for (int c = tid; c < NBADW; c += BLOCK) sm_bad[c] = (c < NDW / 8) ? 0u : 0xFFFFFFFFu;
    for (int c = NW + 1 + tid; c < NW + 8; c += BLOCK) sm_words[c] = 0;
    for (int c = tid; c < kPos / 32 + 3; c += BLOCK) sm_bound[c] = (c < kPos / 32) ? 0u : 0xFFFFFFFFu;  // beyond the tile: no windows
    __syncthreads();
    if (sm_rec[0] != sm_rec[1]) load_record_table(sm_rtab, seq_off, seq_off, sm_rec[0], sm_rec[1], tid);  // (only the seq_off half is used)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t d = tid + (uint32_t)r * BLOCK;
        if (d < (uint32_t)NDW) {
            const uint64_t a = t0 + (uint64_t)d * 4u;
            uint32_t p8 = 0, b4 = 0xFu;
            if (have[r]) {
                p8 = pack4(xs[r], b4);
                uint32_t outside = 0;
                if (a < shift || a + 4u > shift + total)
                    for (int i = 0; i < 4; ++i)
                        if (a + i < shift || a + i >= shift + total) outside |= 1u << i;
                if (b4 & ~outside) sm_anybad = 1u;
                b4 |= outside;
            }
            ((uint8_t*)sm_words)[(d & ~3u) + (3u - (d & 3u))] = (uint8_t)p8;
            if (b4) atomicOr(&sm_bad[d >> 3], b4 << (4 * (d & 7)));
        }
    }
    __syncthreads();

    // ---- B: one hash per m-mer position (4 per lane) ----
    BlockMin bmin;
    {
        const uint64_t mm_seedx = murmur_seedx(mm_seed);
        const uint32_t a = 4u * tid;
        const uint32_t w0 = a >> 4, o = (a & 15u) * 2u;
        const uint32_t x0 = sm_words[w0], x1 = sm_words[w0 + 1], x2 = sm_words[w0 + 2], x3 = sm_words[w0 + 3];
        const uint64_t A64 = ((uint64_t)x0 << 32) | x1, C64 = ((uint64_t)x2 << 32) | x3;
        const uint64_t s0 = o ? ((A64 << o) | ((uint64_t)x2 >> (32 - o))) : A64;
        const uint64_t s1 = o ? (C64 << o) : C64;
        const uint32_t sh = 64u - 2u * m;
        const uint64_t mask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
        uint64_t h[4];
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint64_t v = r == 0 ? (s0 >> sh) : (((s0 << (2 * r)) | (s1 >> (64 - 2 * r))) >> sh);
            h[r] = murmur8(v & mask, mm_seedx);
        }
        if (BLKMIN) {
            bmin = publish_block_minima<BLOCK>(sm_hash, sm_off, tid, h);
        } else {
            bmin = BlockMin{0, 0, 0, 0};
#pragma unroll
            for (uint32_t r = 0; r < 4; ++r) sm_hash[r * hash_row<BLOCK>() + tid] = h[r];
        }
        hash_pad<BLOCK>(sm_hash, tid);
        if (tid < 20u) sm_off[(uint32_t)BLOCK + tid] = 0;
    }
    __syncthreads();

    // ---- C: minimizer position and validity of the lane's 4 windows ----
    const uint32_t a = 4u * tid;
    const uint32_t rlo = sm_rec[0], rhi = sm_rec[1];
    uint32_t code[4];
    {
        uint32_t q[4];
        if (BLKMIN) window_argmin4<BLOCK>(sm_hash, sm_off, tid, W, bmin, q);
        else window_argmin4_raw<BLOCK>(sm_hash, tid, W, q);
        if (rlo == rhi) {
            const uint32_t w_lo = t0 >= shift ? 0u : (uint32_t)(shift - t0);
            uint32_t w_hi = 0;
            const uint64_t limit = sm_roff[1] + shift;
            if (limit >= t0 + k) w_hi = (uint32_t)min((uint64_t)TW, limit - k - t0 + 1u);
#pragma unroll
            for (uint32_t jw = 0; jw < 4; ++jw) code[jw] = (a + jw >= w_lo && a + jw < w_hi) ? q[jw] : 0xFFFEu;
        } else {
            bool found = false;
            uint32_t jcur = 0;
            uint64_t rs = 0, re = 0;
#pragma unroll
            for (uint32_t jw = 0; jw < 4; ++jw) {
                const uint32_t i = a + jw;
                const uint64_t ga = t0 + i;
                uint32_t v = 0xFFFEu;
                if (i < TW && ga >= shift && ga - shift + k <= total) {
                    const uint64_t g = ga - shift;
                    if (!found) {
                        jcur = tile_record(sm_rtab, seq_off, rlo, rhi, g, rs, re);
                        found = true;
                    }
                    rec_walk(sm_rtab, seq_off, rlo, rhi - rlo < (uint32_t)kRecCap, g, jcur, rs, re);
                    if (g + k <= re) v = q[jw];
                }
                code[jw] = v;
            }
        }
        if (sm_anybad != 0u) {
            const uint64_t kmask = (1ull << k) - 1ull;
            const uint32_t bw = a >> 5, bo = a & 31u;
            const uint32_t b0 = sm_bad[bw], b1 = sm_bad[bw + 1], b2 = sm_bad[bw + 2], b3 = sm_bad[bw + 3];
            const uint64_t blo = bo ? (((((uint64_t)b1 << 32) | b0) >> bo) | ((uint64_t)b2 << (64u - bo))) : ((((uint64_t)b1) << 32) | b0);
            const uint32_t bhi = bo ? (uint32_t)(((((uint64_t)b3 << 32) | b2) >> bo)) : b2;
#pragma unroll
            for (uint32_t jw = 0; jw < 4; ++jw) {
                const uint64_t x = jw ? ((blo >> jw) | ((uint64_t)bhi << (64 - jw))) : blo;
                if (code[jw] != 0xFFFEu && (x & kmask)) code[jw] = 0xFFFFu;
            }
        }
#pragma unroll
        for (uint32_t jw = 0; jw < 4; ++jw) sm_code[a + jw] = (uint16_t)code[jw];
    }
    __syncthreads();

    // ---- D: run starts; a window that is none ends the run before it just as well ----
    bool lead[4];
    {
        const uint32_t left = a ? (uint32_t)sm_code[a - 1u] : 0xFFFEu;
        lead[0] = code[0] < 0xFFFEu && code[0] != left;
        lead[1] = code[1] < 0xFFFEu && code[1] != code[0];
        lead[2] = code[2] < 0xFFFEu && code[2] != code[1];
        lead[3] = code[3] < 0xFFFEu && code[3] != code[2];
        uint32_t nib = 0;
#pragma unroll
        for (uint32_t jw = 0; jw < 4; ++jw)
            if (lead[jw] || code[jw] >= 0xFFFEu) nib |= 1u << jw;
        if (nib) atomicOr(&sm_bound[a >> 5], nib << (a & 31u));
#pragma unroll
        for (uint32_t jw = 0; jw < 4; ++jw) lead[jw] = lead[jw] && a + jw >= 1u && a + jw <= S;  // the starts this tile owns
        // The reference's lost window (builder.cpp: lost_window): a stretch of EXACTLY k valid bases that a non-ACGT base closes inside its record has one
        // window; from_string never pushes it (minimizer.hpp:61-66,139-152) but counts its k-mer.  Here: a valid window whose left neighbour is no window
        // (record start, or the bad base is the one before the stretch) and whose right neighbour is a window of the same record that holds a bad base (it
        // can only be base w + k).  It starts no run -- it stays a boundary -- and is counted once, by the tile that owns it.
        if (sm_anybad != 0u) {
            uint32_t nlost = 0;
#pragma unroll
            for (uint32_t jw = 0; jw < 4; ++jw) {
                if (!lead[jw]) continue;
                const uint32_t lc = jw ? code[jw - 1] : left;
                const uint32_t rc = jw < 3 ? code[jw + 1] : (uint32_t)sm_code[a + 4u];
                if (lc >= 0xFFFEu && rc == 0xFFFFu) {
                    lead[jw] = false;
                    ++nlost;
                }
            }
            if (nlost) atomicAdd(lost_windows, (unsigned long long)nlost);
        }
    }
    uint32_t slot[4];
    {
        const uint32_t lane = tid & 63u;
        const uint64_t b0 = __ballot(lead[0]), b1 = __ballot(lead[1]), b2 = __ballot(lead[2]), b3 = __ballot(lead[3]);
        const uint32_t n0 = (uint32_t)__popcll(b0), n1 = (uint32_t)__popcll(b1), n2 = (uint32_t)__popcll(b2), n3 = (uint32_t)__popcll(b3);
        uint32_t base = 0;
        if (lane == 0 && n0 + n1 + n2 + n3) base = atomicAdd(&sm_count, n0 + n1 + n2 + n3);
        base = __shfl(base, 0, 64);
        const uint64_t lt = (1ull << lane) - 1ull;
        slot[0] = base + (uint32_t)__popcll(b0 & lt);
        slot[1] = base + n0 + (uint32_t)__popcll(b1 & lt);
        slot[2] = base + n0 + n1 + (uint32_t)__popcll(b2 & lt);
        slot[3] = base + n0 + n1 + n2 + (uint32_t)__popcll(b3 & lt);
    }
    __syncthreads();
    if (tid == 0) sm_gbase = sm_count ? atomicAdd(cursor, (unsigned long long)sm_count) : 0ull;
    __syncthreads();

    // ---- E: one record per owned run start: minimizer value, first window, offset, length up to the next boundary ----
    const unsigned long long gbase = sm_gbase;
#pragma unroll
    for (uint32_t jw = 0; jw < 4; ++jw) {
        if (!lead[jw]) continue;
        const uint64_t dst = gbase + slot[jw];
        if (dst >= cap) continue;  // the counter still advances: the host sees the shortfall and runs again with room
        const uint32_t i = a + jw, q = code[jw];
        // boundary bits i+1 .. i+64 (the next one is at most W <= 63 windows away)
        const uint32_t bw = (i + 1u) >> 5, bo = (i + 1u) & 31u;
        const uint64_t lo64 = ((uint64_t)sm_bound[bw + 1] << 32) | sm_bound[bw];
        const uint64_t bits = bo ? ((lo64 >> bo) | ((uint64_t)sm_bound[bw + 2] << (64u - bo))) : lo64;
        const uint32_t size = (uint32_t)__ffsll((long long)bits);  // >= 1
        uint64_t mm, hi;
        extract_kmer<false>(sm_words, q, m, mm, hi);
        out_mm[dst] = mm;
        out_val[dst] = RunVal{first_base + (t0 + i - shift), q - i, size};
    }
}

// after the sort: the builder's record, ready to be copied out
__global__ __launch_bounds__(256) void k_assemble_runs(const uint64_t* __restrict__ mm, const RunVal* __restrict__ val, size_t n, SuperKmerRun* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const RunVal v = val[i];
    out[i] = SuperKmerRun{mm[i], v.first, v.first + v.p1, v.size, 0u};
}

// ---- classification of the sorted records (src/minimizer.cpp:5-50) ----
// record i is the first of its minimizer / belongs to a minimizer that has several records
__device__ __forceinline__ void group_flags(const uint64_t* __restrict__ mm, size_t i, size_t n, bool& head, bool& multi) {
    const uint64_t x = mm[i];
    head = i == 0 || mm[i - 1] != x;
    multi = !head || (i + 1 < n && mm[i + 1] == x);
}
// flags[i] = head | multi << 32 (packed so that one scan counts both); *kmers += the sizes (one atomic per wave)
__global__ __launch_bounds__(256) void k_group_flags(const uint64_t* __restrict__ mm, const RunVal* __restrict__ val, size_t n, uint64_t* __restrict__ flags,
                                                       unsigned long long* __restrict__ kmers) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    unsigned long long sz = 0;
    if (i < n) {
        bool head, multi;
        group_flags(mm, i, n, head, multi);
        flags[i] = (uint64_t)head | ((uint64_t)multi << 32);
        sz = val[i].size;
    }
    for (unsigned d = 32; d; d >>= 1) sz += __shfl_down(sz, d, 64);
    if ((threadIdx.x & 63u) == 0 && sz) atomicAdd(kmers, sz);
}
// slots = exclusive scan of flags: low word = index among the distinct minimizers, high word = index among the colliding records
__global__ __launch_bounds__(256) void k_group_scatter(const uint64_t* __restrict__ mm, const RunVal* __restrict__ val, const uint64_t* __restrict__ slots, size_t n,
                                                         uint64_t* __restrict__ uniq_mm, uint16_t* __restrict__ uniq_offset_size, uint64_t* __restrict__ coll_first,
                                                         uint32_t* __restrict__ coll_size) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    bool head, multi;
    group_flags(mm, i, n, head, multi);
    const uint64_t slot = slots[i];
    const RunVal v = val[i];
    if (head) {
        const size_t u = (size_t)(slot & 0xFFFFFFFFull);
        uniq_mm[u] = mm[i];
        uniq_offset_size[u] = multi ? (uint16_t)0 : (uint16_t)(v.p1 | (v.size << 8));
    }
    if (multi) {
        const size_t c = (size_t)(slot >> 32);
        coll_first[c] = v.first;
        coll_size[c] = v.size;
    }
}

}  // namespace lphq
