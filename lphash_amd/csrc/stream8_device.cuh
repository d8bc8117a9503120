// stream8_device.cuh -- the streaming kernel for windows of W = k - m + 1 >= 8 positions: eight m-mer positions and windows per lane.
//
// Same tile (1024 m-mer positions, windows cut as stream_tile_windows says), same phases A-F and same results as k_query_tiles_stream
// (kernels_device.cuh; MPHF::operator()(.., streaming=true), include/partitioned_mphf.hpp:78-184), with 128 threads instead of 256: a lane owns
// the block of 8 consecutive positions a .. a+7, a = 8 * tid.  Why: at W = 44 the 4-per-lane kernel issues 602 VALU instructions per wave of
// which 96 are the hashes (profiles/r03_stream_phases_k63.txt) -- the rest is per-LANE and per-WAVE work (addressing, the block range every
// window of a lane shares, ballots, barriers) that this shape pays once per 8 windows instead of once per 4.  What it took (profiles/r03_stream8.txt):
//   C  W >= 16: window a+j = own suffix S[j] + the blocks all eight windows cover (l+1 .. l+nf0, nf0 = (W-8)/8: 4 reads at W = 44, against 10 per
//      four windows) + at most one more whole block + a prefix of the block after it.  W <= 15: from the raw hashes (core + two edge chains)
//   D  one cross-lane move and one LDS atomic per 8 windows; two waves meet at the barriers instead of four
//   F  the codes leave through LDS: a lane's eight codes are 64 B from its neighbour's, and stores from the registers were what made the first
//      version no faster than four per lane
//   LDS and registers for 14 workgroups per CU (10.7-11.0 KB, <= 71 VGPRs): the kernel's time is nearly proportional to 1 / (waves per CU)
// Ties keep the LEFT position throughout (candidates are taken left to right with a strict `<`), as the strict `>` comparisons of
// include/partitioned_mphf.hpp:122,150-163 do.
#pragma once
#include <gfx950/shift_in.cuh>
#include <gfx950/vgpr_allocation.cuh>
#include <gfx950/wave_scan.cuh>

#include "kernels_device.cuh"
#include "stream_sinks.h"

namespace lphq {

template <int SINK>
struct S8SinkArgs {
    typedef S8NoArgs type;
};
template <>
struct S8SinkArgs<kSinkRuns> {
    typedef S8RunsArgs type;
};
template <>
struct S8SinkArgs<kSinkGather> {
    typedef S8GatherArgs type;
};

constexpr int kS8Block = 128;                          // threads per workgroup; 8 * BLOCK m-mer positions per tile (256 threads / 2048 positions: measured equal at W = 8, 12,
                                                       // profiles/r03_stream8.txt (9); the template parameter stays for the emulator's coverage)
constexpr int kS8RecCap = 16;                          // records of a multi-record tile cached in LDS (more: the global offset arrays are searched)
// Blocks per row of prefix minima.  A lane reads up to block l + 8 (W <= 62), i.e. past its row's BLOCK entries -- but only for windows that begin at or
// beyond TW = stream_tile_windows(W, 8 * BLOCK): a window the tile emits ends at position <= 8 * BLOCK - 1, inside block BLOCK - 1.  What such a read
// returns is discarded with the window (code 0xFFFE), so the rows need no ~0 padding of their own: a read past row c lands in row c + 1 (the rows are
// one array; the odd row length keeps the eight columns of a block in different banks), and kS8Tail entries behind the last row keep the furthest
// one inside the array.  Round 4: 9 padded blocks per row -> 1, 448 B of LDS less per workgroup.
template <int BLOCK>
constexpr uint32_t s8_row() { return (uint32_t)BLOCK + 1u; }
constexpr uint32_t kS8Tail = 8u;
// slot of position i in phases E-F (the leader's base): the position itself (round 4; the column-major slot of phases B-C cost three instructions per
// window in phase F for nothing: neighbouring lanes read the same leader or one a few slots away)
template <int BLOCK>
__device__ __forceinline__ uint32_t hslot8(uint32_t i) { return i; }

// Publishes (column c of sm_hash) the prefix minima P[c] over a .. a+c -- P[7] is the minimum M of the block -- and, 3 bits each in one word,
// which position each of them is (P[0] is position 0).  (The suffix minima a window needs of its own block are formed in phase C, in one chain with the
// minimum of the blocks all eight windows share: window_argmin8.)
template <int BLOCK>
__device__ __forceinline__ void publish_block_minima8(uint64_t* __restrict__ sm_hash, uint32_t* __restrict__ sm_off, uint32_t tid, const uint64_t (&h)[8]) {
    constexpr uint32_t kS8Row = s8_row<BLOCK>();
    HashPos p{h[0], 0};
    sm_hash[tid] = p.h;
    uint32_t ob = 0;
#pragma unroll
    for (uint32_t c = 1; c < 8; ++c) {
        take_if_less(p, h[c], c);
        sm_hash[c * kS8Row + tid] = p.h;
        ob |= p.p << (3u * c);
    }
    sm_off[tid] = ob;
}

// Leftmost minimum of the 8 consecutive windows starting at local positions a .. a+7, each W >= 16 positions wide -> q[0..7] (local positions).
// Window a+j = own suffix S[j] (a+j .. a+7), the blocks every window covers (l+1 .. l+nf0, nf0 = (W - 8) / 8), then -- by (j + W - 8) / 8 and % 8, uniform
// over the wave -- block l+nf0+1 whole and a prefix of block l+nf0+2, or only a prefix of block l+nf0+1.  Every candidate carries its POSITION relative to
// a from the start (round 4): the own block's come out of the 3-bit fields of s.p / sm_off[tid]; the two blocks behind the common ones are at uniform
// distances, so their words of sm_off are read once per lane and a field of them is one v_bfe_u32 per candidate; only the winner of the common blocks
// needs a data-dependent read.  Before, every window carried a (block, column) id through its chain and decoded it afterwards with a dependent LDS read,
// a variable shift and a select: 8 decodes of ~11 instructions per lane, now 1.
// W8 = (W - 8) % 8 as a compile-time constant (the launcher has one instantiation per value), or -1: taken from W at run time.  Which of the eight windows
// take block l+nf0+1 whole, and how long a prefix of the block behind it each of them takes, depends on W only through that remainder; with it fixed at
// compile time every per-window decision below is resolved by the compiler: no scalar branches, LDS offsets as immediates (measured with W fixed
// altogether: -4.8 % VALU, -32 % SALU instructions, -2.5 % kernel time at C4, profiles/r04_stream8.txt).
template <int BLOCK, int W8>
__device__ __forceinline__ void window_argmin8(const uint64_t* __restrict__ sm_hash, const uint32_t* __restrict__ sm_off, uint32_t tid, uint32_t W, const uint64_t (&h)[8],
                                               uint32_t (&q)[8]) {
    constexpr uint32_t kS8Row = s8_row<BLOCK>();
    const uint64_t* colM = sm_hash + 7u * kS8Row + tid;  // M of the block t lanes to the right: colM[t]
    const uint32_t nf0 = (W - 8u) >> 3;                  // blocks covered completely by all eight windows: l+1 .. l+nf0 (>= 1: W >= 16)
    const uint32_t w8 = W8 >= 0 ? (uint32_t)W8 : ((W - 8u) & 7u);
    HashPos r{colM[1], 1u};
    for (uint32_t t = 2; t <= nf0; ++t) take_if_less(r, colM[t], t);
    r.p = (r.p << 3) + ((sm_off[tid + r.p] >> 21) & 7u);  // position relative to a: 8 t + where block t's minimum sits
    const uint32_t wA = sm_off[tid + nf0 + 1u], wB = sm_off[tid + nf0 + 2u];
    const uint32_t relA = (nf0 + 1u) << 3, relB = relA + 8u;  // (uniform)
    const HashPos mA{colM[nf0 + 1u], relA + ((wA >> 21) & 7u)};
    const uint32_t a = 8u * tid;
    // Window a+j needs the leftmost minimum of its own positions j .. 7 AND of the common blocks: ONE chain from the right -- start at the common blocks'
    // minimum, take position 7, 6, ... with `<=` (scanning leftwards, the left one wins ties) -- yields both for all eight windows in 8 compare-selects,
    // where a suffix chain over the own block (6) plus one comparison with the common minimum per window (8) took 14 (round 4, last step).
    HashPos cur = r;
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        const bool le = h[j] <= cur.h;
        cur.h = le ? h[j] : cur.h;
        cur.p = le ? (uint32_t)j : cur.p;
        HashPos w = cur;
        const bool second = (uint32_t)j + w8 >= 8u;   // (uniform; a constant when W8 is) the window takes block l+nf0+1 whole and a prefix of block l+nf0+2
        const uint32_t hc = ((uint32_t)j + w8) & 7u;  // ... of that many positions
        if (second) take_if_less(w, mA.h, mA.p);
        if (hc) {
            const uint32_t word = second ? wB : wA, rel = second ? relB : relA;
            take_if_less(w, sm_hash[(hc - 1u) * kS8Row + tid + nf0 + (second ? 2u : 1u)], rel + ((word >> (3u * (hc - 1u))) & 7u));
        }
        q[j] = a + w.p;
    }
}

// The same for NARROW windows (W <= 15, i.e. P == 4) from the raw hashes (column c of sm_hash holds the hash of position 8 * block + c): a window spans at
// most three blocks, so publishing prefix minima and decoding the winner's place would cost more than they save.  W >= 8: the positions a+7 .. a+W-1 are
// common to the eight windows (the core); window a+j = the left suffix minimum over a+j .. a+6, the core, the right prefix minimum over a+W .. a+W+j-1 --
// W + 7 LDS reads and W + 18 compare-selects per lane.  Ties keep the LEFT position (candidates left to right with a strict `<`; the suffix chain runs right
// to left with `<=`).
template <int BLOCK>
__device__ __forceinline__ void window_argmin8_raw(const uint64_t* __restrict__ sm_hash, uint32_t tid, uint32_t W, uint32_t (&q)[8]) {
    constexpr uint32_t kS8Row = s8_row<BLOCK>();
    auto H = [&](uint32_t c) { return sm_hash[(c & 7u) * kS8Row + tid + (c >> 3)]; };  // hash of position a + c
    const uint32_t a = 8u * tid;
    if (W >= 8u) {
        uint64_t Lh[7];   // Lh[j]: leftmost minimum over offsets j .. 6; where it is: 3 bits each in Lp (one register instead of seven: this kernel's
        uint32_t Lp;      // waves per CU are bounded by its VGPRs as much as by its LDS)
        {
            uint64_t cur = H(6);
            uint32_t cp = 6;
            Lh[6] = cur;
            Lp = cp << 18;
#pragma unroll
            for (int j = 5; j >= 0; --j) {
                const uint64_t h = H((uint32_t)j);
                const bool le = h <= cur;  // `<=`: scanning from the right, the left one wins ties
                cur = le ? h : cur;
                cp = le ? (uint32_t)j : cp;
                Lh[j] = cur;
                Lp |= cp << (3u * (uint32_t)j);
            }
        }
        HashPos c{H(7), 7};
        for (uint32_t o = 8; o < W; ++o) take_if_less(c, H(o), o);
        HashPos r{~0ull, 0};  // right prefix minimum over offsets W .. W + j - 1
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            HashPos w = c;
            if (j < 7u) {
                w = HashPos{Lh[j < 7u ? j : 0], (Lp >> (3u * j)) & 7u};
                take_if_less(w, c.h, c.p);
            }
            if (j > 0u) {
                take_if_less(r, H(W + j - 1u), W + j - 1u);
                take_if_less(w, r.h, r.p);
            }
            q[j] = a + w.p;
        }
    } else {
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            HashPos c{H(j), j};
            for (uint32_t t = 1; t < W; ++t) take_if_less(c, H(j + t), j + t);
            q[j] = a + c.p;
        }
    }
}

// DIRECT: window minima from the raw hashes (narrow windows) instead of block prefix minima
template <int KIND, bool K128, int P, int BLOCK = kS8Block, bool DIRECT = (P == 4), int W8 = -1, int SINK = kSinkCodes>
// (second launch bound = waves per SIMD the register allocation must leave room for: 8 = 64 VGPRs for the wide-window variant, whose LDS now allows the
// 16 workgroups per CU that takes; the narrow-window variant stays at its 71)
__global__ __launch_bounds__(BLOCK, DIRECT ? 1 : 8) void k_query_tiles_stream8(const uint8_t* __restrict__ img, const uint8_t* __restrict__ seq_al, uint64_t shift, uint64_t total,
                                                                  const uint64_t* __restrict__ seq_off, const uint64_t* __restrict__ out_off, uint32_t n_records,
                                                                  uint64_t* __restrict__ out, uint64_t tile_base /* first tile of this launch */,
                                                                  const typename S8SinkArgs<SINK>::type sink) {
    if constexpr (SINK == kSinkGather) {
        // The payload sink's wide-window instantiations need exactly 56 VGPRs, and with a 56-register allocation the first of the eight fallback blocks of phase F --
        // the only code that holds values in v54 / v55 -- came out wrong for ~1 colliding window in 100 once the device was full (never on its first workgroups,
        // never in the emulator; same instructions with a 64-register allocation: never).  What writes there was not found (profiles/r06_transposed_gather_attempt.txt:
        // not a missing wait, not an LDS race, registers held by a probe kernel beside it keep their values); the allocation is what decides, so it is made 64 --
        // which costs nothing: 10 KB of LDS per workgroup already bounds a SIMD at 8 waves.  tests/test_gpu_runs.py::test_fallback_rich_batch_under_load is the guard.
        allocate_64_vgprs();  // (gfx950/vgpr_allocation.cuh)
    }
    const lphq_image_header& H = *(const lphq_image_header*)img;
    constexpr int kS8Pos = 8 * BLOCK, NW = tile_words(kS8Pos);
    constexpr uint32_t kS8Row = s8_row<BLOCK>();
    constexpr int NDW = (NW + 1) * 4;          // dwords of sequence (4 bases each) the tile loads: two per lane, the NDW - 2 * BLOCK beyond by lanes of the last wave
    constexpr uint32_t NX = (uint32_t)(NDW - 2 * BLOCK);
    static_assert(NDW > 2 * BLOCK && NX <= 64, "the dwords beyond the first 2 * BLOCK are loaded by the lanes of the last wave");
    __shared__ uint64_t sm_hash[8 * kS8Row + kS8Tail];  // phase B-C: prefix minima of every block (column c, block b at c * kS8Row + b); phase E-F: the leader's base | mode << 62
    __shared__ uint32_t sm_words[NW + 8];
    __shared__ uint8_t sm_badb[NDW + 12];      // bad-base flags, one byte per sequence dword (written by the lane that packs the dword)
    __shared__ uint32_t sm_anybad[BLOCK / 64];
    // LDS is what limits the waves per CU of this kernel, and waves are what hide its latencies (measured: 13 -> 15 workgroups per CU = +7 %,
    // profiles/r03_stream8.txt): the leader queue holds QCAP leaders (a tile with more is redone slot by slot), the leader's mode rides in
    // the top bits of its base, the record table holds kS8RecCap records -- 10.7 KB per workgroup (11.0 with the 256-entry queue of the narrow-window
    // variant): LDS for 14-15 workgroups per CU; the 71 VGPRs allow 14 (28 waves).
    constexpr uint32_t QCAP = DIRECT ? 2u * BLOCK : BLOCK;  // leaders per round: narrow windows have more super-k-mers per tile (W = 12: ~157 of 1024 positions)
    __shared__ uint16_t sm_queue[QCAP + 8];  // (8 of slack: a lane whose first leader fits writes all of its leaders)
    __shared__ uint32_t sm_off[kS8Row + kS8Tail];  // phase B-C: where in its block each published prefix minimum sits (3 bits each); read up to 8 blocks past a lane's own
    __shared__ uint32_t sm_rec[2], sm_count, sm_slot_count;  // (a counter of its own for the slot-by-slot path: lanes may still be reading sm_count)
    __shared__ uint64_t sm_roff[3];                   // seq_off[rlo], seq_off[rlo+1], out_off[rlo]
    __shared__ uint64_t sm_rtab[2 * kS8RecCap + 1];   // multi-record tiles: the tile's slice of seq_off / out_off
    // kSinkRuns (unused, hence absent, in the other sinks):
    static_assert(SINK != kSinkRuns || kS8Pos <= 1024, "a queue entry of the runs sink holds the window in 10 bits and the minimizer's offset in 6 (W <= 63)");
    __shared__ uint64_t sm_break[BLOCK / 8];          // bit w % 64 of word w / 64: window w ends the run to its left (it leads one itself, or it is no window); lane l writes byte l
    __shared__ uint64_t sm_tile_at, sm_xbase;         // first entry of the tile's reservation in its arena; first explicit code of the round's reservation
    __shared__ uint32_t sm_xcount, sm_xcursor;        // explicit codes the round's colliding runs need; the cursor they are dealt from
    const uint32_t tid = threadIdx.x;
    const uint32_t k = H.k, m = H.m, W = H.W;
    const uint32_t TW = stream_tile_windows(W, kS8Pos);
    const uint64_t t0 = (tile_base + blockIdx.x) * TW;  // shifted coordinate of the tile's first window start

    // ---- A: sequence dwords first (no dependence on the record search), then the search, then pack into LDS ----
    const bool interior = t0 >= shift && t0 + 4ull * (uint64_t)NDW <= shift + total;  // (uniform) every dword the tile loads lies inside the batch
    const uint32_t dd[3] = {2u * tid, 2u * tid + 1u, 2u * (uint32_t)BLOCK + (tid - ((uint32_t)BLOCK - 64u))};
    const bool live2 = tid >= (uint32_t)BLOCK - 64u && tid - ((uint32_t)BLOCK - 64u) < NX;
    uint32_t xs[3] = {0u, 0u, 0u};
    bool have[3];
    if (interior) {
        const uint2 v = *(const uint2*)(seq_al + t0 + 8ull * tid);  // (t0 and seq_al are 16-byte aligned)
        xs[0] = v.x;
        xs[1] = v.y;
        have[0] = have[1] = true;
        have[2] = live2;
        if (live2) xs[2] = *(const uint32_t*)(seq_al + t0 + 4ull * dd[2]);
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint64_t a = t0 + 4ull * dd[r];
            have[r] = (r < 2 || live2) && a < shift + total && a + 4u > shift;
            if (have[r]) xs[r] = *(const uint32_t*)(seq_al + a);
        }
    }
    if (tid < 64) {  // wave 0: record of the first base the tile can touch, and of the last
        const uint64_t gfirst = t0 > shift ? t0 - shift : 0;
        uint64_t glast = t0 + TW + 64 > shift ? t0 + TW + 64 - shift : 0;
        if (glast >= total) glast = total ? total - 1 : 0;
        const uint32_t r = find_record_wave(seq_off, n_records, gfirst);
        const uint64_t s0 = seq_off[r], e0 = seq_off[r + 1];
        const uint32_t r2 = glast < e0 ? r : find_record_wave(seq_off, n_records, glast);  // uniform over the wave
        if (tid == 0) {
            sm_rec[0] = r;
            sm_rec[1] = r2;
            sm_roff[0] = s0;
            sm_roff[1] = e0;
            sm_roff[2] = out_off[r];
            sm_count = 0;
            if (SINK == kSinkRuns) sm_xcount = 0;
        }
    }
    if (tid < 12u) sm_badb[NDW + tid] = 0xFu;  // beyond the tile's dwords: not bases
    for (int c = NW + 1 + tid; c < NW + 8; c += BLOCK) sm_words[c] = 0;
    bool real_bad = false;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (r < 2 || live2) {
            const uint32_t d = dd[r];
            const uint64_t a = t0 + 4ull * d;
            uint32_t p8 = 0, b4 = 0xFu;
            if (have[r]) {
                p8 = pack4(xs[r], b4);
                uint32_t outside = 0;  // bytes of the dword that are not bases of the batch
                if (!interior && (a < shift || a + 4u > shift + total))
                    for (int i = 0; i < 4; ++i)
                        if (a + i < shift || a + i >= shift + total) outside |= 1u << i;
                real_bad = real_bad || (b4 & ~outside);  // a real non-ACGT base
                b4 |= outside;
            }
            ((uint8_t*)sm_words)[(d & ~3u) + (3u - (d & 3u))] = (uint8_t)p8;  // first base in the top bits of each 32-bit word
            sm_badb[d] = (uint8_t)b4;
        }
    }
    {
        const bool wave_bad = __any(real_bad);
        if ((tid & 63u) == 0u) sm_anybad[tid >> 6] = wave_bad ? 1u : 0u;
    }
    __syncthreads();
    if (sm_rec[0] != sm_rec[1]) load_record_table<kS8RecCap>(sm_rtab, seq_off, out_off, sm_rec[0], sm_rec[1], tid);  // (visible to phase C through the next barrier)
    static_assert(kS8RecCap + 1 <= BLOCK, "load_record_table: one lane per table entry");

    // ---- B: hashes of the m-mers starting at local positions 8*lane .. 8*lane+7; prefix minima of the block published for phase C ----
    const uint32_t a = 8u * tid;
    uint64_t h[8];  // the lane's eight hashes: the wide-window variant uses them again in phase C (its own block's part of every window)
    {
        const uint64_t mm_seedx = murmur_seedx(H.mm_seed);
        const uint32_t w0 = a >> 4, o = (a & 15u) * 2u;  // o = 0 or 16
        const uint32_t x0 = sm_words[w0], x1 = sm_words[w0 + 1], x2 = sm_words[w0 + 2], x3 = sm_words[w0 + 3];
        const uint64_t A64 = ((uint64_t)x0 << 32) | x1, C64 = ((uint64_t)x2 << 32) | x3;
        const uint64_t s0 = o ? ((A64 << o) | ((uint64_t)x2 >> (32 - o))) : A64;  // bits [0, 64) of the stream at base a
        const uint64_t s1 = o ? (C64 << o) : C64;                                  // bits [64, 128): the top 14 are needed at most
        const uint32_t sh = 64u - 2u * m;
        const uint64_t mask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
        if (m <= 25u) {  // (uniform) the eight m-mers end inside the first 64 bits: one shift each
#pragma unroll
            for (uint32_t r = 0; r < 8; ++r) h[r] = murmur8((s0 >> (sh - 2 * r)) & mask, mm_seedx);
        } else {
#pragma unroll
            for (uint32_t r = 0; r < 8; ++r) {
                const uint64_t v = r == 0 ? (s0 >> sh) : (((s0 << (2 * r)) | (s1 >> (64 - 2 * r))) >> sh);
                h[r] = murmur8(v & mask, mm_seedx);
            }
        }
        if (DIRECT) {
#pragma unroll
            for (uint32_t r = 0; r < 8; ++r) sm_hash[r * kS8Row + tid] = h[r];
        } else {
            publish_block_minima8<BLOCK>(sm_hash, sm_off, tid, h);
        }
        // (blocks beyond the tile are read by windows the tile does not emit only: see s8_row)
    }
    __syncthreads();

    // ---- C + D: leftmost minimum of the lane's 8 windows, their validity; leaders -> queue (publish_leaders below) ----
    const uint32_t rlo = sm_rec[0], rhi = sm_rec[1];
    const bool one_record = rlo == rhi;
    bool anybad = false;
#pragma unroll
    for (int wv = 0; wv < BLOCK / 64; ++wv) anybad = anybad || sm_anybad[wv] != 0u;
    // one record: window i of the tile is a window of the record iff w_lo <= i < w_hi
    const uint32_t w_lo = t0 >= shift ? 0u : (uint32_t)(shift - t0);
    uint32_t w_hi = 0;
    {
        const uint64_t limit = sm_roff[1] + shift;  // shifted coordinate of the record's end
        if (limit >= t0 + k) w_hi = (uint32_t)min((uint64_t)TW, limit - k - t0 + 1u);
    }
    // (uniform) a PLAIN tile: inside one record from its first window to its last (w_lo = 0, w_hi = TW) and without a non-ACGT base -- all tiles of a long
    // record but its first and last.  Its windows are valid or not by whole lanes (TW is a multiple of 16), so what the general path decides per window
    // (round 3: ~55 instructions per lane in phases C, D and F) is one compare per lane here.
    const bool plain = one_record && !anybad && w_lo == 0u && w_hi == TW;
    uint32_t code[8];  // minimizer position of the window; 0xFFFF: holds a non-ACGT base; 0xFFFE: not a window of any record
    uint32_t jrec = 0, jcur = 0;  // multi-record tiles: record (relative to rlo) of the lane's first in-batch window, and a cursor
    {
        uint32_t q[8];
        if (DIRECT) window_argmin8_raw<BLOCK>(sm_hash, tid, W, q);
        else window_argmin8<BLOCK, W8>(sm_hash, sm_off, tid, W, h, q);
        if (plain) {  // every window below TW (a multiple of 8: whole lanes) is a window of the record
            const bool lane_in = a < TW;
#pragma unroll
            for (uint32_t jw = 0; jw < 8; ++jw) code[jw] = lane_in ? q[jw] : 0xFFFEu;
        } else if (one_record) {
#pragma unroll
            for (uint32_t jw = 0; jw < 8; ++jw) code[jw] = (a + jw >= w_lo && a + jw < w_hi) ? q[jw] : 0xFFFEu;
        } else {
            bool found = false;
            uint64_t rs = 0, re = 0;
#pragma unroll
            for (uint32_t jw = 0; jw < 8; ++jw) {
                const uint32_t i = a + jw;
                const uint64_t ga = t0 + i;
                uint32_t v = 0xFFFEu;
                if (i < TW && ga >= shift && ga - shift + k <= total) {
                    const uint64_t g = ga - shift;
                    if (!found) {  // one search per lane; its record index is kept for phase F
                        jrec = tile_record<kS8RecCap>(sm_rtab, seq_off, rlo, rhi, g, rs, re);
                        jcur = jrec;
                        found = true;
                    }
                    rec_walk(sm_rtab, seq_off, rlo, rhi - rlo < (uint32_t)kS8RecCap, g, jcur, rs, re);
                    if (g + k <= re) v = q[jw];
                }
                code[jw] = v;
            }
        }
        if (anybad) {  // uniform over the workgroup
            const uint64_t kmask = (1ull << k) - 1ull;  // k <= 63
            // bad-base bits [a, a + 72): windows a..a+7 need bits up to a + 7 + 62; one nibble per sequence dword, the lane's first dword is 2 * tid
            uint64_t blo = 0;
#pragma unroll
            for (uint32_t d = 0; d < 16; ++d) blo |= (uint64_t)(sm_badb[2u * tid + d] & 0xFu) << (4 * d);
            const uint32_t bhi = (sm_badb[2u * tid + 16u] & 0xFu) | ((uint32_t)(sm_badb[2u * tid + 17u] & 0xFu) << 4);
#pragma unroll
            for (uint32_t jw = 0; jw < 8; ++jw) {
                const uint64_t x = jw ? ((blo >> jw) | ((uint64_t)bhi << (64 - jw))) : blo;
                if (code[jw] != 0xFFFEu && (x & kmask)) code[jw] = 0xFFFFu;
            }
        }
    }
    // D: windows whose minimizer position differs from their left neighbour's lead a super-k-mer.  Leader number n of the tile (numbered wave by wave,
    // window slot by window slot, lane by lane -- any order will do, lookups are independent) goes to queue entry n - r0 of the round that starts at r0.
    // (the first QCAP leaders go to the queue; a tile with more is redone slot by slot below)
    uint32_t lmask = 0;  // bit jw: the lane's window jw leads a super-k-mer
    {
        const uint32_t lane = tid & 63u;
        uint32_t prev = __shfl(code[7], (int)((lane + 63u) & 63u), 64);  // last window of the lane to the left
        if (lane == 0) prev = 0xFFFEu;                                     // it lives in another wave: queue the window regardless
        if (plain) {  // whole lanes are windows of the record or not: the validity of a leader is the lane's (gfx950/shift_in.cuh: 2 instructions per window)
#pragma unroll
            for (int jw = 7; jw >= 0; --jw) shift_in_ne(lmask, code[jw], jw ? code[jw - 1] : prev);
            lmask = a < TW ? lmask : 0u;
        } else {
#pragma unroll
            for (uint32_t jw = 0; jw < 8; ++jw) {
                const bool ld = code[jw] < 0xFFFEu && code[jw] != (jw ? code[jw - 1] : prev);
                lmask |= ld ? 1u << jw : 0u;
            }
        }
        // Round 4: the lane's leaders are numbered by ONE prefix sum of popcount(lmask) over the wave (six DPP adds, gfx950/wave_scan.cuh) and written
        // one after the other -- leaders in lane order instead of slot order, which the lookups do not care about.  Round 3 took a ballot, a population
        // count and an mbcnt pair per window slot, twice.
        const uint32_t mine = (uint32_t)__popc(lmask);
        const uint32_t incl = wave_inclusive_add(mine);
        const uint32_t wave_leaders = __shfl(incl, 63, 64);
        uint32_t b = 0;
        if (lane == 0 && wave_leaders) b = atomicAdd(&sm_count, wave_leaders);
        uint32_t at = __shfl(b, 0, 64) + incl - mine;  // queue entry of the lane's first leader
        if (at < QCAP) {  // (a lane writes at most 8 entries from there: the queue has 8 entries of slack; a tile with more than QCAP leaders is redone below)
#pragma unroll
            for (uint32_t jw = 0; jw < 8; ++jw) {
                // (the runs sink also needs to know WHICH window leads: its entry is the window and the minimizer's offset in it, 10 + 6 bits)
                // (window | offset << 10 = minimizer position * 1024 - window * 1023: no bits overlap)
                if ((lmask >> jw) & 1u) sm_queue[at] = SINK == kSinkRuns ? (uint16_t)((code[jw] << 10) - (a + jw) * 1023u) : (uint16_t)code[jw];
                at += (lmask >> jw) & 1u;
            }
        }
        if constexpr (SINK == kSinkRuns) {  // where runs end: at every window that leads one or does not exist (non-ACGT base, beyond the record or the tile)
            uint32_t vmask = 0;
            if (plain) vmask = a < TW ? 0xFFu : 0u;
            else {
#pragma unroll
                for (uint32_t jw = 0; jw < 8; ++jw) vmask |= code[jw] < 0xFFFEu ? 1u << jw : 0u;
            }
            ((uint8_t*)sm_break)[tid] = (uint8_t)((lmask | ~vmask) & 0xFFu);
        }
    }
    __syncthreads();

    // ---- E: one lookup per super-k-mer, dense lanes; a round = up to BLOCK leaders, one per lane ----
    const uint32_t nlead = sm_count;
    if constexpr (SINK == kSinkRuns) {
        // one reservation per tile, asked for here and used after the lookups: the atomic's round trip hides behind them.  (One counter per arena, not one
        // for the batch: atomics on one address serialise at ~80 M/s on this chip, tools/gather_ceiling.hip.)
        if (tid == 0) sm_tile_at = nlead ? atomicAdd((unsigned long long*)&sink.counts[(tile_base + blockIdx.x) & (LPHQ_RUN_ARENAS - 1u)].runs, (unsigned long long)nlead) : 0ull;
    }
    auto lookup_queued = [&](uint32_t n) {  // the first n queue entries (n uniform)
        for (uint32_t x = tid; x < n; x += (uint32_t)BLOCK) {
            const uint32_t e = sm_queue[x];
            const uint32_t q = SINK == kSinkRuns ? (e & 1023u) + (e >> 10) : e;
            uint64_t mm, hi;
            extract_kmer<false>(sm_words, q, m, mm, hi);
            const LookupCtx c = lookup_ctx<KIND, K128, P>(img, H, mm);
            sm_hash[hslot8<BLOCK>(q)] = c.base | ((uint64_t)c.mode << LPHQ_TAIL_MODE_SHIFT);  // (bases are ranks < 2^62: the packing of the image's tail table)
        }
    };
    // kSinkRuns: one lphq_run per queue entry, written by the lanes that did the lookups -- DENSE lanes, one leader each (written from the windows' own lanes,
    // phase F, a run cost every wave eight divergent copies of this: +26 lane-instructions per window, profiles/r06_pmc_c4sruns_raw.txt of the first version).
    // A run = the windows from its leader to the next window that leads or does not exist, or to the end of the wave: inside it the minimizer stays where it is, so
    // the code moves by -1 (mode 0: base + p, and p shrinks as the window slides) or +1 (mode 1) per window -- the case-0 updates of
    // include/partitioned_mphf.hpp:131-145.  Colliding leaders (mode 2) only count their windows in the first pass; the second pass, entered when the tile has
    // one, gives them their place in the arena's explicit-code array and looks their k-mers up one by one (src/partitioned_mphf.cpp:312-316).
    const uint64_t first0 = sm_roff[2] + (t0 - shift - sm_roff[0]);  // index of window 0 of the tile in the code space (one record under the tile; wraps harmlessly before it)
    auto run_length = [&](uint32_t i) {  // windows from i to the next set bit of sm_break, inside the wave's 512 windows (eight 64-bit words: a run rarely leaves its word)
        const uint32_t end = (i | 511u) + 1u;
        uint32_t w = i + 1u, next = end;
        if (w < end) {
            uint64_t bits = sm_break[w >> 6] >> (w & 63u);
            if (bits) next = w + (uint32_t)__ffsll((long long)bits) - 1u;
            else
                for (w = (w | 63u) + 1u; w < end; w += 64u) {
                    bits = sm_break[w >> 6];
                    if (bits) {
                        next = w + (uint32_t)__ffsll((long long)bits) - 1u;
                        break;
                    }
                }
        }
        return next - i;
    };
    auto run_first = [&](uint32_t i) {  // index of window i of the tile in the code space
        uint64_t first = first0 + i;
        if (!one_record) {
            const uint64_t g = t0 + i - shift;
            uint64_t rs, re;
            const uint32_t j = tile_record<kS8RecCap>(sm_rtab, seq_off, rlo, rhi, g, rs, re);
            first = tile_record_out<kS8RecCap>(sm_rtab, out_off, rlo, rhi, j) + (g - rs);
        }
        return first;
    };
    auto emit_queued = [&](uint32_t n, uint32_t round_base, bool second) {
        if constexpr (SINK == kSinkRuns) {
            const uint32_t arena = (uint32_t)((tile_base + blockIdx.x) & (uint64_t)(LPHQ_RUN_ARENAS - 1u)), lane = tid & 63u;
            lphq_run* const rdst = sink.runs + (uint64_t)arena * sink.arena_runs;
            for (uint32_t x0 = 0; x0 < n; x0 += (uint32_t)BLOCK) {  // (uniform: the second pass works wave by wave)
                const uint32_t x = x0 + tid;
                uint32_t i = 0, p = 0, mode = 0, len = 0;
                uint64_t base = 0;
                if (x < n) {
                    const uint32_t e = sm_queue[x];
                    i = e & 1023u;
                    p = e >> 10;
                    const uint64_t packed = sm_hash[hslot8<BLOCK>(i + p)];
                    mode = (uint32_t)(packed >> LPHQ_TAIL_MODE_SHIFT);
                    base = packed & LPHQ_TAIL_BASE_MASK;
                    if (!second || mode == 2u) len = run_length(i);  // (the second pass is for the explicit runs only)
                }
                const uint64_t slot = sm_tile_at + round_base + x;
                if (!second) {  // affine runs are written; explicit runs count the codes they will need
                    if (len && mode != 2u) {
                        lphq_run r;
                        r.code = mode == 0u ? base + p : base - p;
                        r.where = (run_first(i) & 0xFFFFFFFFFFull) | ((uint64_t)len << 40) | ((uint64_t)(mode ^ 1u) << 62);
                        if (slot < sink.arena_runs) rdst[slot] = r;
                    } else if (len) {
                        atomicAdd(&sm_xcount, len);
                    }
                    continue;
                }
                // second pass: the wave's explicit runs take consecutive places in the round's reservation (a prefix sum over the lanes), and their windows -- all
                // of them together -- are dealt to the wave's lanes 64 at a time: a lane finds the run its item belongs to by a binary search over the lanes'
                // exclusive sums (six ds_bpermute) and looks its k-mer up in the fallback function; dense lanes whatever the runs' lengths
                const uint32_t incl = wave_inclusive_add(len), total = __shfl(incl, 63, 64);
                if (total == 0u) continue;  // (uniform over the wave)
                uint32_t wb = 0;
                if (lane == 0u) wb = atomicAdd(&sm_xcursor, total);
                const uint64_t wbase = sm_xbase + __shfl(wb, 0, 64);
                const uint32_t excl = incl - len;
                if (len) {
                    lphq_run r;
                    r.code = wbase + excl;
                    r.where = (run_first(i) & 0xFFFFFFFFFFull) | ((uint64_t)len << 40) | ((uint64_t)LPHQ_RUN_EXPLICIT << 62);
                    if (slot < sink.arena_runs) rdst[slot] = r;
                }
                uint64_t* const xdst = sink.codes + (uint64_t)arena * sink.arena_codes;
                LookupCtx c;
                c.base = H.collision_base;
                c.mode = 2u;
                for (uint32_t e0 = 0; e0 < total; e0 += 64u) {
                    const uint32_t e = e0 + lane;
                    uint32_t lo = 0, hi = 63u;  // the last lane whose exclusive sum is <= e owns item e (lanes without an explicit run repeat their neighbour's sum)
#pragma unroll
                    for (int step = 0; step < 6; ++step) {
                        const uint32_t mid = (lo + hi + 1u) >> 1;
                        const uint32_t v = __shfl(excl, (int)mid, 64);
                        if (v <= e) lo = mid;
                        else hi = mid - 1u;
                    }
                    const uint32_t wi = __shfl(i, (int)lo, 64) + (e - __shfl(excl, (int)lo, 64));
                    if (e < total) {
                        uint64_t klo = 0, khi = 0;
                        extract_kmer<K128>(sm_words, wi, k, klo, khi);
                        const uint64_t v = finish_hval<K128>(img, H, c, klo, khi, 0u);  // src/partitioned_mphf.cpp:312-316
                        if (wbase + e < sink.arena_codes) xdst[wbase + e] = v;
                    }
                }
            }
        }
    };
    // (uniform) the common case: one round over the queue as phase D filled it (super-k-mers average (W + 1) / 2 windows).  More leaders than the queue
    // holds: the tile is done window slot by window slot -- a lane leads at most once per slot, so the BLOCK leaders of a slot always fit -- eight rounds of
    // three barriers.  ONE copy of the lookup serves both (round 3 had two, and the rare one's registers were what the compiler spilled).
    const bool overflow = nlead > QCAP;
    const uint32_t rounds = overflow ? 8u : 1u;
    uint32_t emitted = 0;  // kSinkRuns: runs of the rounds before (a run's slot = the tile's reservation + emitted + its queue entry)
#pragma unroll 1
    for (uint32_t jw = 0; jw < rounds; ++jw) {
        uint32_t n = nlead;
        if (overflow) {
            if (tid == 0) sm_slot_count = 0;
            __syncthreads();
            // window slot jw's code is code[0] of an array rotated by one per round (eight rounds put it back as it was: phase F reads it again)
            const uint32_t cj = code[0];
#pragma unroll
            for (uint32_t t = 0; t < 7; ++t) code[t] = code[t + 1];
            code[7] = cj;
            const bool ld = (lmask >> jw) & 1u;
            const uint64_t bal = __ballot(ld);
            uint32_t b = 0;
            if ((tid & 63u) == 0 && bal) b = atomicAdd(&sm_slot_count, (uint32_t)__popcll(bal));
            b = __shfl(b, 0, 64);
            if (ld)
                sm_queue[b + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] =
                    SINK == kSinkRuns ? (uint16_t)((cj << 10) - (a + jw) * 1023u) : (uint16_t)cj;
            __syncthreads();
            n = sm_slot_count;
        }
        lookup_queued(n);
        __syncthreads();
        if constexpr (SINK == kSinkRuns) {
            emit_queued(n, emitted, false);
            __syncthreads();
            if (sm_xcount != 0u) {  // (uniform) some run of the round is explicit -- its leader counted its windows: reserve their codes, then the second pass
                if (tid == 0) {
                    sm_xbase = atomicAdd((unsigned long long*)&sink.counts[(tile_base + blockIdx.x) & (LPHQ_RUN_ARENAS - 1u)].codes, (unsigned long long)sm_xcount);
                    sm_xcursor = 0;
                }
                __syncthreads();
                emit_queued(n, emitted, true);
                __syncthreads();
                if (tid == 0) sm_xcount = 0;  // (for the next round; ordered by its first barrier)
            }
            emitted += n;
        }
    }
    if constexpr (SINK == kSinkRuns) return;  // the runs sink has no phase F: everything it writes comes from the leaders' lanes

    // ---- F: the lane's eight codes ----
    uint64_t hv[8];
    uint32_t modes = 0;  // OR of the mode fields of the lane's windows (bit 1: a colliding minimizer)
    uint32_t coll = 0;   // which of the lane's windows sit under a colliding minimizer (worked out off the common path)
    if (plain) {
        // the codes of lanes at or beyond TW are never stored (the stores below are bounded by w_hi = TW): nothing to invalidate per window
#pragma unroll
        for (uint32_t jw = 0; jw < 8; ++jw) {
            const uint32_t qq = code[jw] & (uint32_t)(kS8Pos - 1);
            const uint64_t packed = sm_hash[hslot8<BLOCK>(qq)];
            const uint64_t base = packed & LPHQ_TAIL_BASE_MASK;
            const uint32_t mode = (uint32_t)(packed >> LPHQ_TAIL_MODE_SHIFT);
            const uint32_t p = qq - (a + jw);
            const int32_t delta = mode == 0u ? (int32_t)p : -(int32_t)p;
            hv[jw] = base + (uint64_t)(int64_t)delta;
            modes |= mode;
        }
        modes = a < TW ? modes : 0u;  // (what a lane beyond TW read is whatever phases B-C left in slot 1022)
    } else {
#pragma unroll
        for (uint32_t jw = 0; jw < 8; ++jw) {
            const uint32_t q = code[jw];
            const bool valid = q < 0xFFFEu;
            const uint32_t qq = q & (uint32_t)(kS8Pos - 1);  // (an invalid window reads some slot of the tile: the value is discarded)
            const uint64_t packed = sm_hash[hslot8<BLOCK>(qq)];
            const uint64_t base = packed & LPHQ_TAIL_BASE_MASK;
            const uint32_t mode = (uint32_t)(packed >> LPHQ_TAIL_MODE_SHIFT);
            const uint32_t p = qq - (a + jw);
            const int32_t delta = mode == 0u ? (int32_t)p : -(int32_t)p;  // |p| < 2^16; (mode 2 is redone below)
            const uint64_t hval = base + (uint64_t)(int64_t)delta;
            hv[jw] = valid ? hval : ~0ull;  // LPHQ_INVALID
            modes |= valid ? mode : 0u;  // (what an invalid window read is whatever phases B-C left in that slot)
        }
    }
    if (__any((modes & 2u) != 0u)) {  // windows under a colliding minimizer: the fallback function of the k-mer (src/partitioned_mphf.cpp:312-316)
#pragma unroll
        for (uint32_t jw = 0; jw < 8; ++jw)
            if (code[jw] < 0xFFFEu && (uint32_t)(sm_hash[hslot8<BLOCK>(code[jw])] >> LPHQ_TAIL_MODE_SHIFT) == 2u) coll |= 1u << jw;
#pragma unroll
        for (uint32_t jw = 0; jw < 8; ++jw)
            if ((coll >> jw) & 1u) {
                LookupCtx c;
                c.base = sm_hash[hslot8<BLOCK>(code[jw])] & LPHQ_TAIL_BASE_MASK;
                c.mode = 2u;
                uint64_t lo = 0, hi = 0;
                extract_kmer<K128>(sm_words, a + jw, k, lo, hi);
                hv[jw] = finish_hval<K128>(img, H, c, lo, hi, 0u);
            }
    }
    if constexpr (SINK == kSinkCodes) {
    if (one_record) {
        // A lane holds 8 consecutive codes: stored from the registers, every 16-byte store of a wave would touch 64 different 64-byte segments (lane
        // stride 64 B) -- measured, that made phase F 45 % slower than the 4-per-lane kernel's (profiles/r03_stream8.txt).  The codes go through LDS
        // instead (the hash array is free once every lane has read its bases) and leave as four fully coalesced stores: 2 KB contiguous per wave each.
        __syncthreads();
        ulonglong2* sm_out = (ulonglong2*)sm_hash;  // kS8Pos codes = 8 KB <= sizeof(sm_hash)
        static_assert(sizeof(uint64_t) * (8 * kS8Row + kS8Tail) >= sizeof(uint64_t) * kS8Pos, "the staged codes fit the hash array");
#pragma unroll
        for (uint32_t jw = 0; jw < 8; jw += 2) sm_out[4u * tid + jw / 2u] = make_ulonglong2(hv[jw], hv[jw + 1]);
        __syncthreads();
        uint64_t* const dst0 = out + (sm_roff[2] + (t0 - shift - sm_roff[0]));  // code of window 0 of the tile (wraps harmlessly for windows before the record)
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)kS8Pos / (2u * BLOCK); ++i) {
            const uint32_t idx = i * 2u * BLOCK + 2u * tid;  // windows idx, idx + 1
            const ulonglong2 v = sm_out[idx / 2u];
            if (idx >= w_lo && idx + 2u <= w_hi) store_codes_nt(dst0 + idx, v.x, v.y);
            else {
                if (idx >= w_lo && idx < w_hi) __builtin_nontemporal_store((uint64_t)v.x, dst0 + idx);
                if (idx + 1u >= w_lo && idx + 1u < w_hi) __builtin_nontemporal_store((uint64_t)v.y, dst0 + idx + 1u);
            }
        }
    } else {
        // the walk of phase C again, from the record found there: where each window's code goes
        const bool cached = rhi - rlo < (uint32_t)kS8RecCap;
        uint32_t j = jrec;
        uint64_t rs = cached ? sm_rtab[j] : seq_off[rlo + j], re = cached ? sm_rtab[j + 1] : seq_off[rlo + j + 1];
        uint64_t oidx[8];
#pragma unroll
        for (uint32_t jw = 0; jw < 8; ++jw) {
            oidx[jw] = 0;
            if (code[jw] == 0xFFFEu) continue;
            const uint64_t g = t0 + a + jw - shift;
            rec_walk(sm_rtab, seq_off, rlo, cached, g, j, rs, re);
            oidx[jw] = tile_record_out<kS8RecCap>(sm_rtab, out_off, rlo, rhi, j) + (g - rs);
        }
        bool all8 = oidx[7] == oidx[0] + 7u;
#pragma unroll
        for (uint32_t jw = 0; jw < 8; ++jw) all8 = all8 && code[jw] != 0xFFFEu;
        if (all8) {  // eight windows of one record (the common case for reads)
#pragma unroll
            for (uint32_t jw = 0; jw < 8; jw += 2) store_codes_nt(out + oidx[0] + jw, hv[jw], hv[jw + 1]);
        } else {
#pragma unroll
            for (uint32_t jw = 0; jw < 8; ++jw)
                if (code[jw] != 0xFFFEu) __builtin_nontemporal_store(hv[jw], out + oidx[jw]);
        }
    }
    } else if constexpr (SINK == kSinkGather) {
        // ---- the payload sink ----
        // where the lane's windows sit in the code space (what the codes sink computes on its way to the stores): one record under the tile -> window i of
        // the tile is at first0 + i; several -> the walk of phase C again, from the record found there
        auto walk_windows = [&](uint64_t (&oidx)[8]) {
            const bool cached = rhi - rlo < (uint32_t)kS8RecCap;
            uint32_t j = jrec;
            uint64_t rs = cached ? sm_rtab[j] : seq_off[rlo + j], re = cached ? sm_rtab[j + 1] : seq_off[rlo + j + 1];
#pragma unroll
            for (uint32_t jw = 0; jw < 8; ++jw) {
                oidx[jw] = 0;
                if (code[jw] == 0xFFFEu) continue;
                const uint64_t g = t0 + a + jw - shift;
                rec_walk(sm_rtab, seq_off, rlo, cached, g, j, rs, re);
                oidx[jw] = tile_record_out<kS8RecCap>(sm_rtab, out_off, rlo, rhi, j) + (g - rs);
            }
        };
        {
            // table[code] per window; element size uniform over the launch.
            const uint32_t el = sink.elem_log2;
            auto load_elem = [&](uint64_t c) -> uint64_t {
                const bool in = c < sink.n_slots;  // (LPHQ_INVALID and a non-member's wrapped code are not slots)
                const uint64_t i = in ? c : 0ull;
                uint64_t v;
                if (el == 0u) v = sink.table[i];
                else if (el == 1u) v = ((const uint16_t*)sink.table)[i];
                else if (el == 2u) v = ((const uint32_t*)sink.table)[i];
                else v = ((const uint64_t*)sink.table)[i];
                return in ? v : sink.missing;
            };
            uint64_t pv[8];  // the lane's eight elements
            struct __attribute__((packed, aligned(1))) u64a1 { uint64_t v; };
            struct __attribute__((packed, aligned(1))) u32a1 { uint32_t v; };
            struct __attribute__((packed, aligned(1))) u16a1 { uint16_t v; };
            auto store_elem = [&](uint64_t widx, uint64_t v) {  // one element, at window index widx of the output
                uint8_t* const p = sink.out + (widx << el);
                if (el == 0u) *p = (uint8_t)v;
                else if (el == 1u) ((u16a1*)p)->v = (uint16_t)v;
                else if (el == 2u) ((u32a1*)p)->v = (uint32_t)v;
                else ((u64a1*)p)->v = v;
            };
            {
                // A lane holds the codes of 8 CONSECUTIVE windows: loading table[code] from there, the 64 lanes of one load instruction sit 8 windows apart --
                // a different run, so a different line of the table, for almost every lane (C4: ~23 lines per instruction, C3: 64; eight instructions), and
                // the pass was bound by those line requests (measured: + 2.9 ms at C4, + 0.38 ms at C3 over the codes sink, in proportion to the RUNS).  The
                // codes go through LDS first (as the codes sink stages them for its stores) and come back transposed -- lane l takes windows l, l + BLOCK, ... --
                // so that one instruction covers 64 consecutive windows: the 3 - 10 runs under them, as many lines.  The elements then go to their place in
                // the staged tile (window w at byte w << el); a tile under one record leaves in 16-byte pieces, a tile of many records lane by lane (each lane
                // takes its eight elements back and stores them where the walk of the records says).
                static_assert(kS8Pos == 8 * BLOCK, "eight windows per lane");
                __syncthreads();  // (every lane has read the bases of its leaders)
                ulonglong2* const sm_codes = (ulonglong2*)sm_hash;  // kS8Pos codes = 8 KB <= sizeof(sm_hash) (asserted by the codes sink's twin)
#pragma unroll
                for (uint32_t jw = 0; jw < 8; jw += 2) sm_codes[4u * tid + jw / 2u] = make_ulonglong2(hv[jw], hv[jw + 1]);
                __syncthreads();
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i) pv[i] = load_elem(sm_hash[i * (uint32_t)BLOCK + tid]);
                __syncthreads();  // (every code has been read: the array now takes the elements)
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i) {
                    const uint32_t w = i * (uint32_t)BLOCK + tid;
                    if (el == 0u) ((uint8_t*)sm_hash)[w] = (uint8_t)pv[i];
                    else if (el == 1u) ((uint16_t*)sm_hash)[w] = (uint16_t)pv[i];
                    else if (el == 2u) ((uint32_t*)sm_hash)[w] = (uint32_t)pv[i];
                    else sm_hash[w] = pv[i];
                }
                __syncthreads();
            }
            uint64_t* const sm64 = sm_hash;
            if (one_record) {
                // bytes [w_lo << el, w_hi << el) of the staged tile -> the output, 16 at a time
                const uint32_t lo_b = w_lo << el, hi_b = w_hi << el;
                uint8_t* const dstb = sink.out + (first0 << el);  // byte address of window 0 of the tile
                typedef uint64_t u64x2a1 __attribute__((vector_size(16), aligned(1)));
                for (uint32_t c = tid; c < (((uint32_t)kS8Pos << el) >> 4); c += (uint32_t)BLOCK) {
                    const uint32_t b0 = 16u * c;
                    if (b0 >= hi_b || b0 + 16u <= lo_b) continue;
                    const uint64_t v0 = sm64[2u * c], v1 = sm64[2u * c + 1u];
                    if (b0 >= lo_b && b0 + 16u <= hi_b) {
                        u64x2a1 v;
                        v[0] = v0;
                        v[1] = v1;
                        __builtin_nontemporal_store(v, (u64x2a1*)(dstb + b0));
                    } else {  // a piece at the edge of the record: element by element
                        for (uint32_t e = 0; e < (16u >> el); ++e) {
                            const uint32_t w = (b0 >> el) + e, bit = (e << el) * 8u;
                            if (w >= w_lo && w < w_hi) store_elem(first0 + w, (bit < 64u ? v0 >> bit : v1 >> (bit - 64u)));
                        }
                    }
                }
            } else {
                // the lane's eight elements as they lie in the staged tile: 8 << el bytes = 1 << el words
                const uint32_t nq = 1u << el;
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i)
                    if (i < nq) pv[i] = sm64[tid * nq + i];
                auto elem_of = [&](uint32_t jw) -> uint64_t {  // (jw is a constant after unrolling)
                    if (el == 3u) return pv[jw];
                    if (el == 2u) return (pv[jw >> 1] >> ((jw & 1u) * 32u)) & 0xFFFFFFFFull;
                    if (el == 1u) return (pv[jw >> 2] >> ((jw & 3u) * 16u)) & 0xFFFFull;
                    return (pv[0] >> (jw * 8u)) & 0xFFull;
                };
                uint64_t oidx[8];
                walk_windows(oidx);
                bool all8 = oidx[7] == oidx[0] + 7u;
#pragma unroll
                for (uint32_t jw = 0; jw < 8; ++jw) all8 = all8 && code[jw] != 0xFFFEu;
                if (all8) {  // eight windows of one record: 8 << el contiguous bytes
                    uint8_t* const p = sink.out + (oidx[0] << el);
#pragma unroll
                    for (uint32_t i = 0; i < 8; ++i)
                        if (i < nq) ((u64a1*)(p + 8u * i))->v = pv[i];
                } else {
#pragma unroll
                    for (uint32_t jw = 0; jw < 8; ++jw)
                        if (code[jw] != 0xFFFEu) store_elem(oidx[jw], elem_of(jw));
                }
            }
        }
    }
}

// CALL(W8) for the value of (W - 8) % 8 at hand: what the launcher (stream8.hip) and the wave emulator (tests/emu) use to pick the instantiation of the
// wide-window variant
#define LPHQ_S8_SWITCH_W8(W, CALL) \
    switch (((W) - 8u) & 7u) {     \
        case 0: CALL(0); break;    \
        case 1: CALL(1); break;    \
        case 2: CALL(2); break;    \
        case 3: CALL(3); break;    \
        case 4: CALL(4); break;    \
        case 5: CALL(5); break;    \
        case 6: CALL(6); break;    \
        default: CALL(7); break;   \
    }

}  // namespace lphq
