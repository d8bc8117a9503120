// kernels_device.cuh -- the gfx950 kernels of the LPHash query path (launchers: kernels.hip).
//
//   k_query_packed        random (non-streaming) batch: one lane = one packed k-mer
//                         (MPHF::operator()(.., streaming=false), partitioned_mphf.hpp:185-195)
//   k_query_tiles_stream  all windows of a batch of ASCII records, tiled over the concatenated bases: one minimizer hash
//                         per base, one lookup per super-k-mer (MPHF::operator()(.., streaming=true), partitioned_mphf.hpp:78-184)
//   k_query_tiles_random  the same batch shape with every window doing the full random lookup
//   k_pack_kmers          debug::string_to_integer_no_reverse (mphf_utils.hpp:110-116) for chosen windows
//   k_count_valid         size of the reference's result vector for a streaming pass
//
// Wave64, 256-thread workgroups, one k-mer per lane / one tile per workgroup, so grids are >> 256 workgroups and all
// 256 CUs / 8 XCDs are busy; no inter-workgroup communication, so placement is irrelevant to correctness.
// LPHQ_PACKED_GRID, read once by the launcher, is a tuning aid only (validated and clamped; it changes no result).
#pragma once
#include <hip/hip_runtime.h>

#include "device_query.cuh"
#include "stream_sinks.h"

namespace lphq {

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------
// random batch over packed k-mers
// ------------------------------------------------------------------------------------------
// The k-mer and code streams carry the non-temporal hint: they are read / written once, the index lines are what should stay
// in L2 (+1.5 %, profiles/r02_nt_ab.txt).
template <int KIND, bool K128, int P>
// (header by value here: measured 2 % faster than reading it from the image, the opposite of the tile kernels)
__global__ __launch_bounds__(kBlock) void k_query_packed(const lphq_image_header H, const uint8_t* __restrict__ img,
                                                          const uint64_t* __restrict__ kmers, size_t n, uint64_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint64_t lo, hi = 0;
        if (K128) {
            lo = __builtin_nontemporal_load(kmers + 2 * i);
            hi = __builtin_nontemporal_load(kmers + 2 * i + 1);
        } else {
            lo = __builtin_nontemporal_load(kmers + i);
        }
        __builtin_nontemporal_store(query_kmer<KIND, K128, P>(img, H, lo, hi), out + i);
    }
}

// ------------------------------------------------------------------------------------------
// ASCII -> 2-bit helpers
// ------------------------------------------------------------------------------------------
// seq_nt4_table (constants.cpp:5-13): A/a=0 C/c=1 G/g=2 T/t/U/u=3, everything else 4 (invalid).
// For the 10 valid characters the code is ((c >> 1) ^ (c >> 2)) & 3.
__device__ __forceinline__ bool base_valid(uint32_t c) {
    const uint32_t u = c & 0xDFu;                                                         // fold case
    const uint32_t B = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20) | (1u << 21);      // A C G T U relative to 0x40
    return ((u & 0xE0u) == 0x40u) && ((B >> (u & 31u)) & 1u);
}
// four characters (first at the lowest byte) -> 8 bits, first base in the highest 2 bits; bad = 4-bit mask (bit i: character i is not a base).
// All four at once: the 2-bit code of every byte, the letter that code stands for (one v_perm_b32 into "ACGT"), and a byte is a base iff it is that
// letter after case folding -- or one above it where the code is 3 (U/u share T's code) -- which is the set base_valid() accepts.  ~20 instructions
// instead of ~50 for four base_valid() calls; the streaming kernel spends them on every base of the batch.
__device__ __forceinline__ uint32_t pack4(uint32_t x, uint32_t& bad) {
    const uint32_t t_raw = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
    const uint32_t want = __builtin_amdgcn_perm(0u, 0x54474341u, t_raw);  // byte i = "ACGT"[code i]
    uint32_t d = (x & 0xDFDFDFDFu) ^ want;                                 // 0: A C G T (either case), 1: U/u under code 3, anything else: not a base
    d &= ~(t_raw & (t_raw >> 1) & 0x01010101u);
    const uint32_t nz = ((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) >> 7) & 0x01010101u;  // 1 in every byte that is not a base
    bad = (nz * 0x01020408u) >> 24;                                                    // bits 0, 8, 16, 24 -> bits 0..3
    const uint32_t t = t_raw & ~(nz * 3u);  // invalid -> 0 ('A'), mphf_utils.hpp:108
    return (t * 0x40100401u) >> 24;
}

template <bool K128>
__global__ __launch_bounds__(kBlock) void k_pack_kmers(const char* __restrict__ seq, const uint64_t* __restrict__ pos, size_t n, uint32_t k,
                                                        uint64_t* __restrict__ out) {
    struct __attribute__((packed, aligned(1))) u32a1 { uint32_t v; };  // 4 bases per (unaligned) load instead of 1
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint8_t* s = (const uint8_t*)seq + pos[i];
        uint64_t lo = 0, hi = 0;
        uint32_t j = 0;
        for (; j + 4 <= k; j += 4) {
            uint32_t bad;
            const uint64_t p8 = pack4(((const u32a1*)(s + j))->v, bad);  // non-ACGT -> 0 ('A'), mphf_utils.hpp:108
            if (K128) hi = (hi << 8) | (lo >> 56);
            lo = (lo << 8) | p8;
        }
        for (; j < k; ++j) {
            const uint32_t c = s[j];
            const uint64_t code = base_valid(c) ? (((c >> 1) ^ (c >> 2)) & 3u) : 0u;
            if (K128) hi = (hi << 2) | (lo >> 62);
            lo = (lo << 2) | code;
        }
        if (K128) {
            ((ulonglong2*)out)[i] = make_ulonglong2(lo, hi);
        } else {
            out[i] = lo;
        }
    }
}

// ------------------------------------------------------------------------------------------
// tiles over concatenated records
// ------------------------------------------------------------------------------------------
constexpr int kTileW = 2048;  // window starts per workgroup, random-over-records kernel
// 2-bit packed u32 words (16 bases each, first base in the top bits) for TW window starts + k-1 (k <= 63) overlap
constexpr int tile_words(int tw) { return (tw + 64) / 16; }
constexpr int kTileWords = tile_words(kTileW);

// last r in [lo, hi] with offs[r] <= g   (offs non-decreasing, offs[lo] <= g)
__device__ __forceinline__ uint32_t find_record(const uint64_t* __restrict__ offs, uint32_t lo, uint32_t hi, uint64_t g) {
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo + 1) / 2;
        if (offs[mid] <= g) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// Same result, found by one wave in ~log64(n) rounds: 64 probes per round instead of a 1-load-per-step chain
// (every lane of the calling wave must pass the same arguments).
__device__ __forceinline__ uint32_t find_record_wave(const uint64_t* __restrict__ offs, uint32_t n_records, uint64_t g) {
    uint32_t lo = 0, hi = n_records - 1;
    const uint32_t lane = threadIdx.x & 63u;
    while (hi > lo) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint64_t idx = (uint64_t)lo + (uint64_t)(lane + 1u) * step;
        const bool ok = idx <= hi && offs[idx] <= g;  // monotone in the lane: offs is non-decreasing
        const uint32_t cnt = (uint32_t)__popcll(__ballot(ok));
        lo += cnt * step;
        hi = min(hi, lo + step - 1u);
    }
    return lo;
}

// Records of a tile cached in LDS.  A streaming tile spans <= 1088 bases, so it touches more than kRecCap records only if
// they average < 17 bases; such tiles (and nothing else) keep searching the global offset arrays.  Layout of the table:
// seq_off[rlo .. rlo + n] (n + 1 entries), then from entry kRecCap + 1 on out_off[rlo .. rlo + n - 1].
constexpr int kRecCap = 64;
// (CAP: records the table holds; the eight-per-lane kernel keeps 32 to leave LDS for one more workgroup per CU)
template <int CAP = kRecCap>
__device__ __forceinline__ void load_record_table(uint64_t* __restrict__ rtab, const uint64_t* __restrict__ seq_off, const uint64_t* __restrict__ out_off,
                                                  uint32_t rlo, uint32_t rhi, uint32_t tid) {
    const uint32_t n = rhi - rlo + 1u;
    if (n > 1u && n <= (uint32_t)CAP) {
        if (tid <= n) rtab[tid] = seq_off[rlo + tid];
        if (tid < n) rtab[CAP + 1 + tid] = out_off[rlo + tid];
    }
}
// record of base g (rlo <= record <= rhi, rlo != rhi): index, first base and end
template <int CAP = kRecCap>
__device__ __forceinline__ uint32_t tile_record(const uint64_t* __restrict__ rtab, const uint64_t* __restrict__ seq_off, uint32_t rlo, uint32_t rhi, uint64_t g,
                                                uint64_t& start, uint64_t& end) {
    if (rhi - rlo < (uint32_t)CAP) {
        const uint32_t j = find_record(rtab, 0, rhi - rlo, g);
        start = rtab[j];
        end = rtab[j + 1];
        return j;
    }
    const uint32_t r = find_record(seq_off, rlo, rhi, g);
    start = seq_off[r];
    end = seq_off[r + 1];
    return r - rlo;
}
template <int CAP = kRecCap>
__device__ __forceinline__ uint64_t tile_record_out(const uint64_t* __restrict__ rtab, const uint64_t* __restrict__ out_off, uint32_t rlo, uint32_t rhi, uint32_t j) {
    return (rhi - rlo < (uint32_t)CAP) ? rtab[CAP + 1 + j] : out_off[rlo + j];
}

// 2k bits starting at base q of the LDS-packed tile
template <bool K128>
__device__ __forceinline__ void extract_kmer(const uint32_t* __restrict__ sm, uint32_t q, uint32_t k, uint64_t& lo, uint64_t& hi) {
    const uint32_t w0 = q >> 4, o = (q & 15u) * 2u;
    if (K128) {
        const uint32_t a = sm[w0], b = sm[w0 + 1], c = sm[w0 + 2], d = sm[w0 + 3], e = sm[w0 + 4];
        // 128-bit big-endian stream starting at bit o of a
        const uint32_t x0 = o ? ((a << o) | (b >> (32 - o))) : a;
        const uint32_t x1 = o ? ((b << o) | (c >> (32 - o))) : b;
        const uint32_t x2 = o ? ((c << o) | (d >> (32 - o))) : c;
        const uint32_t x3 = o ? ((d << o) | (e >> (32 - o))) : d;
        const uint64_t H64 = ((uint64_t)x0 << 32) | x1, L64 = ((uint64_t)x2 << 32) | x3;
        const uint32_t s = 128u - 2u * k;  // 2..126
        if (s >= 64u) {
            lo = H64 >> (s - 64u);
            hi = 0;
        } else {
            lo = (L64 >> s) | (H64 << (64u - s));
            hi = H64 >> s;
        }
    } else {
        const uint32_t a = sm[w0], b = sm[w0 + 1], c = sm[w0 + 2];
        const uint32_t x0 = o ? ((a << o) | (b >> (32 - o))) : a;
        const uint32_t x1 = o ? ((b << o) | (c >> (32 - o))) : b;
        lo = ((((uint64_t)x0 << 32) | x1)) >> (64u - 2u * k);
        hi = 0;
    }
}

// Tile prologue of the random-over-records kernel: 16 bases per lane -> one 2-bit packed word (+ bad-base mask).
// Coordinates are shifted by `shift` = misalignment of the sequence pointer, so that every 16-byte chunk load is aligned;
// base g of the batch lives at shifted coordinate g + shift.
template <bool WANT_BAD, int NWORDS, int BLOCK = kBlock>
__device__ __forceinline__ void load_tile(const uint8_t* __restrict__ seq_al, uint64_t shift, uint64_t total, uint64_t t0 /* shifted, 16-aligned */,
                                          uint32_t* __restrict__ sm_words, uint32_t* __restrict__ sm_bad) {
    for (int c = threadIdx.x; c < NWORDS + 1; c += BLOCK) {
        const uint64_t a = t0 + (uint64_t)c * 16u;  // shifted coordinate of the chunk
        uint32_t word = 0, badm = 0xFFFFu;
        if (a < shift + total && a + 16u > shift) {
            const uint4 v = *(const uint4*)(seq_al + a);
            const uint32_t xs[4] = {v.x, v.y, v.z, v.w};
            badm = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t b4;
                const uint32_t p8 = pack4(xs[i], b4);
                word |= p8 << (24 - 8 * i);
                badm |= b4 << (4 * i);
            }
            if (a < shift || a + 16u > shift + total)  // bytes outside [shift, shift+total) are not bases
                for (int i = 0; i < 16; ++i)
                    if (a + i < shift || a + i >= shift + total) badm |= 1u << i;
        }
        sm_words[c] = word;
        if (WANT_BAD) ((uint16_t*)sm_bad)[c] = (uint16_t)badm;
    }
}

template <int KIND, bool K128, int P>
__global__ __launch_bounds__(kBlock) void k_query_tiles_random(const uint8_t* __restrict__ img, const uint8_t* __restrict__ seq_al,
                                                                uint64_t shift, uint64_t total, const uint64_t* __restrict__ seq_off,
                                                                const uint64_t* __restrict__ out_off, uint32_t n_records, uint64_t* __restrict__ out,
                                                                uint64_t tile_base /* first tile of this launch: batches beyond one grid are launched in pieces */) {
    const lphq_image_header& H = *(const lphq_image_header*)img;
    __shared__ uint32_t sm_words[kTileWords + 8];
    __shared__ uint32_t sm_bad[(kTileWords + 8) / 2 + 2];  // competitor harness (KIND 2, 3) only: 16 bad-base bits per 16-base chunk
    __shared__ uint32_t sm_rec[2];
    __shared__ uint64_t sm_roff[3];                // seq_off[rlo], seq_off[rlo+1], out_off[rlo]
    __shared__ uint64_t sm_rtab[2 * kRecCap + 1];  // multi-record tiles: the tile's slice of seq_off / out_off
    __shared__ uint32_t sm_queue[kTileW];          // multi-record tiles: window start | record-in-tile << 11 of every k-mer
    __shared__ uint32_t sm_qn;
    static_assert(kTileW == 2048 && kTileW % kBlock == 0, "queue entries carry the window start in 11 bits");
    const uint64_t t0 = (tile_base + blockIdx.x) * kTileW;  // shifted coordinate of the tile's first window start
    if (threadIdx.x == 0) sm_qn = 0;
    if (threadIdx.x < 64) {  // wave 0: record of the first base the tile can touch, and of the last (64 probes per round, as the streaming kernel)
        const uint64_t gfirst = t0 > shift ? t0 - shift : 0;
        uint64_t glast = t0 + kTileW + 64 > shift ? t0 + kTileW + 64 - shift : 0;
        if (glast >= total) glast = total ? total - 1 : 0;
        const uint32_t r = find_record_wave(seq_off, n_records, gfirst);
        const uint64_t s0 = seq_off[r], e0 = seq_off[r + 1];
        const uint32_t r2 = glast < e0 ? r : find_record_wave(seq_off, n_records, glast);
        if (threadIdx.x == 0) {
            sm_rec[0] = r;
            sm_rec[1] = r2;
            sm_roff[0] = s0;
            sm_roff[1] = e0;
            sm_roff[2] = out_off[r];
        }
    }
    // the competitor harness (KIND 2: PTHash, 3: BBHash) iterates like ptbb_file_itr: windows holding a non-ACGT base yield nothing -> LPHQ_INVALID
    load_tile<(KIND >= 2), kTileWords>(seq_al, shift, total, t0, sm_words, sm_bad);
    for (int c = kTileWords + 1 + threadIdx.x; c < kTileWords + 8; c += kBlock) {
        sm_words[c] = 0;
        if (KIND >= 2) ((uint16_t*)sm_bad)[c] = 0xFFFFu;
    }
    __syncthreads();
    const uint32_t rlo = sm_rec[0], rhi = sm_rec[1];
    if (rlo != rhi) {  // (uniform) read batches: the tile's record offsets into LDS, searched there
        load_record_table(sm_rtab, seq_off, out_off, rlo, rhi, threadIdx.x);
        __syncthreads();
    }
    auto window = [&](uint32_t i, uint64_t dst /* out index of window start i */) {
        uint64_t lo, hi;
        extract_kmer<K128>(sm_words, i, H.k, lo, hi);
        uint64_t code = query_kmer<KIND, K128, P>(img, H, lo, hi);
        if (KIND >= 2) {  // bad-base bits [i, i + k), k <= 63: five 16-bit chunks cover them
            const uint16_t* bad16 = (const uint16_t*)sm_bad;
            const uint32_t c0 = i >> 4, o = i & 15u;
            const uint64_t lo64 = (uint64_t)bad16[c0] | ((uint64_t)bad16[c0 + 1] << 16) | ((uint64_t)bad16[c0 + 2] << 32) | ((uint64_t)bad16[c0 + 3] << 48);
            const uint64_t bits = o ? ((lo64 >> o) | ((uint64_t)bad16[c0 + 4] << (64u - o))) : lo64;
            if (bits & ((1ull << H.k) - 1ull)) code = ~0ull;
        }
        __builtin_nontemporal_store(code, out + dst);
    };
    // (uniform) one record under the whole tile, or so few that nearly every window start is a k-mer: windows in place
    if ((rhi - rlo) * (H.k - 1u) < (uint32_t)kTileW / 16u) {
        for (int i = threadIdx.x; i < kTileW; i += kBlock) {
            const uint64_t a = t0 + i;
            if (a < shift) continue;
            const uint64_t g = a - shift;
            if (g + H.k > total) continue;
            uint64_t rs = sm_roff[0], re = sm_roff[1], ob = sm_roff[2];
            if (rlo != rhi) ob = tile_record_out(sm_rtab, out_off, rlo, rhi, tile_record(sm_rtab, seq_off, rlo, rhi, g, rs, re));
            if (g + H.k > re) continue;  // window would run over the end of its record
            window((uint32_t)i, ob + (g - rs));
        }
        return;
    }
    // read batches: the window starts that hold a k-mer (read length - k + 1 of every read) are queued first, so that the lookups run
    // with dense lanes however short the reads are
    for (int i = threadIdx.x; i < kTileW; i += kBlock) {  // (kTileW is a multiple of kBlock: whole waves iterate together)
        const uint64_t a = t0 + i;
        bool ok = a >= shift && a - shift + H.k <= total;
        uint32_t j = 0;
        if (ok) {
            uint64_t rs, re;
            j = tile_record(sm_rtab, seq_off, rlo, rhi, a - shift, rs, re);
            ok = a - shift + H.k <= re;
        }
        const uint64_t mask = __ballot(ok);
        uint32_t base = 0;
        if ((threadIdx.x & 63u) == 0 && mask) base = atomicAdd(&sm_qn, (uint32_t)__popcll(mask));
        base = __shfl(base, 0);
        if (ok) sm_queue[base + (uint32_t)__popcll(mask & ((1ull << (threadIdx.x & 63u)) - 1ull))] = (uint32_t)i | (j << 11);
    }
    __syncthreads();
    const uint32_t qn = sm_qn;
    for (uint32_t q = threadIdx.x; q < qn; q += kBlock) {
        const uint32_t e = sm_queue[q], i = e & 2047u, j = e >> 11;
        const uint64_t rs = (rhi - rlo < (uint32_t)kRecCap) ? sm_rtab[j] : seq_off[rlo + j];
        window(i, tile_record_out(sm_rtab, out_off, rlo, rhi, j) + (t0 + i - shift - rs));
    }
}


// ------------------------------------------------------------------------------------------
// streaming kernel: MPHF::operator()(.., streaming=true)   include/partitioned_mphf.hpp:78-184
// ------------------------------------------------------------------------------------------
// The reference walks a record base by base with a ring buffer of W = k-m+1 minimizer hashes and
// only calls query() when the minimizer changes (one call per super-k-mer); consecutive k-mers of a
// super-k-mer get hval +- 1.  Because the closed forms of query() are linear in the offset p of the
// minimizer (SURVEY section 8a note 3), the same codes come out position-parallel:
//   A  bases of the tile -> 2-bit words + bad-base bitmask in LDS
//   B  one MurmurHash2_64 per m-mer position (partitioned_mphf.hpp:116); the tile is sized so that the
//      positions are exactly 4 per lane
//   C  each lane takes 4 consecutive windows: their W+3 hashes are read once; the W-3 positions common to
//      all four are reduced once, the 3+3 edge positions as suffix / prefix minima.  Ties keep the LEFT
//      position, as the strict `>` comparisons of partitioned_mphf.hpp:122,150-163 do
//   D  windows whose minimizer position differs from their left neighbour's are super-k-mer leaders;
//      they are compacted (wave ballots) into an LDS queue so the gather chain runs with dense lanes
//   E  leaders: PTHash + wavelet block + plane block -> (base, mode) stored by minimizer position
//   F  every window: hval = base +- p (or the per-k-mer fallback lookup for colliding minimizers,
//      partitioned_mphf.hpp:133-135), coalesced stores
// Windows holding a non-ACGT base get LPHQ_INVALID (the reference emits nothing for them).
// two consecutive codes as one 16-byte non-temporal store; dst is only 8-byte aligned (gfx950 global accesses may be unaligned)
__device__ __forceinline__ void store_codes_nt(uint64_t* dst, uint64_t c0, uint64_t c1) {
    typedef uint64_t u64x2 __attribute__((vector_size(16), aligned(8)));
    u64x2 v;
    v[0] = c0;
    v[1] = c1;
    __builtin_nontemporal_store(v, (u64x2*)dst);
}
constexpr int kStreamBlockDefault = 256;  // threads per workgroup of the streaming kernel; 4 m-mer positions per lane
constexpr uint32_t kStream8MinW = 8;                   // the launcher uses this kernel for W >= kStream8MinW: window minima from the raw hashes for W <= 15 (P == 4), by
                                                       // block decomposition from W = 16 on (P == 6).  Against four positions per lane (profiles/r03_stream8.txt):
                                                       // W = 8 +1.6 %, W = 12 +3.3 %, W = 17 +7.5 %, W = 24 +5 %, W = 31 +7 %, W = 44 +20 %.  Narrower windows
                                                       // (every other window leads a super-k-mer) stay with four per lane
__host__ __device__ inline uint32_t stream_tile_windows(uint32_t W, int npos) { return (uint32_t)(npos - (int)W + 1) & ~15u; }

struct HashPos {
    uint64_t h;
    uint32_t p;
};
__device__ __forceinline__ void take_if_less(HashPos& m, uint64_t h, uint32_t p) {  // strict: the left one wins ties
    const bool lt = h < m.h;
    m.h = lt ? h : m.h;
    m.p = lt ? p : m.p;
}

// hash slot of m-mer position i: lane l owns positions 4l..4l+3 and reads up to 4l+W+2; storing position i at
// (i % 4) * ROW + i / 4 makes both the writes and the W+3 reads of consecutive lanes hit consecutive 8-byte slots (no LDS
// bank conflicts; the plain layout was 4-way conflicted: profiles/r01_stream_phase_counters.txt).  ROW = BLOCK + 17: the
// last lanes read up to position 4*BLOCK + 60 (W <= 62), i.e. 16 columns past the tile; those slots hold ~0 (never a
// minimum), so no index needs clamping and, for lane l, position 4l + c is at a compile-time offset from column l.
template <int BLOCK>
constexpr uint32_t hash_row() { return (uint32_t)BLOCK + 17u; }
template <int BLOCK>
constexpr uint32_t hash_slots() { return 4u * hash_row<BLOCK>(); }
template <int BLOCK>
__device__ __forceinline__ uint32_t hslot(uint32_t i) { return (i & 3u) * hash_row<BLOCK>() + (i >> 2); }
// ~0 in the slots of positions 4*BLOCK .. 4*BLOCK + 67 (call with every lane; part of phase B)
template <int BLOCK>
__device__ __forceinline__ void hash_pad(uint64_t* __restrict__ sm_hash, uint32_t tid) {
    if (tid < 4u * 17u) sm_hash[(tid / 17u) * hash_row<BLOCK>() + (uint32_t)BLOCK + tid % 17u] = ~0ull;
}

// Phases B + C for wide windows (W >= kBlockMinW): leftmost minimum of the 4 consecutive windows starting at local positions
// a .. a+3, a = 4 * tid, each W positions wide -> q[0..3] (local positions), by BLOCK DECOMPOSITION of the sliding minimum.  A lane hashes the four positions of its
// own block [a, a+4) and keeps, in registers, the suffix minima S[j] over a+j .. a+3; it publishes in LDS (column c of
// sm_hash, hslot layout) the prefix minima P[c] over a .. a+c -- P[3] is the minimum M of the whole block -- and, in one byte,
// which position each of them is.  Window a+j is then S[j] + the blocks it covers completely + a prefix of the block after them:
//     min( S[j],  M[l+1] .. M[l+nf],  P[l+nf+1][hc-1] ),   nf = (j + W - 4) / 4,  hc = (j + W - 4) % 4
// i.e. ~W/4 + 3 LDS reads and ~W/4 + 12 compare-selects per lane instead of W + 3 and ~W + 8 (W = 12: 5 reads instead of 15;
// W = 44: 13 instead of 47); the block range M[l+1 .. l+(W-4)/4] is shared by the lane's four windows.  Candidates are taken
// left to right with a strict `<`, so ties keep the LEFT position, as the strict `>` comparisons of
// partitioned_mphf.hpp:122,150-163 do.  Measured against the direct form on one box (profiles/r02_argmin_ab.txt): W = 44 +5 %,
// W = 31 +1 %, W = 12 -4 % (no fewer LDS instructions there, and every window pays the decode of its winner's position), hence
// the threshold.
constexpr uint32_t kBlockMinW = 24;
struct BlockMin {  // a lane's suffix minima, alive from phase B to phase C
    uint64_t h1, h2, h3;  // S[1], S[2], S[3] (S[0] = M, re-read from LDS)
    uint32_t p;           // their positions within the block, 2 bits each: S[1] | S[2] << 2 | S[3] << 4
};
// byte published per lane: position (0..3) within the block of the prefix minimum P[c] in bits [2c, 2c + 2) (P[0] is position 0; P[3] = M):
// one bit-field extract, no branches (the four-way select this replaces compiled to nested EXEC-masked branches, ~10 scalar instructions per window)
__device__ __forceinline__ uint32_t prefix_off(uint32_t ob, uint32_t c) { return (ob >> (2u * c)) & 3u; }

template <int BLOCK>
__device__ __forceinline__ BlockMin publish_block_minima(uint64_t* __restrict__ sm_hash, uint8_t* __restrict__ sm_off, uint32_t tid, const uint64_t (&h)[4]) {
    constexpr uint32_t ROW = hash_row<BLOCK>();
    HashPos p{h[0], 0};
    sm_hash[tid] = p.h;
    take_if_less(p, h[1], 1);
    sm_hash[ROW + tid] = p.h;
    uint32_t ob = p.p << 2;
    take_if_less(p, h[2], 2);
    sm_hash[2 * ROW + tid] = p.h;
    ob |= p.p << 4;
    take_if_less(p, h[3], 3);
    sm_hash[3 * ROW + tid] = p.h;
    ob |= p.p << 6;
    sm_off[tid] = (uint8_t)ob;
    BlockMin s;
    s.h3 = h[3];
    uint32_t p3 = 3, p2, p1;
    s.h2 = (h[2] <= s.h3) ? h[2] : s.h3;  // `<=`: scanning from the right, the left one wins ties
    p2 = (h[2] <= s.h3) ? 2u : p3;
    s.h1 = (h[1] <= s.h2) ? h[1] : s.h2;
    p1 = (h[1] <= s.h2) ? 1u : p2;
    s.p = p1 | (p2 << 2) | (p3 << 4);
    return s;
}

template <int BLOCK>
__device__ __forceinline__ void window_argmin4(const uint64_t* __restrict__ sm_hash, const uint8_t* __restrict__ sm_off, uint32_t tid, uint32_t W,
                                               const BlockMin& s, uint32_t (&q)[4]) {
    constexpr uint32_t ROW = hash_row<BLOCK>();
    const uint64_t* colM = sm_hash + 3 * ROW + tid;  // M of the block t lanes to the right: colM[t]
    // id of a candidate: (t << 2) | c for column c of the block t lanes to the right (t >= 1); the position itself for the own block
    const uint32_t nf0 = (W - 4u) >> 2;  // blocks covered completely by all four windows: l+1 .. l+nf0
    HashPos r{~0ull, 0};
    if (nf0) {
        r = HashPos{colM[1], (1u << 2) | 3u};
        for (uint32_t t = 2; t <= nf0; ++t) take_if_less(r, colM[t], (t << 2) | 3u);
    }
    const uint64_t m0 = colM[0];
    const uint32_t ob0 = sm_off[tid];
    HashPos w[4] = {HashPos{m0, (ob0 >> 6) & 3u}, HashPos{s.h1, s.p & 3u}, HashPos{s.h2, (s.p >> 2) & 3u}, HashPos{s.h3, (s.p >> 4) & 3u}};
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t nf = (j + W - 4u) >> 2, hc = (j + W - 4u) & 3u;  // uniform over the wave
        take_if_less(w[j], r.h, r.p);  // (no full block: r.h = ~0, never smaller -- unconditional, a uniform `if` here compiled to three more selects per window)
        if (nf > nf0) take_if_less(w[j], colM[nf], (nf << 2) | 3u);
        if (hc) take_if_less(w[j], sm_hash[(hc - 1u) * ROW + tid + nf + 1u], ((nf + 1u) << 2) | (hc - 1u));
        // the winner is column c of the block t lanes to the right (its place inside that block is in the block's byte), or -- t = 0 -- position c of the
        // own block: one LDS byte read for every lane (t = 0 reads its own and ignores it) instead of a divergent branch
        const uint32_t t = w[j].p >> 2, c = w[j].p & 3u;
        const uint32_t off = prefix_off(sm_off[tid + t], c);
        q[j] = 4u * (tid + t) + (t ? off : c);
    }
}

// Phase C, direct form (W < kBlockMinW): raw hashes in sm_hash (hslot layout); leftmost minimum of the 4 consecutive windows
// starting at local positions a .. a+3, a = 4 * tid, each W positions wide -> q[0..3] (local positions).  Offsets from a are
// uniform over the wave, so they live in scalar registers and the W-3 positions common to the four windows are read four per
// step at immediate LDS offsets.  Ties keep the LEFT position, as the strict `>` comparisons of partitioned_mphf.hpp:122,150-163 do.
template <int BLOCK>
__device__ __forceinline__ void window_argmin4_raw(const uint64_t* __restrict__ sm_hash, uint32_t tid, uint32_t W, uint32_t (&q)[4]) {
    constexpr uint32_t ROW = hash_row<BLOCK>();
    const uint64_t* col = sm_hash + tid;  // position a + c: col[(c & 3) * ROW + (c >> 2)]
    auto H = [&](uint32_t c) { return col[(c & 3u) * ROW + (c >> 2)]; };
    const uint32_t a = 4u * tid;
    if (W >= 4u) {
        // left edge: suffix minima over offsets 2, 1, 0 (`<=`: the left one wins ties)
        HashPos s2{H(2), 2}, s1, s0;
        {
            const uint64_t h1 = H(1), h0 = H(0);
            s1 = (h1 <= s2.h) ? HashPos{h1, 1} : s2;
            s0 = (h0 <= s1.h) ? HashPos{h0, 0} : s1;
        }
        // core: offsets 3 .. W-1, common to the four windows
        HashPos c{H(3), 3};
        uint32_t o = 4;
        for (; o + 4u <= W; o += 4u) {
            const uint64_t* p = col + (o >> 2);
            const uint64_t h0 = p[0], h1 = p[ROW], h2 = p[2 * ROW], h3 = p[3 * ROW];
            take_if_less(c, h0, o);
            take_if_less(c, h1, o + 1u);
            take_if_less(c, h2, o + 2u);
            take_if_less(c, h3, o + 3u);
        }
        for (; o < W; ++o) take_if_less(c, H(o), o);
        // right edge: prefix minima over offsets W, W+1, W+2
        HashPos w0 = s0, w1 = s1, w2 = s2, w3 = c;
        take_if_less(w0, c.h, c.p);
        take_if_less(w1, c.h, c.p);
        take_if_less(w2, c.h, c.p);
        HashPos r{H(W), W};
        take_if_less(w1, r.h, r.p);
        take_if_less(w2, r.h, r.p);
        take_if_less(w3, r.h, r.p);
        r = HashPos{H(W + 1u), W + 1u};
        take_if_less(w2, r.h, r.p);
        take_if_less(w3, r.h, r.p);
        r = HashPos{H(W + 2u), W + 2u};
        take_if_less(w3, r.h, r.p);
        q[0] = a + w0.p;
        q[1] = a + w1.p;
        q[2] = a + w2.p;
        q[3] = a + w3.p;
    } else {
#pragma unroll
        for (uint32_t jw = 0; jw < 4; ++jw) {
            HashPos c{H(jw), jw};
            for (uint32_t t = 1; t < W; ++t) take_if_less(c, H(jw + t), jw + t);
            q[jw] = a + c.p;
        }
    }
}

// Multi-record tiles: a lane resolves the record of its FIRST window with one search and walks on from there for the other three (a lane's four
// windows are consecutive bases, so they leave a record at most through its end).  rec_walk: record of base g, given a record (j, [rs, re)) that
// starts at or before g; the offsets come from the tile's LDS table, or from global memory for tiles with more than kRecCap records.
__device__ __forceinline__ void rec_walk(const uint64_t* __restrict__ rtab, const uint64_t* __restrict__ seq_off, uint32_t rlo, bool cached, uint64_t g,
                                         uint32_t& j, uint64_t& rs, uint64_t& re) {
    while (g >= re) {  // (skips empty records)
        ++j;
        rs = re;
        re = cached ? rtab[j + 1] : seq_off[rlo + j + 1];
    }
}

// Shaped by the phase profile of its predecessor (profiles/r02_stream_phases.txt, r02_stream_v4_ab.txt: that kernel moved 4.1 GB
// per launch at ~5.8 TB/s and was short of gathers in flight, because a workgroup holds its slot ~15 us of which the lookup
// phase E -- the only one with loads in flight -- was 30 %; the other phases were latency (A) and instruction (C, D, F) time):
//   A  the lane's sequence dwords are requested before the record search, so the search's dependent loads overlap with them;
//      one wave searches (the last base of a tile is almost always in the record of the first)
//   C  window validity from a per-tile [w_lo, w_hi) range when the tile lies in one record, and no bad-base arithmetic when
//      the tile has no non-ACGT base (the common case): ~2 instructions per window instead of ~25
//   D  fused into C: a lane keeps its four consecutive windows in registers; leaders are found with one cross-lane move, four
//      ballots and ONE LDS atomic per wave.  The first window of a wave is always queued as a leader (its left neighbour lives
//      in another wave): at most three redundant lookups per tile, which write the value the real leader writes
//   F  codes from registers, two 16-byte non-temporal stores per lane when its four windows are all valid (the codes are
//      written once and never read here; the hint keeps more of L2 for the index lines: +2.5-5 %, profiles/r02_nt_ab.txt)
// BLKMIN: window minima by block decomposition (the launcher picks it when W >= kBlockMinW; a template parameter because both
// forms in one kernel cost 78 VGPRs instead of 57, i.e. 6 instead of 8 waves per SIMD)
template <int KIND, bool K128, int P, int BLOCK, bool BLKMIN>
__global__ __launch_bounds__(BLOCK) void k_query_tiles_stream(const uint8_t* __restrict__ img, const uint8_t* __restrict__ seq_al,
                                                                 uint64_t shift, uint64_t total, const uint64_t* __restrict__ seq_off,
                                                                 const uint64_t* __restrict__ out_off, uint32_t n_records, uint64_t* __restrict__ out,
                                                                 uint64_t tile_base /* first tile of this launch */) {
    const lphq_image_header& H = *(const lphq_image_header*)img;
    constexpr int kStreamPos = 4 * BLOCK, NW = tile_words(kStreamPos);
    constexpr int NDW = (NW + 1) * 4;        // dwords of sequence (4 bases each) the tile loads; NDW - BLOCK = 20 lanes of the last wave load two
    static_assert(NDW > BLOCK && NDW <= BLOCK + 64, "the dwords beyond the first BLOCK are loaded by the lanes of the last wave");
    __shared__ uint64_t sm_hash[hash_slots<BLOCK>()];  // phase B-C: minimizer hash per m-mer position (hslot order), or the blocks' prefix minima when W >= kBlockMinW; phase E-F: the leader's base
    __shared__ uint32_t sm_words[NW + 8];
    __shared__ uint8_t sm_badb[NDW + 12];  // bad-base flags, one byte per sequence dword: bit i = base i of the dword is not a base of the batch (every entry is
                                           // written by the lane that packs the dword, so the packing needs no barrier after an initialisation)
    __shared__ uint32_t sm_anybad4[BLOCK / 64];  // per wave: one of its dwords holds a real non-ACGT base
    __shared__ uint16_t sm_queue[kStreamPos];
    __shared__ uint8_t sm_mode[kStreamPos];
    __shared__ uint8_t sm_off[hash_row<BLOCK>() + 3];  // phase B-C: where in its block each published prefix minimum sits
    __shared__ uint32_t sm_rec[2], sm_count;
    __shared__ uint64_t sm_roff[3];                // seq_off[rlo], seq_off[rlo+1], out_off[rlo]
    __shared__ uint64_t sm_rtab[2 * kRecCap + 1];  // multi-record tiles: the tile's slice of seq_off / out_off
    const uint32_t tid = threadIdx.x;
    const uint32_t k = H.k, m = H.m, W = H.W;
    const uint32_t TW = stream_tile_windows(W, kStreamPos);
    const uint64_t t0 = (tile_base + blockIdx.x) * TW;  // shifted coordinate of the tile's first window start

    // ---- A: sequence dwords first (no dependence on the record search), then the search, then pack into LDS ----
    // (uniform) every dword the tile loads lies inside the batch: true for all tiles but the first and the last, whose lanes check their bytes one by one
    const bool interior = t0 >= shift && t0 + 4ull * (uint64_t)NDW <= shift + total;
    uint32_t xs[2] = {0u, 0u};
    bool have[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t d = r == 0 ? tid : tid + 64u;  // second dword: the last wave's lanes (wave 0 has the record search on its hands)
        const uint64_t a = t0 + (uint64_t)d * 4u;
        if (interior) have[r] = d >= (uint32_t)r * BLOCK && d < (uint32_t)NDW;
        else have[r] = d >= (uint32_t)r * BLOCK && d < (uint32_t)NDW && a < shift + total && a + 4u > shift;
        if (have[r]) xs[r] = *(const uint32_t*)(seq_al + a);
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
        }
    }
    if (tid < 12u) sm_badb[NDW + tid] = 0xFu;  // beyond the tile's dwords: not bases
    for (int c = NW + 1 + tid; c < NW + 8; c += BLOCK) sm_words[c] = 0;
    // (no barrier here: the packing below touches only what its own lane loaded; the record search of wave 0 is awaited at the next barrier)
    bool real_bad = false;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t d = r == 0 ? tid : tid + 64u;
        if (d >= (uint32_t)r * BLOCK && d < (uint32_t)NDW) {
            const uint64_t a = t0 + (uint64_t)d * 4u;
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
        if ((tid & 63u) == 0u) sm_anybad4[tid >> 6] = wave_bad ? 1u : 0u;
    }
    __syncthreads();
    if (sm_rec[0] != sm_rec[1]) load_record_table(sm_rtab, seq_off, out_off, sm_rec[0], sm_rec[1], tid);  // (visible to phase C through the barrier after phase B)

    // ---- B: hashes of the m-mers starting at local positions 4*lane .. 4*lane+3; block minima published for phase C ----
    BlockMin bmin;
    {
        const uint64_t mm_seedx = murmur_seedx(H.mm_seed);
        const uint32_t a = 4u * tid;
        const uint32_t w0 = a >> 4, o = (a & 15u) * 2u;
        const uint32_t x0 = sm_words[w0], x1 = sm_words[w0 + 1], x2 = sm_words[w0 + 2], x3 = sm_words[w0 + 3];
        const uint64_t A64 = ((uint64_t)x0 << 32) | x1, C64 = ((uint64_t)x2 << 32) | x3;
        const uint64_t s0 = o ? ((A64 << o) | ((uint64_t)x2 >> (32 - o))) : A64;  // bits [0, 64) of the stream at base a
        const uint64_t s1 = o ? (C64 << o) : C64;                                  // bits [64, 128): only the top 6 are needed
        const uint32_t sh = 64u - 2u * m;
        const uint64_t mask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
        uint64_t h[4];
        if (m <= 29u) {  // (uniform) the four m-mers end inside the first 64 bits: one shift each
#pragma unroll
            for (uint32_t r = 0; r < 4; ++r) h[r] = murmur8((s0 >> (sh - 2 * r)) & mask, mm_seedx);
        } else {
#pragma unroll
            for (uint32_t r = 0; r < 4; ++r) {
                const uint64_t v = r == 0 ? (s0 >> sh) : (((s0 << (2 * r)) | (s1 >> (64 - 2 * r))) >> sh);
                h[r] = murmur8(v & mask, mm_seedx);
            }
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

    // ---- C + D: leftmost minimum of the lane's 4 windows, their validity, leaders -> queue ----
    const uint32_t a = 4u * tid;
    const uint32_t rlo = sm_rec[0], rhi = sm_rec[1];
    const bool one_record = rlo == rhi;
    bool anybad = false;
#pragma unroll
    for (int wv = 0; wv < BLOCK / 64; ++wv) anybad = anybad || sm_anybad4[wv] != 0u;
    // one record: window i of the tile is a window of the record iff w_lo <= i < w_hi
    const uint32_t w_lo = t0 >= shift ? 0u : (uint32_t)(shift - t0);
    uint32_t w_hi = 0;
    {
        const uint64_t limit = sm_roff[1] + shift;  // shifted coordinate of the record's end
        if (limit >= t0 + k) w_hi = (uint32_t)min((uint64_t)TW, limit - k - t0 + 1u);
    }
    uint32_t code[4];  // minimizer position of the window; 0xFFFF: holds a non-ACGT base; 0xFFFE: not a window of any record
    uint32_t jrec = 0, jcur = 0;  // multi-record tiles: record (relative to rlo) of the lane's first in-batch window, and a cursor
    {
        uint32_t q[4];
        if (BLKMIN) window_argmin4<BLOCK>(sm_hash, sm_off, tid, W, bmin, q);
        else window_argmin4_raw<BLOCK>(sm_hash, tid, W, q);
        if (one_record) {
#pragma unroll
            for (uint32_t jw = 0; jw < 4; ++jw) code[jw] = (a + jw >= w_lo && a + jw < w_hi) ? q[jw] : 0xFFFEu;
        } else {
            bool have = false;
            uint64_t rs = 0, re = 0;
#pragma unroll
            for (uint32_t jw = 0; jw < 4; ++jw) {
                const uint32_t i = a + jw;
                const uint64_t ga = t0 + i;
                uint32_t v = 0xFFFEu;
                if (i < TW && ga >= shift && ga - shift + k <= total) {
                    const uint64_t g = ga - shift;
                    if (!have) {  // one search per lane; its record index is kept for phase F
                        jrec = tile_record(sm_rtab, seq_off, rlo, rhi, g, rs, re);
                        jcur = jrec;
                        have = true;
                    }
                    rec_walk(sm_rtab, seq_off, rlo, rhi - rlo < (uint32_t)kRecCap, g, jcur, rs, re);
                    if (g + k <= re) v = q[jw];
                }
                code[jw] = v;
            }
        }
        if (anybad) {  // uniform over the workgroup
            const uint64_t kmask = (1ull << k) - 1ull;  // k <= 63
            // bad-base bits [a, a + 68): windows a..a+3 need bits up to a + 3 + 62; one nibble per sequence dword, the lane's first dword is tid
            uint64_t blo = 0;
#pragma unroll
            for (uint32_t d = 0; d < 16; ++d) blo |= (uint64_t)(sm_badb[tid + d] & 0xFu) << (4 * d);
            const uint32_t bhi = sm_badb[tid + 16u] & 0xFu;
#pragma unroll
            for (uint32_t jw = 0; jw < 4; ++jw) {
                const uint64_t x = jw ? ((blo >> jw) | ((uint64_t)bhi << (64 - jw))) : blo;
                if (code[jw] != 0xFFFEu && (x & kmask)) code[jw] = 0xFFFFu;
            }
        }
    }
    {
        const uint32_t lane = tid & 63u;
        uint32_t prev = __shfl(code[3], (int)((lane + 63u) & 63u), 64);  // last window of the lane to the left
        if (lane == 0) prev = 0xFFFEu;                                     // it lives in another wave: queue the window regardless
        const bool l0 = code[0] < 0xFFFEu && code[0] != prev, l1 = code[1] < 0xFFFEu && code[1] != code[0];
        const bool l2 = code[2] < 0xFFFEu && code[2] != code[1], l3 = code[3] < 0xFFFEu && code[3] != code[2];
        const uint64_t b0 = __ballot(l0), b1 = __ballot(l1), b2 = __ballot(l2), b3 = __ballot(l3);
        const uint32_t n0 = (uint32_t)__popcll(b0), n1 = (uint32_t)__popcll(b1), n2 = (uint32_t)__popcll(b2), n3 = (uint32_t)__popcll(b3);
        uint32_t base = 0;
        if (lane == 0 && (n0 + n1 + n2 + n3)) base = atomicAdd(&sm_count, n0 + n1 + n2 + n3);
        base = __shfl(base, 0, 64);
        const uint64_t lt = (1ull << lane) - 1ull;
        if (l0) sm_queue[base + (uint32_t)__popcll(b0 & lt)] = (uint16_t)code[0];
        if (l1) sm_queue[base + n0 + (uint32_t)__popcll(b1 & lt)] = (uint16_t)code[1];
        if (l2) sm_queue[base + n0 + n1 + (uint32_t)__popcll(b2 & lt)] = (uint16_t)code[2];
        if (l3) sm_queue[base + n0 + n1 + n2 + (uint32_t)__popcll(b3 & lt)] = (uint16_t)code[3];
    }
    __syncthreads();

    // ---- E: one lookup per super-k-mer, dense lanes ----
    const uint32_t nlead = sm_count;
    for (uint32_t x = tid; x < nlead; x += BLOCK) {
        const uint32_t q = sm_queue[x];
        uint64_t mm, hi;
        extract_kmer<false>(sm_words, q, m, mm, hi);
        const LookupCtx c = lookup_ctx<KIND, K128, P>(img, H, mm);
        sm_hash[hslot<BLOCK>(q)] = c.base;
        sm_mode[q] = (uint8_t)c.mode;
    }
    __syncthreads();

    // ---- F: the lane's four codes ----
    // straight-line for the common case (base +- offset); windows under a colliding minimizer (fallback function of the k-mer,
    // partitioned_mphf.cpp:312-316) are redone afterwards, in the rare waves that hold one
    uint64_t hv[4];
    uint32_t coll = 0;
#pragma unroll
    for (uint32_t jw = 0; jw < 4; ++jw) {
        const uint32_t q = code[jw];
        const bool valid = q < 0xFFFEu;
        const uint32_t qq = valid ? q : a + jw;  // (any slot of the tile: the value read is discarded)
        const uint64_t base = sm_hash[hslot<BLOCK>(qq)];
        const uint32_t mode = sm_mode[qq];
        const uint32_t p = qq - (a + jw);
        const int32_t delta = mode == 0u ? (int32_t)p : -(int32_t)p;  // |p| < 2^16; (mode 2 is redone below)
        const uint64_t hval = base + (uint64_t)(int64_t)delta;
        hv[jw] = valid ? hval : ~0ull;  // LPHQ_INVALID
        coll |= (valid && mode == 2u) ? 1u << jw : 0u;
    }
    if (__any(coll != 0u)) {
#pragma unroll
        for (uint32_t jw = 0; jw < 4; ++jw)
            if ((coll >> jw) & 1u) {
                LookupCtx c;
                c.base = sm_hash[hslot<BLOCK>(code[jw])];
                c.mode = 2u;
                uint64_t lo = 0, hi = 0;
                extract_kmer<K128>(sm_words, a + jw, k, lo, hi);
                hv[jw] = finish_hval<K128>(img, H, c, lo, hi, 0u);
            }
    }
    if (one_record) {
        // A lane holds 4 consecutive codes: stored from the registers, a 16-byte store of a wave covers 2 KB of memory half-filled (lane stride 32 B).
        // Through LDS instead (the hash array is free once every lane has read its bases) they leave as fully coalesced stores: 1 KB contiguous per
        // wave and instruction (the eight-per-lane kernel, where the stride is 64 B, gained 8 % from this: profiles/r03_stream8.txt).
        __syncthreads();
        ulonglong2* sm_out = (ulonglong2*)sm_hash;  // kStreamPos codes <= the hash array
        static_assert(hash_slots<BLOCK>() >= (uint32_t)kStreamPos, "the staged codes fit the hash array");
        sm_out[2u * tid] = make_ulonglong2(hv[0], hv[1]);
        sm_out[2u * tid + 1u] = make_ulonglong2(hv[2], hv[3]);
        __syncthreads();
        uint64_t* const dst0 = out + (sm_roff[2] + (t0 - shift - sm_roff[0]));  // code of window 0 of the tile (wraps harmlessly for windows before the record)
#pragma unroll
        for (uint32_t i = 0; i < 2u; ++i) {
            const uint32_t idx = i * 2u * (uint32_t)BLOCK + 2u * tid;  // windows idx, idx + 1
            const ulonglong2 v = sm_out[idx / 2u];
            if (idx >= w_lo && idx + 2u <= w_hi) store_codes_nt(dst0 + idx, v.x, v.y);
            else {
                if (idx >= w_lo && idx < w_hi) __builtin_nontemporal_store((uint64_t)v.x, dst0 + idx);
                if (idx + 1u >= w_lo && idx + 1u < w_hi) __builtin_nontemporal_store((uint64_t)v.y, dst0 + idx + 1u);
            }
        }
    } else {
        // the walk of phase C again, from the record found there: where each window's code goes
        const bool cached = rhi - rlo < (uint32_t)kRecCap;
        uint32_t j = jrec;
        uint64_t rs = cached ? sm_rtab[j] : seq_off[rlo + j], re = cached ? sm_rtab[j + 1] : seq_off[rlo + j + 1];
        uint64_t oidx[4];
#pragma unroll
        for (uint32_t jw = 0; jw < 4; ++jw) {
            oidx[jw] = 0;
            if (code[jw] == 0xFFFEu) continue;
            const uint64_t g = t0 + a + jw - shift;
            rec_walk(sm_rtab, seq_off, rlo, cached, g, j, rs, re);
            oidx[jw] = tile_record_out(sm_rtab, out_off, rlo, rhi, j) + (g - rs);
        }
        if (code[0] != 0xFFFEu && code[3] != 0xFFFEu && oidx[3] == oidx[0] + 3u) {  // four windows of one record (the common case for reads)
            store_codes_nt(out + oidx[0], hv[0], hv[1]);
            store_codes_nt(out + oidx[0] + 2, hv[2], hv[3]);
        } else {
#pragma unroll
            for (uint32_t jw = 0; jw < 4; ++jw)
                if (code[jw] != 0xFFFEu) __builtin_nontemporal_store(hv[jw], out + oidx[jw]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// number of valid codes (size of the reference's result vector for a streaming pass)
// ------------------------------------------------------------------------------------------
template <int BLOCK = kBlock>  // (a template only so that every translation unit including this header may hold a copy)
__global__ __launch_bounds__(kBlock) void k_count_valid(const uint64_t* __restrict__ codes, size_t n, unsigned long long* __restrict__ count) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) c += codes[i] != ~0ull;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// bytes of seq[0, n) that are no base (base_valid): the batched host-buffer path (lphq_session_*) asks this once per batch -- zero, the usual
// answer, means that every window was emitted and the host has nothing to compact (the reference's streaming pass drops the windows that hold such a
// byte, include/partitioned_mphf.hpp:179-183)
template <int UNUSED = 0>
__global__ __launch_bounds__(kBlock) void k_count_bad_bases(const uint8_t* __restrict__ seq, size_t n, unsigned long long* __restrict__ count) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) c += !base_valid(seq[i]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// ------------------------------------------------------------------------------------------
// result consumer: payload[code] for every code -- the main downstream use of an MPHF
// ------------------------------------------------------------------------------------------
// The reference throws the codes away (src/query.cpp:54); a caller that indexes per-k-mer data by them gathers here without a
// trip through the host.  Codes that are not slots of the function (>= n_slots: LPHQ_INVALID, or the wrapped arithmetic of a
// non-member k-mer) yield `missing`.  The code stream is read once (non-temporal), the table is what should stay cached.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_gather_payload(const uint64_t* __restrict__ codes, size_t n, const T* __restrict__ table, uint64_t n_slots, T missing,
                                                          T* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t c = __builtin_nontemporal_load(codes + i);
        __builtin_nontemporal_store(c < n_slots ? table[c] : missing, out + i);
    }
}

// ------------------------------------------------------------------------------------------
// the run-encoded streaming result (include/lphq.h: lphq_run; produced by k_query_tiles_stream8<.., kSinkRuns>)
// ------------------------------------------------------------------------------------------
// lphq_expand_runs: out[first + i] = code +- i (or the explicit code) for every run of every arena.  A workgroup takes 256 runs of one arena at a time:
// their lengths are summed in LDS (inclusive scan), and the windows of the 256 runs are then dealt out to the lanes one each per step, found by a binary
// search over the scan -- dense lanes whatever the mix of run lengths (1 .. 512).  `out` was filled with LPHQ_INVALID by the launcher.  status: set to 1 when an
// arena had overflowed or a run points outside [0, n_out).
template <int BLOCK = kBlock>
__global__ __launch_bounds__(kBlock) void k_expand_runs(const lphq_run* __restrict__ runs, uint64_t arena_runs, const uint64_t* __restrict__ codes, uint64_t arena_codes,
                                                        const lphq_run_counts* __restrict__ counts, uint64_t* __restrict__ out, uint64_t n_out,
                                                        unsigned long long* __restrict__ status) {
    __shared__ uint64_t sm_code[kBlock], sm_where[kBlock];
    __shared__ uint32_t sm_incl[kBlock], sm_wsum[kBlock / 64];
    const uint32_t tid = threadIdx.x, arena = blockIdx.y;
    const uint64_t have = counts[arena].runs, n = have < arena_runs ? have : arena_runs;
    bool bad = have > arena_runs || counts[arena].codes > arena_codes;
    const lphq_run* const src = runs + (uint64_t)arena * arena_runs;
    const uint64_t* const xsrc = codes + (uint64_t)arena * arena_codes;
    for (uint64_t c0 = (uint64_t)blockIdx.x * kBlock; c0 < n; c0 += (uint64_t)gridDim.x * kBlock) {
        lphq_run r{0, 0};
        if (c0 + tid < n) r = src[c0 + tid];
        uint32_t len = LPHQ_RUN_LENGTH(r);
        const uint64_t first = LPHQ_RUN_FIRST(r);
        if (first + len > n_out || (LPHQ_RUN_KIND(r) == LPHQ_RUN_EXPLICIT && (r.code > arena_codes || len > arena_codes - r.code)) || LPHQ_RUN_KIND(r) == 3u) {  // (no sum that could wrap)  // not a run of this batch: skipped
            bad = bad || len != 0u;
            len = 0;
        }
        // inclusive scan of the lengths over the workgroup: shuffles inside a wave, the waves' sums through LDS
        uint32_t incl = len;
        for (uint32_t o = 1; o < 64u; o <<= 1) {
            const uint32_t up = __shfl(incl, (int)((tid & 63u) >= o ? (tid & 63u) - o : 0u), 64);
            if ((tid & 63u) >= o) incl += up;
        }
        if ((tid & 63u) == 63u) sm_wsum[tid >> 6] = incl;
        __syncthreads();
        for (uint32_t w = 0; w < (tid >> 6); ++w) incl += sm_wsum[w];
        sm_code[tid] = r.code;
        sm_where[tid] = (r.where & ~(0x3FFFFFull << 40)) | ((uint64_t)len << 40);
        sm_incl[tid] = incl;
        __syncthreads();
        const uint32_t total = sm_incl[kBlock - 1];
        for (uint32_t e = tid; e < total; e += kBlock) {
            uint32_t lo = 0, hi = kBlock - 1;  // first run whose inclusive sum exceeds e
            while (lo < hi) {
                const uint32_t mid = (lo + hi) / 2u;
                if (sm_incl[mid] > e) hi = mid;
                else lo = mid + 1u;
            }
            const uint64_t w = sm_where[lo], code = sm_code[lo];
            const uint32_t i = e - (sm_incl[lo] - (uint32_t)((w >> 40) & 0x3FFFFFu)), kind = (uint32_t)(w >> 62);
            const uint64_t v = kind == (uint32_t)LPHQ_RUN_EXPLICIT ? xsrc[code + i] : (kind == (uint32_t)LPHQ_RUN_UP ? code + i : code - i);
            out[(w & 0xFFFFFFFFFFull) + i] = v;
        }
        __syncthreads();
    }
    if (bad && status) *status = 1ull;
}

// The arenas' counters back to zero before a streaming pass appends to them: a 64-thread kernel instead of hipMemsetAsync, whose barrier packets cost the
// stream ~30 us per call (a C3-sized pass is 490 us)
template <int UNUSED = 0>
__global__ __launch_bounds__(64) void k_zero_run_counts(lphq_run_counts* __restrict__ counts) {
    uint64_t* const w = (uint64_t*)counts;
    for (uint32_t i = threadIdx.x; i < (uint32_t)(sizeof(lphq_run_counts) / 8u) * (uint32_t)LPHQ_RUN_ARENAS; i += 64u) w[i] = 0;
}

// Sessions move a batch's runs over PCIe as ONE dense array: arena after arena, explicit-code indices rebased to the dense code array.  totals[0] = runs,
// totals[1] = explicit codes, totals[2] = 1 if an arena had overflowed (the batch must be answered some other way), totals[3] = windows in all runs (what the
// reference's vector.size() adds up to over the batch; zeroed by the launcher).  Every workgroup works the 32 prefix sums
// out for itself (a few hundred instructions; no second launch, no scan library).
template <int BLOCK = kBlock>
__global__ __launch_bounds__(kBlock) void k_compact_runs(const lphq_run* __restrict__ runs, uint64_t arena_runs, const uint64_t* __restrict__ codes, uint64_t arena_codes,
                                                         const lphq_run_counts* __restrict__ counts, lphq_run* __restrict__ dense_runs, uint64_t cap_dense_runs,
                                                         uint64_t* __restrict__ dense_codes, uint64_t cap_dense_codes, unsigned long long* __restrict__ totals) {
    __shared__ uint64_t sm_rbase[LPHQ_RUN_ARENAS + 1], sm_xbase[LPHQ_RUN_ARENAS + 1];
    __shared__ uint32_t sm_over;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        uint64_t r = 0, x = 0;
        uint32_t over = 0;
#pragma unroll 1  // (unrolled, the 32 steps lived in registers all at once: 119 VGPRs, 68 scalar registers spilled)
        for (uint32_t a = 0; a < (uint32_t)LPHQ_RUN_ARENAS; ++a) {
            sm_rbase[a] = r;
            sm_xbase[a] = x;
            const uint64_t nr = counts[a].runs, nx = counts[a].codes;
            over |= (nr > arena_runs || nx > arena_codes) ? 1u : 0u;
            r += nr < arena_runs ? nr : arena_runs;
            x += nx < arena_codes ? nx : arena_codes;
        }
        sm_rbase[LPHQ_RUN_ARENAS] = r;
        sm_xbase[LPHQ_RUN_ARENAS] = x;
        sm_over = over | (r > cap_dense_runs || x > cap_dense_codes ? 1u : 0u);
    }
    __syncthreads();
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        totals[0] = sm_rbase[LPHQ_RUN_ARENAS];
        totals[1] = sm_xbase[LPHQ_RUN_ARENAS];
        totals[2] = sm_over;
    }
    if (sm_over) return;
    const uint32_t a = blockIdx.y;
    const uint64_t nr = sm_rbase[a + 1] - sm_rbase[a], nx = sm_xbase[a + 1] - sm_xbase[a];
    const lphq_run* const src = runs + (uint64_t)a * arena_runs;
    unsigned long long windows = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + tid; i < nr; i += (uint64_t)gridDim.x * kBlock) {
        lphq_run r = src[i];
        if (LPHQ_RUN_KIND(r) == (uint32_t)LPHQ_RUN_EXPLICIT) r.code += sm_xbase[a];
        dense_runs[sm_rbase[a] + i] = r;
        windows += LPHQ_RUN_LENGTH(r);
    }
    for (int o = 32; o > 0; o >>= 1) windows += __shfl_down(windows, o, 64);
    if ((tid & 63u) == 0u && windows) atomicAdd(&totals[3], windows);
    const uint64_t* const xsrc = codes + (uint64_t)a * arena_codes;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + tid; i < nx; i += (uint64_t)gridDim.x * kBlock) dense_codes[sm_xbase[a] + i] = xsrc[i];
}

}  // namespace lphq
