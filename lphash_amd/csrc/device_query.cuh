// device_query.cuh -- gfx950 device functions of the lookup chain over the flat image
// (image_layout.h).  Pure integer work, wave64; no MFMA.  Reference semantics cited per function
// (paths relative to the reference tree).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <gfx950/take_min.cuh>

#include "image_layout.h"

namespace lphq {

// MurmurHash2_64 (MurmurHash64A) of one 8-byte word; seedx = seed ^ (8 * m) hoisted by the caller.
// pthash::murmurhash2_64::hash(uint64_t, seed) -- partitioned_mphf.hpp:116, mphf_utils.hpp:128
#define LPHQ_MURMUR_M 0xc6a4a7935bd1e995ULL
__device__ __forceinline__ uint64_t murmur_seedx(uint64_t seed) { return seed ^ (8ull * LPHQ_MURMUR_M); }
__device__ __forceinline__ uint64_t murmur8(uint64_t v, uint64_t seedx) {
    uint64_t k = v * LPHQ_MURMUR_M;
    k ^= k >> 47;
    k *= LPHQ_MURMUR_M;
    uint64_t h = seedx ^ k;
    h *= LPHQ_MURMUR_M;
    h ^= h >> 47;
    h *= LPHQ_MURMUR_M;
    h ^= h >> 47;
    return h;
}

// MurmurHash2_64 of the 16 bytes of a 128-bit k-mer (low word first: little-endian __uint128_t), seedx16 = seed ^ (16 * m)
__device__ __forceinline__ uint64_t murmur_seedx16(uint64_t seed) { return seed ^ (16ull * LPHQ_MURMUR_M); }
__device__ __forceinline__ uint64_t murmur16(uint64_t lo, uint64_t hi, uint64_t seedx16) {
    uint64_t h = seedx16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uint64_t k = (i ? hi : lo) * LPHQ_MURMUR_M;
        k ^= k >> 47;
        k *= LPHQ_MURMUR_M;
        h ^= k;
        h *= LPHQ_MURMUR_M;
    }
    h ^= h >> 47;
    h *= LPHQ_MURMUR_M;
    h ^= h >> 47;
    return h;
}

// One 2-by-1 division step by a normalised invariant divisor (Moeller & Granlund 2011, Alg. 4):
// remainder of (r * 2^32 + u) / dn for r < dn.
__device__ __forceinline__ uint32_t div_step(uint32_t r, uint32_t u, uint32_t dn, uint32_t inv) {
    uint64_t q = (uint64_t)inv * r + (((uint64_t)r << 32) | u);
    uint32_t q1 = (uint32_t)(q >> 32) + 1u, q0 = (uint32_t)q;
    uint32_t rem = u - q1 * dn;
    if (rem > q0) rem += dn;
    if (rem >= dn) rem -= dn;
    return rem;
}
// Exact a % d for 0 < d < 2^32.  pthash fastmod::fastmod_u64(a, M, d) is the exact remainder for all 64-bit a, d; the
// 128-bit M of the file is replaced by a 32-bit reciprocal.  Every divisor of an attached image is in range: table sizes
// beyond 2^32 - 1 are rejected when the image is built (image.cpp) and when a device image is adopted (capi.cpp:
// header_is_consistent, which also requires non-zero bucket counts), and an empty function never divides (phf_position).
// (A generic 64-bit division path would cost ~450 instructions and ~15 VGPRs per inlined copy for a case that cannot occur.)
__device__ __forceinline__ uint64_t mod_u64(uint64_t a, uint32_t dn, uint32_t inv, uint32_t shift) {
    uint32_t u2 = (uint32_t)((a >> 32) >> (32 - shift));
    uint32_t u1 = (uint32_t)((a << shift) >> 32);
    uint32_t u0 = (uint32_t)(a << shift);
    uint32_t r = div_step(u2, u1, dn, inv);
    r = div_step(r, u0, dn, inv);
    return r >> shift;
}
__device__ __forceinline__ uint64_t mod_u64(uint64_t a, const lphq_divisor& dv) { return mod_u64(a, dv.dn, dv.inv, dv.shift); }

// pthash::single_phf<..., dictionary_dictionary, true>::position(hash)  (SURVEY Appendix A):
//   bucket = skew_bucketer(hash); pilot = dict[ranks[bucket]]; p = (hash ^ H(pilot)) mod table_size;
//   p < num_keys ? p : free_slots[p - num_keys]
// Device layout: hp[code] = H(pilot) precomputed; free_slots is a plain u32 table.
// The chain is three dependent gathers; it is written as issue / consume stages so that a caller can put other work
// between a stage's load and the next stage's use of it (phf_position() is the
// stages back to back).
struct PhfRank {       // where the codeword of a bucket sits
    uint32_t sh;       // bit offset of the codeword within the two dwords loaded by phf_load_code
    uint32_t width;    // codeword width
    bool front;        // front (dense-bucket) or back dictionary
};
__device__ __forceinline__ uint64_t phf_load_code(const uint8_t* __restrict__ img, const lphq_phf_desc& f, uint64_t x, PhfRank& rk) {
    const uint64_t T = 0x9999999999999800ULL;  // uint64_t(0.6 * UINT64_MAX), skew_bucketer
    const bool dense = x < T;
    // Both alternatives are read into (scalar) registers first and the VALUES are selected: `dense ? f.a : f.b` on
    // the fields themselves selects the ADDRESS, and with the descriptor in memory (tile kernels) that is a per-lane
    // vector load of a header field at the head of the gather chain.
    const lphq_divisor dd = f.div_dense, ds = f.div_sparse;
    // bucket numbers fit 32 bits (num_dense + num_sparse <= 2^32 - 1: image.cpp / header_is_consistent), so the arithmetic up to the bit offset does too
    uint32_t b = (uint32_t)mod_u64(x, dense ? dd.dn : ds.dn, dense ? dd.inv : ds.inv, dense ? dd.shift : ds.shift);
    if (!dense) b += (uint32_t)f.num_dense;
    const uint32_t fsz = (uint32_t)f.front_size;
    const bool front = b < fsz;
    const uint32_t bi = front ? b : b - fsz;
    const uint32_t fw = f.front_width, bw = f.back_width;
    const uint64_t fo = f.off_front_ranks, bo = f.off_back_ranks;
    const uint32_t width = front ? fw : bw;
    const uint32_t* ranks = (const uint32_t*)(img + (front ? fo : bo));
    const uint64_t bit = (uint64_t)bi * width;
    const uint64_t dw = bit >> 5;
    rk.sh = (uint32_t)bit & 31u;
    rk.width = width;
    rk.front = front;
    struct __attribute__((packed, aligned(4))) u64a4 { uint64_t v; };  // one 8-byte request at 4-byte alignment (gfx950 allows it)
    return ((const u64a4*)(ranks + dw))->v;
}
__device__ __forceinline__ uint64_t phf_load_hp(const uint8_t* __restrict__ img, const lphq_phf_desc& f, uint64_t two, const PhfRank& rk) {
    const uint32_t code = (uint32_t)(two >> rk.sh) & (rk.width >= 32 ? ~0u : ((1u << rk.width) - 1u));
    const uint64_t fo = f.off_front_hp, bo = f.off_back_hp;
    return ((const uint64_t*)(img + (rk.front ? fo : bo)))[code];
}
__device__ __forceinline__ uint64_t phf_finish(const uint8_t* __restrict__ img, const lphq_phf_desc& f, uint64_t x, uint64_t hp) {
    uint32_t p = (uint32_t)mod_u64(x ^ hp, f.div_table);  // table_size <= 2^32 - 1
    const uint32_t nk = (uint32_t)f.num_keys;
    if (p >= nk) p = ((const uint32_t*)(img + f.off_free))[p - nk];
    return p;
}
// hash128 hashers: the bucket comes from hash.first(), the table position from hash.second()
__device__ __forceinline__ uint64_t phf_position2(const uint8_t* __restrict__ img, const lphq_phf_desc& f, uint64_t first, uint64_t second) {
    if (f.num_keys == 0) return 0;  // empty function: unreachable for well-formed files
    PhfRank rk;
    const uint64_t two = phf_load_code(img, f, first, rk);
    return phf_finish(img, f, second, phf_load_hp(img, f, two, rk));
}
// hash64 hashers (LPHash's own functions): first() == second()
__device__ __forceinline__ uint64_t phf_position(const uint8_t* __restrict__ img, const lphq_phf_desc& f, uint64_t x) { return phf_position2(img, f, x, x); }

// mask of the first nb bits of a 32-bit word, nb in [0, 32]
__device__ __forceinline__ uint32_t low_mask(int nb) { return nb >= 32 ? ~0u : (nb <= 0 ? 0u : ((1u << nb) - 1u)); }

// quartet_wtree::rank_of (src/quartet_wtree.cpp:84-99) and ef_sequence::{access,pair,diff}
// (include/ef_sequence.hpp:77-99) fused over ONE 128-B block (image_layout.h: lphq_uni_block).
// Returns the type of minimizer order h, base = global-rank base of its super-k-mer (the `res.global_rank`
// terms of src/partitioned_mphf.cpp:297-336, already including (k-m+1)*n_maximal), size and position.
template <int P>
struct UniRegs {  // one fused block in registers
    static constexpr uint32_t T = LPHQ_UNI_T(P), E = LPHQ_UNI_PER_BLOCK(P), NWORDS = (2 + 2 * P) * T, NV = (NWORDS + 3) / 4;
    uint4 b01, b23;
    uint4 v[NV];
};
template <int P>
__device__ __forceinline__ void uni_load(const uint8_t* __restrict__ img, uint64_t off_uni, uint32_t h, UniRegs<P>& r) {
    const uint4* p = (const uint4*)(img + off_uni) + (size_t)(h / UniRegs<P>::E) * 8;
    r.b01 = p[0];
    r.b23 = p[1];
#pragma unroll
    for (uint32_t i = 0; i < UniRegs<P>::NV; ++i) r.v[i] = p[2 + i];
}
template <int P>
__device__ __forceinline__ void uni_decode(const UniRegs<P>& r, uint32_t W, uint32_t h, uint32_t& type, uint64_t& base, uint32_t& size, uint32_t& pos) {
    constexpr uint32_t T = UniRegs<P>::T, E = UniRegs<P>::E, NV = UniRegs<P>::NV;
    const uint32_t j = h % E;
    const uint4 b01 = r.b01, b23 = r.b23;
    uint32_t w[NV * 4];
#pragma unroll
    for (uint32_t i = 0; i < NV; ++i) {
        w[4 * i] = r.v[i].x;
        w[4 * i + 1] = r.v[i].y;
        w[4 * i + 2] = r.v[i].z;
        w[4 * i + 3] = r.v[i].w;
    }
    const uint32_t wj = j >> 5, bj = j & 31u;
    auto pick = [&](uint32_t first) {  // word wj of the plane starting at w[first]
        uint32_t r = w[first];
#pragma unroll
        for (uint32_t t = 1; t < T; ++t) r = (wj == t) ? w[first + t] : r;
        return r;
    };
    const uint32_t sm = (pick(0) >> bj) & 1u, sl = (pick(T) >> bj) & 1u;
    const uint32_t xm = sm ? 0u : ~0u, xl = sl ? 0u : ~0u;
    uint32_t mw[T];  // entries before j with the same type
    uint32_t rank = 0;
#pragma unroll
    for (uint32_t t = 0; t < T; ++t) {
        mw[t] = (w[t] ^ xm) & (w[T + t] ^ xl) & low_mask((int)j - 32 * (int)t);
        rank += __popc(mw[t]);
    }
    uint32_t sum = 0;
    size = 0;
    pos = 0;
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)P; ++q) {
        uint32_t c = 0;
#pragma unroll
        for (uint32_t t = 0; t < T; ++t) c += __popc(w[2 * T + q * T + t] & mw[t]);
        sum += c << q;
        size |= ((pick(2 * T + q * T) >> bj) & 1u) << q;
        pos |= ((pick(2 * T + P * T + q * T) >> bj) & 1u) << q;
    }
    type = (sm << 1) | sl;
    const uint64_t base0 = ((uint64_t)b01.y << 32) | b01.x, base1 = ((uint64_t)b01.w << 32) | b01.z;
    const uint64_t base2 = ((uint64_t)b23.y << 32) | b23.x, base3 = ((uint64_t)b23.w << 32) | b23.z;
    const uint64_t blo = sl ? base2 : base1, bhi = sl ? base3 : base0;  // type 0:LEFT->1, 1:RIGHT->2, 2:MAXIMAL->0, 3:NONE->3
    base = (sm ? bhi : blo) + (type == 2u ? W * rank : sum);
}
template <int P>
__device__ __forceinline__ void uni_lookup(const uint8_t* __restrict__ img, uint64_t off_uni, uint32_t W, uint32_t h, uint32_t& type, uint64_t& base,
                                           uint32_t& size, uint32_t& pos) {
    UniRegs<P> r;
    uni_load<P>(img, off_uni, h, r);
    uni_decode<P>(r, W, h, type, base, size, pos);
}

// (size, position) pair block, 32 entries: NONE minimizers (partitioned_mphf.cpp:326-333) and the whole
// unpartitioned index (sizes.pair + positions.diff, unpartitioned_mphf.cpp:194-204).
struct PairRegs {
    uint4 v0, v1, v2, v3;
};
__device__ __forceinline__ void pair_load(const uint8_t* __restrict__ img, uint64_t off, uint32_t r, PairRegs& g) {
    const uint4* p = (const uint4*)(img + off) + (size_t)(r / LPHQ_PAIR_PER_BLOCK) * 4;
    g.v0 = p[0];
    g.v1 = p[1];
    g.v2 = p[2];
    g.v3 = p[3];
}
template <int P>
__device__ __forceinline__ void pair_decode(const PairRegs& g, uint32_t r, uint64_t& prefix, uint32_t& size, uint32_t& pos) {
    const uint32_t j = r & (LPHQ_PAIR_PER_BLOCK - 1u);
    const uint4 v0 = g.v0, v1 = g.v1, v2 = g.v2, v3 = g.v3;
    const uint32_t w[14] = {v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    const uint32_t msk = (1u << j) - 1u;
    uint32_t sum = 0;
    size = 0;
    pos = 0;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        sum += __popc(w[q] & msk) << q;
        size |= ((w[q] >> j) & 1u) << q;
        pos |= ((w[P + q] >> j) & 1u) << q;
    }
    prefix = (((uint64_t)v0.y << 32) | v0.x) + sum;
}
template <int P>
__device__ __forceinline__ void pair_lookup(const uint8_t* __restrict__ img, uint64_t off, uint32_t r, uint64_t& prefix, uint32_t& size, uint32_t& pos) {
    PairRegs g;
    pair_load(img, off, r, g);
    pair_decode<P>(g, r, prefix, size, pos);
}

// fallback_hasher::hash(kmer_t, seed)   include/constants.hpp:56-70
template <bool K128>
__device__ __forceinline__ uint64_t fallback_hash(uint64_t lo, uint64_t hi, uint64_t seed) {
    if (K128) return murmur8(lo, murmur_seedx(seed)) ^ murmur8(hi, murmur_seedx(~seed));
    return murmur8(lo, murmur_seedx(seed));
}

// debug::compute_minimizer_triplet   include/mphf_utils.hpp:118-137: scan the W m-mers from the right end,
// `<=` keeps the LEFTMOST minimum; returns the m-mer and p = offset of the minimizer from the left.
// Same result, scanned from the LEFT end with a strict `<` (the first of equal hashes stays): the k-mer is kept left-aligned, so the m-mer is
// the top 2m bits (one shift, no mask), and hash and index are updated under ONE compare; the m-mer itself is cut out again afterwards.
// ~31 instructions per m-mer instead of ~38 (k=63: 44 m-mers per lookup, and that kernel is bound by VALU issue, profiles/r02_c4_pmc_summary.txt).
template <bool K128>
__device__ __forceinline__ void minimizer_of(uint64_t lo, uint64_t hi, uint32_t k, uint32_t m, uint32_t W, uint64_t mm_seedx, uint64_t& mm, uint32_t& p) {
    const uint32_t top = 64u - 2u * m;  // m <= 32
    uint64_t min_hash = ~0ull;          // (a first hash of ~0 leaves j_min = 0, which is what it should be)
    uint32_t j_min = 0;
    auto step = [&](uint64_t v, uint32_t j) { take_min_uniform(min_hash, j_min, murmur8(v, mm_seedx), j); };  // (j is uniform over the wave)
    if (K128) {
        // The k-mer left-aligned in (h:l).  The m-mers are cut from a 64-bit window `cur` of that stream which is renewed every T = 33 - m
        // steps (a window holds T m-mers: 2m + 2(T-1) <= 64); inside a group an m-mer is two shifts by uniform amounts of the same register
        // pair -- no 128-bit shift per step and no chain through the k-mer (k=63, m=20: 2.3 instead of 4.5 instructions per m-mer).
        const uint32_t a = 128u - 2u * k;  // 2 .. 126
        const uint64_t h = a >= 64u ? lo << (a - 64u) : ((hi << a) | (lo >> (64u - a)));
        const uint64_t l = a >= 64u ? 0ull : lo << a;
        const uint32_t T = 33u - m;
        for (uint32_t j0 = 0; j0 < W; j0 += T) {  // (uniform)
            const uint32_t s = 2u * j0;            // < 126
            const uint64_t cur = s == 0u ? h : (s < 64u ? ((h << s) | (l >> (64u - s))) : (l << (s - 64u)));
            const uint32_t n = min(T, W - j0);
            for (uint32_t t = 0; t < n; ++t) step((cur << (2u * t)) >> top, j0 + t);
        }
    } else {
        uint64_t l = lo << (64u - 2u * k);  // k <= 31: the k-mer left-aligned, the m-mer is its top 2m bits (one shift, no mask)
        for (uint32_t j = 0; j < W; ++j) {
            step(l >> top, j);
            l <<= 2;
        }
    }
    const uint32_t s = 2u * (k - m - j_min);  // the minimizer starts j_min bases from the left: bits [s, s + 2m) of the k-mer
    const uint64_t mask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
    uint64_t v;
    if (K128) v = s == 0u ? lo : (s < 64u ? ((lo >> s) | (hi << (64u - s))) : (hi >> (s - 64u)));
    else v = lo >> s;
    mm = v & mask;
    p = j_min;  // == k - (pos + m) with pos counted from the right end, mphf_utils.hpp:136
}

// mphf::query (src/partitioned_mphf.cpp:292-339) / mphf_alt::query (src/unpartitioned_mphf.cpp:191-206):
// hash code of k-mer (lo,hi) whose minimizer is mm at offset p.  All arithmetic is u64 with wraparound,
// as in the reference, so non-member k-mers produce the same (meaningless) codes.
struct LookupCtx {   // what a streaming follower needs from its super-k-mer leader
    uint64_t base;   // hval = base + sign * p  (collision: base = collision_base, hval = base + fallback(kmer))
    uint32_t mode;   // 0: base + p, 1: base - p, 2: collision
};
// the type-dependent closed forms, from the decoded block entry
__device__ __forceinline__ LookupCtx ctx_partitioned(const lphq_image_header& H, uint32_t type, uint64_t base, uint32_t size, uint32_t pos) {
    LookupCtx c;
    if (type == 2u || type == 0u) {  // MAXIMAL: (k-m+1)*rank + p ; LEFT: EF[r] + W*n_max + p
        c.base = base;
        c.mode = 0;
    } else if (type == 3u) {  // NONE: EF[none_sizes_start+r] + W*n_max + (p1 - p)
        c.base = base + pos;
        c.mode = 1;
    } else if (size == 0u) {  // collision: EF[none_pos_start] + W*n_max + fallback(kmer)
        c.base = H.collision_base;
        c.mode = 2;
    } else {  // RIGHT: val1 + W*n_max + (k - m - p)
        c.base = base + (H.k - H.m);
        c.mode = 1;
    }
    return c;
}
__device__ __forceinline__ LookupCtx ctx_unpartitioned(const lphq_image_header& H, uint64_t prefix, uint32_t size, uint32_t pos) {
    LookupCtx c;
    if (size == 0u) {  // num_kmers_in_main_index + fallback(kmer)
        c.base = H.collision_base;
        c.mode = 2;
    } else {  // val1 + p1 - p
        c.base = prefix + pos;
        c.mode = 1;
    }
    return c;
}
template <int KIND, bool K128, int P>
__device__ __forceinline__ LookupCtx lookup_ctx(const uint8_t* __restrict__ img, const lphq_image_header& H, uint64_t mm) {
    // single_phf::position up to the table position p; positions beyond num_keys (1 - alpha of them) are answered by the tail table
    // (image_layout.h) instead of free_slots -> block: its load is issued next to the block load of the other lanes, so no wave
    // waits for a third dependent gather.  (Tail lanes read block 0, a line every wave touches, and discard it.)
    const lphq_phf_desc& f = H.mm;
    const uint64_t x = murmur8(mm, murmur_seedx(f.seed));
    uint32_t p = 0;
    if (f.num_keys != 0) {  // (uniform; an empty function answers 0, as phf_position2)
        PhfRank rk;
        const uint64_t two = phf_load_code(img, f, x, rk);
        p = (uint32_t)mod_u64(x ^ phf_load_hp(img, f, two, rk), f.div_table);
    }
    const uint32_t nk = (uint32_t)f.num_keys;
    const bool tail = p >= nk && nk != 0u;
    uint64_t tv = 0;
    if (tail) tv = ((const uint64_t*)(img + H.off_tail))[p - nk];
    const uint32_t h = tail ? 0u : p;
    LookupCtx c;
    if (KIND == 0) {
        uint32_t type, size, pos;
        uint64_t base;
        uni_lookup<P>(img, H.off_uni, H.W, h, type, base, size, pos);
        c = ctx_partitioned(H, type, base, size, pos);
    } else {
        uint64_t prefix;
        uint32_t size, pos;
        pair_lookup<P>(img, H.off_upair, h, prefix, size, pos);
        c = ctx_unpartitioned(H, prefix, size, pos);
    }
    if (tail) {
        c.base = tv & LPHQ_TAIL_BASE_MASK;
        c.mode = (uint32_t)(tv >> LPHQ_TAIL_MODE_SHIFT);
    }
    return c;
}
template <bool K128>
__device__ __forceinline__ uint64_t finish_hval(const uint8_t* __restrict__ img, const lphq_image_header& H, const LookupCtx& c, uint64_t lo, uint64_t hi, uint32_t p) {
    if (c.mode == 2u) {
        // (rare) the fallback function's descriptor is read from the image, whatever copy of the header the caller holds: the random kernel takes the header
        // by value, and 30 more scalar fields alive across its hot path were spilled to VGPR lanes and read back (v_readlane) by every wave
        const lphq_phf_desc& fb = ((const lphq_image_header*)img)->fb;
        return c.base + phf_position(img, fb, fallback_hash<K128>(lo, hi, fb.seed));
    }
    return c.mode == 0u ? c.base + p : c.base - p;
}

// Competitor harness (KIND 2): pthash_mphf(kmer) of src/ptbb/ptbb_query.cpp:47 with ptbb::PTHasher (src/ptbb/ptbb.hpp:16-23):
// hash128{Murmur(&kmer, sizeof kmer, seed), Murmur(&kmer, sizeof kmer, ~seed)}
template <bool K128>
__device__ __forceinline__ uint64_t pthash_kmer(const uint8_t* __restrict__ img, const lphq_image_header& H, uint64_t lo, uint64_t hi) {
    const uint64_t seed = H.mm.seed;
    const uint64_t a = K128 ? murmur16(lo, hi, murmur_seedx16(seed)) : murmur8(lo, murmur_seedx(seed));
    const uint64_t b = K128 ? murmur16(lo, hi, murmur_seedx16(~seed)) : murmur8(lo, murmur_seedx(~seed));
    return phf_position2(img, H.mm, a, b);
}

// Competitor harness (KIND 3): boomphf::mphf<kmer_t, ptbb::BBHasher>::lookup (external/BooPHF.hpp:791-822) over the flat image of bbhash.hpp.
// Level i probes bit fastrange64(h_i, domain_i) of its bit array, h_0 / h_1 = BBHasher with the two fixed seeds, h_i (i >= 2) the xorshift
// sequence seeded by them (getLevel :1059-1088, XorshiftHashFunctors :405-423); the first level whose bit is set answers with the rank of that
// bit (ranks sampled every 512 bits, bitVector::rank :616-626); a key that no level holds is looked up in the exact map, ULLONG_MAX if absent.
// BBHasher (src/ptbb/ptbb.hpp:25-32) = Murmur(key, seed) ^ Murmur(key, ~seed).
__device__ __forceinline__ uint64_t bbh_slot_hash_dev(uint64_t lo, uint64_t hi) {  // == bbh_slot_hash of image_layout.h (the host fills the table with it)
    uint64_t x = (lo ^ (hi * 0x9E3779B97F4A7C15ULL)) * 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 29;
    return x * 0x94D049BB133111EBULL;
}
template <bool K128>
__device__ __forceinline__ uint64_t bbhash_kmer(const uint8_t* __restrict__ img, const lphq_image_header& H, uint64_t lo, uint64_t hi) {
    auto hasher = [&](uint64_t seed) {
        return K128 ? (murmur16(lo, hi, murmur_seedx16(seed)) ^ murmur16(lo, hi, murmur_seedx16(~seed)))
                    : (murmur8(lo, murmur_seedx(seed)) ^ murmur8(lo, murmur_seedx(~seed)));
    };
    const lphq_bbh_level* lv = (const lphq_bbh_level*)(img + H.off_uni);
    const uint32_t nl = (uint32_t)H.n_maximal;
    uint64_t s0 = 0, s1 = 0;
    for (uint32_t ii = 0; ii + 1 < nl; ++ii) {
        uint64_t hsh;
        if (ii == 0) hsh = s0 = hasher(0xAAAAAAAA55555555ULL);
        else if (ii == 1) hsh = s1 = hasher(0x33333333CCCCCCCCULL);
        else {
            uint64_t t1 = s0;
            const uint64_t t0 = s1;
            s0 = t0;
            t1 ^= t1 << 23;
            s1 = t1 ^ t0 ^ (t1 >> 17) ^ (t0 >> 26);
            hsh = s1 + t0;
        }
        const uint64_t pos = __umul64hi(hsh, lv[ii].hash_domain);  // fastrange64
        const uint64_t* bits = (const uint64_t*)(img + lv[ii].off_bits);
        const uint64_t wi = pos >> 6, word = bits[wi];
        if ((word >> (pos & 63)) & 1ull) {
            uint64_t r = ((const uint64_t*)(img + lv[ii].off_ranks))[pos >> 9];
            for (uint64_t w = (pos >> 9) << 3; w < wi; ++w) r += (uint64_t)__popcll(bits[w]);
            return r + (uint64_t)__popcll(word & ((1ull << (pos & 63)) - 1ull));
        }
    }
    // exact map of the keys that reached the last level: open addressing, linear probing (image_layout.h)
    const uint32_t lg = H.reserved0;
    const uint64_t cap = 1ull << lg;
    const uint64_t* keys = (const uint64_t*)(img + H.off_upair);
    const uint64_t* vals = keys + cap * (K128 ? 2 : 1);
    // (at most cap probes: a well-formed table is at most half full, but an adopted image -- lphq_open_device_image -- whose table had no empty
    // slot must not spin a non-member's lane forever)
    uint64_t slot = bbh_slot_hash_dev(lo, hi) >> (64 - lg);
    for (uint64_t probes = 0; probes < cap; ++probes, slot = (slot + 1) & (cap - 1)) {
        const uint64_t v = vals[slot];
        if (v == ~0ull) break;  // empty slot: not a key
        if (K128 ? (keys[2 * slot] == lo && keys[2 * slot + 1] == hi) : keys[slot] == lo) return v + H.collision_base;
    }
    return ~0ull;  // ULLONG_MAX: not a key
}

template <int KIND, bool K128, int P>
__device__ __forceinline__ uint64_t query_kmer(const uint8_t* __restrict__ img, const lphq_image_header& H, uint64_t lo, uint64_t hi) {
    if (KIND == 2) return pthash_kmer<K128>(img, H, lo, hi);
    if (KIND == 3) return bbhash_kmer<K128>(img, H, lo, hi);
    uint64_t mm;
    uint32_t p;
    minimizer_of<K128>(lo, hi, H.k, H.m, H.W, murmur_seedx(H.mm_seed), mm, p);
    const LookupCtx c = lookup_ctx<KIND, K128, P>(img, H, mm);
    return finish_hval<K128>(img, H, c, lo, hi, p);
}

}  // namespace lphq
