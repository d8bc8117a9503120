// device_query.cuh -- gfx950 device functions of the lookup chain over the flat image
// (image_layout.h).  Pure integer work, wave64; no MFMA.  Reference semantics cited per function
// (paths relative to the reference tree).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
// Exact a % d.  pthash fastmod::fastmod_u64(a, M, d) is the exact remainder for all 64-bit a, d; the
// 128-bit M of the file is replaced by a 32-bit reciprocal when d < 2^32 (always, for < 4 G keys).
__device__ __forceinline__ uint64_t mod_u64(uint64_t a, uint32_t d, uint32_t dn, uint32_t inv, uint32_t shift, uint64_t d64) {
    if (d) {
        uint32_t u2 = (uint32_t)((a >> 32) >> (32 - shift));
        uint32_t u1 = (uint32_t)((a << shift) >> 32);
        uint32_t u0 = (uint32_t)(a << shift);
        uint32_t r = div_step(u2, u1, dn, inv);
        r = div_step(r, u0, dn, inv);
        return r >> shift;
    }
    return d64 ? a % d64 : 0;
}
__device__ __forceinline__ uint64_t mod_u64(uint64_t a, const lphq_divisor& dv) { return mod_u64(a, dv.d, dv.dn, dv.inv, dv.shift, dv.d64); }

// pthash::single_phf<..., dictionary_dictionary, true>::position(hash)  (SURVEY Appendix A):
//   bucket = skew_bucketer(hash); pilot = dict[ranks[bucket]]; p = (hash ^ H(pilot)) mod table_size;
//   p < num_keys ? p : free_slots[p - num_keys]
// Device layout: hp[code] = H(pilot) precomputed; free_slots is a plain u32 table.
__device__ __forceinline__ uint64_t phf_position(const uint8_t* __restrict__ img, const lphq_phf_desc& f, uint64_t x) {
    if (f.num_keys == 0) return 0;  // empty function: unreachable for well-formed files
    const uint64_t T = 0x9999999999999800ULL;  // uint64_t(0.6 * UINT64_MAX), skew_bucketer
    const bool dense = x < T;
    uint64_t b = mod_u64(x, dense ? f.div_dense.d : f.div_sparse.d, dense ? f.div_dense.dn : f.div_sparse.dn,
                         dense ? f.div_dense.inv : f.div_sparse.inv, dense ? f.div_dense.shift : f.div_sparse.shift,
                         dense ? f.div_dense.d64 : f.div_sparse.d64);
    if (!dense) b += f.num_dense;
    const bool front = b < f.front_size;
    const uint64_t bi = front ? b : b - f.front_size;
    const uint32_t width = front ? f.front_width : f.back_width;
    const uint32_t* ranks = (const uint32_t*)(img + (front ? f.off_front_ranks : f.off_back_ranks));
    const uint64_t bit = bi * width;
    const uint64_t dw = bit >> 5;
    const uint32_t sh = (uint32_t)bit & 31u;
    const uint32_t lo = ranks[dw], hi = ranks[dw + 1];
    const uint32_t code = (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & (width >= 32 ? ~0u : ((1u << width) - 1u));
    const uint64_t hp = ((const uint64_t*)(img + (front ? f.off_front_hp : f.off_back_hp)))[code];
    uint64_t p = mod_u64(x ^ hp, f.div_table);
    if (p >= f.num_keys) p = ((const uint32_t*)(img + f.off_free))[p - f.num_keys];
    return p;
}

// mask of the first nb bits of a 32-bit word, nb in [0, 32]
__device__ __forceinline__ uint32_t low_mask(int nb) { return nb >= 32 ? ~0u : (nb <= 0 ? 0u : ((1u << nb) - 1u)); }

// quartet_wtree::rank_of(idx) -> (type, rank of that type before idx)   src/quartet_wtree.cpp:84-99
// over one 64-B block: 4 cumulative counts + msb/lsb planes of 192 symbols.
__device__ __forceinline__ void wt_rank_of(const uint8_t* __restrict__ img, uint64_t off_wt, uint32_t h, uint32_t& type, uint32_t& rank) {
    const uint32_t b = h / LPHQ_WT_PER_BLOCK, j = h - b * LPHQ_WT_PER_BLOCK;
    const uint4* p = (const uint4*)(img + off_wt) + (size_t)b * 4;
    const uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
    const uint32_t M[6] = {v1.x, v1.y, v1.z, v1.w, v2.x, v2.y};
    const uint32_t L[6] = {v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    const uint32_t wj = j >> 5, bj = j & 31u;
    uint32_t mw = 0, lw = 0;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        mw = (wj == (uint32_t)t) ? M[t] : mw;
        lw = (wj == (uint32_t)t) ? L[t] : lw;
    }
    const uint32_t sm = (mw >> bj) & 1u, sl = (lw >> bj) & 1u;
    const uint32_t xm = sm ? 0u : ~0u, xl = sl ? 0u : ~0u;
    uint32_t r = 0;
#pragma unroll
    for (int t = 0; t < 6; ++t) r += __popc((M[t] ^ xm) & (L[t] ^ xl) & low_mask((int)j - 32 * t));
    type = (sm << 1) | sl;
    const uint32_t c01 = sl ? v0.y : v0.x, c23 = sl ? v0.w : v0.z;
    rank = (sm ? c23 : c01) + r;
}

// Single-value plane block (LEFT positions / RIGHT_OR_COLLISION sizes): prefix = global-rank base of
// entry r (ef_sequence::access / pair().first + (k-m+1)*n_maximal, partitioned_mphf.cpp:299-318),
// own = stored value r (pair().second - pair().first).
template <int P>
__device__ __forceinline__ void lr_lookup(const uint8_t* __restrict__ img, uint64_t off, uint32_t r, uint64_t& prefix, uint32_t& own) {
    constexpr uint32_t T = LPHQ_LR_T(P), E = LPHQ_LR_PER_BLOCK(P);
    const uint32_t b = r / E, j = r - b * E;
    const uint4* p = (const uint4*)(img + off) + (size_t)b * 4;
    const uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
    const uint32_t w[14] = {v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    const uint32_t wj = j >> 5, bj = j & 31u;
    uint32_t msk[T];
#pragma unroll
    for (uint32_t t = 0; t < T; ++t) msk[t] = low_mask((int)j - 32 * (int)t);
    uint32_t sum = 0;
    own = 0;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        uint32_t c = 0, sel = 0;
#pragma unroll
        for (uint32_t t = 0; t < T; ++t) {
            c += __popc(w[q * T + t] & msk[t]);
            sel = (wj == t) ? w[q * T + t] : sel;
        }
        sum += c << q;
        own |= ((sel >> bj) & 1u) << q;
    }
    prefix = (((uint64_t)v0.y << 32) | v0.x) + sum;
}

// (size, position) pair block, 32 entries: NONE minimizers (partitioned_mphf.cpp:326-333) and the whole
// unpartitioned index (sizes.pair + positions.diff, unpartitioned_mphf.cpp:194-204).
template <int P>
__device__ __forceinline__ void pair_lookup(const uint8_t* __restrict__ img, uint64_t off, uint32_t r, uint64_t& prefix, uint32_t& size, uint32_t& pos) {
    const uint32_t b = r / LPHQ_PAIR_PER_BLOCK, j = r & (LPHQ_PAIR_PER_BLOCK - 1u);
    const uint4* p = (const uint4*)(img + off) + (size_t)b * 4;
    const uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
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

// fallback_hasher::hash(kmer_t, seed)   include/constants.hpp:56-70
template <bool K128>
__device__ __forceinline__ uint64_t fallback_hash(uint64_t lo, uint64_t hi, uint64_t seed) {
    if (K128) return murmur8(lo, murmur_seedx(seed)) ^ murmur8(hi, murmur_seedx(~seed));
    return murmur8(lo, murmur_seedx(seed));
}

// debug::compute_minimizer_triplet   include/mphf_utils.hpp:118-137: scan the W m-mers from the right end,
// `<=` keeps the LEFTMOST minimum; returns the m-mer and p = offset of the minimizer from the left.
template <bool K128>
__device__ __forceinline__ void minimizer_of(uint64_t lo, uint64_t hi, uint32_t k, uint32_t m, uint32_t W, uint64_t mm_seedx, uint64_t& mm, uint32_t& p) {
    const uint64_t mask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
    uint64_t min_hash = ~0ull;
    uint32_t pos = 0;
    mm = 0;
#pragma unroll 2
    for (uint32_t i = 0; i < W; ++i) {
        const uint64_t mmer = lo & mask;
        const uint64_t hsh = murmur8(mmer, mm_seedx);
        if (hsh <= min_hash) {
            min_hash = hsh;
            mm = mmer;
            pos = i;
        }
        lo >>= 2;
        if (K128) {
            lo |= hi << 62;
            hi >>= 2;
        }
    }
    p = k - (pos + m);
}

// mphf::query (src/partitioned_mphf.cpp:292-339) / mphf_alt::query (src/unpartitioned_mphf.cpp:191-206):
// hash code of k-mer (lo,hi) whose minimizer is mm at offset p.  All arithmetic is u64 with wraparound,
// as in the reference, so non-member k-mers produce the same (meaningless) codes.
struct LookupCtx {   // what a streaming follower needs from its super-k-mer leader
    uint64_t base;   // hval = base + sign * p  (collision: base = collision_base, hval = base + fallback(kmer))
    uint32_t mode;   // 0: base + p, 1: base - p, 2: collision
};
template <int KIND, bool K128, int P>
__device__ __forceinline__ LookupCtx lookup_ctx(const uint8_t* __restrict__ img, const lphq_image_header& H, uint64_t mm) {
    LookupCtx c;
    const uint32_t h = (uint32_t)phf_position(img, H.mm, murmur8(mm, murmur_seedx(H.mm.seed)));
    if (KIND == 0) {
        uint32_t type, rank;
        wt_rank_of(img, H.off_wt, h, type, rank);
        if (type == 2u) {  // MAXIMAL: (k-m+1)*rank + p
            c.base = (uint64_t)H.W * rank;
            c.mode = 0;
        } else if (type == 3u) {  // NONE: EF[none_sizes_start+r] + W*n_max + (p1 - p)
            uint64_t prefix;
            uint32_t size, pos;
            pair_lookup<P>(img, H.off_none, rank, prefix, size, pos);
            c.base = prefix + pos;
            c.mode = 1;
        } else {  // LEFT / RIGHT_OR_COLLISION share the block format
            uint64_t prefix;
            uint32_t own;
            lr_lookup<P>(img, type ? H.off_right : H.off_left, rank, prefix, own);
            if (type == 0u) {  // LEFT: EF[r] + W*n_max + p
                c.base = prefix;
                c.mode = 0;
            } else if (own == 0u) {  // collision: EF[none_pos_start] + W*n_max + fallback(kmer)
                c.base = H.collision_base;
                c.mode = 2;
            } else {  // RIGHT: val1 + W*n_max + (k - m - p)
                c.base = prefix + (H.k - H.m);
                c.mode = 1;
            }
        }
    } else {
        uint64_t prefix;
        uint32_t size, pos;
        pair_lookup<P>(img, H.off_upair, h, prefix, size, pos);
        if (size == 0u) {  // num_kmers_in_main_index + fallback(kmer)
            c.base = H.collision_base;
            c.mode = 2;
        } else {  // val1 + p1 - p
            c.base = prefix + pos;
            c.mode = 1;
        }
    }
    return c;
}
template <bool K128>
__device__ __forceinline__ uint64_t finish_hval(const uint8_t* __restrict__ img, const lphq_image_header& H, const LookupCtx& c, uint64_t lo, uint64_t hi, uint32_t p) {
    if (c.mode == 2u) return c.base + phf_position(img, H.fb, fallback_hash<K128>(lo, hi, H.fb.seed));
    return c.mode == 0u ? c.base + p : c.base - p;
}

template <int KIND, bool K128, int P>
__device__ __forceinline__ uint64_t query_kmer(const uint8_t* __restrict__ img, const lphq_image_header& H, uint64_t lo, uint64_t hi) {
    uint64_t mm;
    uint32_t p;
    minimizer_of<K128>(lo, hi, H.k, H.m, H.W, murmur_seedx(H.mm_seed), mm, p);
    const LookupCtx c = lookup_ctx<KIND, K128, P>(img, H, mm);
    return finish_hval<K128>(img, H, c, lo, hi, p);
}

}  // namespace lphq
