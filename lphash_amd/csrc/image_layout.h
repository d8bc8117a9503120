/* image_layout.h -- the flat, position-independent device image (shared by host and kernels).
 *
 * The DISK format is the reference's (lph_format.hpp).  The DEVICE layout is ours: every
 * structure the query chain touches is re-packed so that one lookup costs one cache line per
 * structure (measured on MI355X: scattered gathers are line-rate bound, ~55-68 G lines/s beyond
 * L2 regardless of 8..64 B width -- profiles/r01_microbench.txt):
 *
 *   PTHash pilots   ranks kept as the reference's compact codewords; the dictionary is replaced
 *                   by hp[code] = MurmurHash2_64(pilot, seed), saving one hash per lookup
 *                   (single_phf::position hashes the pilot on every call).
 *   free slots      PTHash's Elias-Fano -> plain u32 table (one load instead of a select); for the minimizer function of an LP-MPHF the
 *                   table is not read at all: tail[] below answers positions beyond num_keys directly.
 *   quartet_wtree + ef_sequence (partitioned)
 *                   three rank9 bit vectors (2 levels x {bits word, rank pair} = 4 lines) and the
 *                   Elias-Fano prefix sums (select + low bits = 3-4 lines, twice for NONE)
 *                   -> ONE 128-B block per 64 (k-m+1 <= 15) or 32 minimizers, indexed by the minimizer
 *                   order: the four per-type global-rank bases at the block start, the msb/lsb type
 *                   planes, and bit planes of the stored size / position values.  type, rank and
 *                   prefix sum of a lookup = one line + masked popcounts.  (Beyond L2 the fetch
 *                   granule is the 128-B line, profiles/r01_microbench2.txt, so 128 B cost what 16 B do.)
 *   ef_sequence x2 (unpartitioned)
 *                   -> 64-B "pair blocks": u64 prefix of sizes + size / position planes of 32 minimizers.
 *
 * All offsets are byte offsets from the image base; sections are 256-B aligned.
 */
#ifndef LPHQ_IMAGE_LAYOUT_H
#define LPHQ_IMAGE_LAYOUT_H

#include <stdint.h>

#define LPHQ_IMAGE_MAGIC 0x3130474d4951504cULL /* "LPQIMG01" */
#define LPHQ_PAIR_PER_BLOCK 32u                /* entries per (size,pos) pair block         */
/* fused block: T words per plane, 2 type planes + P size planes + P position planes, (2 + 2P) * T <= 24 words */
#define LPHQ_UNI_T(P) ((P) == 4 ? 2u : 1u)
#define LPHQ_UNI_PER_BLOCK(P) (32u * LPHQ_UNI_T(P))

/* 2-by-1 division by an invariant 32-bit divisor (Moeller & Granlund 2011, Alg. 4) */
typedef struct lphq_divisor {
    uint32_t d;     /* divisor (0 when the 32-bit path does not apply)            */
    uint32_t dn;    /* d << shift (normalised, top bit set)                      */
    uint32_t inv;   /* floor((2^64-1)/dn) - 2^32                                 */
    uint32_t shift; /* clz(d)                                                    */
    uint64_t d64;   /* divisor as 64-bit (generic path when d == 0)              */
} lphq_divisor;

typedef struct lphq_phf_desc {
    uint64_t seed, num_keys, table_size;
    uint64_t num_dense, num_sparse;
    uint64_t front_size;                      /* buckets served by the front dictionary         */
    uint64_t off_front_ranks, off_back_ranks; /* compact codewords (u64 words, 1 padding word) */
    uint64_t off_front_hp, off_back_hp;       /* u64 hp[code] = Murmur(pilot(code), seed)       */
    uint64_t off_free;                        /* u32 free_slots[table_size - num_keys]          */
    uint32_t front_width, back_width;
    uint32_t n_front_dict, n_back_dict;
    lphq_divisor div_table, div_dense, div_sparse;
} lphq_phf_desc;

typedef struct lphq_image_header {
    uint64_t magic, total_bytes;
    uint32_t kind, kmer_bits, k, m, W, P, reserved0, reserved1;
    uint64_t mm_seed, nkmers, distinct_minimizers, fallback_kmers;
    uint64_t n_maximal;
    uint64_t collision_base; /* partitioned: EF[none_pos_start] + W*n_maximal ; unpartitioned: num_kmers_in_main_index */
    uint64_t off_uni;        /* partitioned: lphq_uni_block[] indexed by minimizer order / LPHQ_UNI_PER_BLOCK(P) */
    uint64_t off_upair;      /* unpartitioned: pair blocks indexed by minimizer order       */
    uint64_t kmers_by_type[4]; /* k-mers under LEFT, RIGHT, MAXIMAL, NONE minimizers (census, for reporting) */
    lphq_phf_desc mm, fb;
    uint64_t off_tail;       /* kinds 0, 1: u64 tail[mm.table_size - mm.num_keys], see LPHQ_TAIL_* below */
} lphq_image_header;

/* Table positions p >= num_keys of the minimizer function (1 - alpha = 6 % of them): single_phf::position sends them through
 * free_slots to a minimizer order h < num_keys, whose block then gives the lookup its (base, mode).  tail[p - num_keys] holds that
 * (base, mode) itself -- what the block of order free_slots[p - num_keys] decodes to -- so such a lookup is pilot line -> tail line
 * instead of pilot line -> free-slot line -> block line: one dependent gather fewer in (almost) every wave, 0.06 line misses fewer
 * per lookup.  mode (0: base + p, 1: base - p, 2: collision) in the top two bits, base below. */
#define LPHQ_TAIL_MODE_SHIFT 62
#define LPHQ_TAIL_BASE_MASK ((1ULL << LPHQ_TAIL_MODE_SHIFT) - 1ULL)

/* 128-B fused block of minimizers [E*b, E*b+E), E = LPHQ_UNI_PER_BLOCK(P), T = LPHQ_UNI_T(P):
 *   base[0] = (k-m+1) * #MAXIMAL before the block                         (MAXIMAL: base + W*rank_in_block + p)
 *   base[1] = W*n_maximal + sum of LEFT values before the block           (EF[r] + W*n_maximal)
 *   base[2] = W*n_maximal + all LEFT + RIGHT sizes before the block       (pair(rcs+r).first + W*n_maximal)
 *   base[3] = W*n_maximal + all LEFT + all RIGHT + NONE sizes before      (EF[nss+r] + W*n_maximal)
 *   w[0..T) msb plane, w[T..2T) lsb plane (type = msb<<1|lsb, quartet_wtree.hpp:7),
 *   w[2T + q*T + t] bit plane q of the stored size (LEFT: p1+1, RIGHT: size or 0 = collision, NONE: size),
 *   w[2T + P*T + q*T + t] bit plane q of the NONE position p1. */
typedef struct lphq_uni_block {
    uint64_t base[4];
    uint32_t w[24];
} lphq_uni_block;

/* 64-B pair block of the unpartitioned function, 32 minimizers: abs = sum of the sizes before the block (sizes.pair().first),
 * w[q] = bit plane q of the size, w[P + q] = bit plane q of the position p1 */
typedef struct lphq_plane_block {
    uint64_t abs;
    uint32_t w[14];
} lphq_plane_block;

/* Competitor harness, BBHash half (image kind 3; bbhash.hpp): header fields reused -- n_maximal = number of levels, collision_base = number of keys
 * the bit arrays hold (the exact map's values are offset by it), fallback_kmers = entries of the exact map, off_uni = lphq_bbh_level[levels],
 * off_upair = the exact map as an open-addressing table of 2^reserved0 slots (linear probing from the top reserved0 bits of bbh_slot_hash): the keys
 * (kmer_bits / 64 words each) followed by the u64 values, ~0 marking an empty slot. */
/* (host side; the kernels carry the same four lines as bbh_slot_hash_dev in device_query.cuh) */
static inline uint64_t bbh_slot_hash(uint64_t lo, uint64_t hi) {
    uint64_t x = (lo ^ (hi * 0x9E3779B97F4A7C15ULL)) * 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 29;
    return x * 0x94D049BB133111EBULL;
}
typedef struct lphq_bbh_level {
    uint64_t hash_domain; /* bits of the level (a multiple of 64) */
    uint64_t off_bits;    /* u64 words, 1 + hash_domain / 64 of them */
    uint64_t off_ranks;   /* u64 ones before every 512-bit block, counted over all earlier levels too */
    uint64_t nwords;
} lphq_bbh_level;

#endif
