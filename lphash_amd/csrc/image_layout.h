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
 *   free slots      PTHash's Elias-Fano -> plain u32 table (one load instead of a select).
 *   quartet_wtree   three rank9 bit vectors (2 levels x {bits word, rank pair} = 4 lines)
 *                   -> one 64-B block per 192 minimizers: 4 cumulative type counts + the msb and
 *                   lsb bit planes.  rank_of() = one line + 6 masked popcounts.
 *   ef_sequence     Elias-Fano prefix sums (select + low bits = 3-4 lines, twice for NONE)
 *                   -> per-type 64-B "plane blocks": u64 absolute prefix + bit planes of the
 *                   small stored values; prefix/own value = one line + masked popcounts.
 *
 * All offsets are byte offsets from the image base; sections are 256-B aligned.
 */
#ifndef LPHQ_IMAGE_LAYOUT_H
#define LPHQ_IMAGE_LAYOUT_H

#include <stdint.h>

#define LPHQ_IMAGE_MAGIC 0x3130474d4951504cULL /* "LPQIMG01" */
#define LPHQ_WT_PER_BLOCK 192u                 /* minimizers per wavelet block              */
#define LPHQ_PAIR_PER_BLOCK 32u                /* entries per (size,pos) pair block         */
/* single-value plane blocks: P planes x T words, P*T <= 14 */
#define LPHQ_LR_T(P) ((P) == 4 ? 3u : 2u)
#define LPHQ_LR_PER_BLOCK(P) (32u * LPHQ_LR_T(P))

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
    uint64_t off_wt;         /* wt_block[]                                                  */
    uint64_t off_left, off_right, off_none; /* plane blocks (absolute bases already include W*n_maximal and region starts) */
    uint64_t off_upair;      /* unpartitioned: pair blocks indexed by minimizer order       */
    lphq_phf_desc mm, fb;
} lphq_image_header;

/* 64-B wavelet block: symbols j = 0..191 of minimizers [192*b, 192*b+192) */
typedef struct lphq_wt_block {
    uint32_t cnt[4];  /* # of LEFT, RIGHT_OR_COLLISION, MAXIMAL, NONE before this block     */
    uint32_t msb[6];  /* bit j%32 of msb[j/32] = root bit (1: MAXIMAL/NONE)                  */
    uint32_t lsb[6];  /* bit of the child vector (left_right or max_none)                   */
} lphq_wt_block;

/* 64-B plane block: abs = global-rank base of the first entry; planes of the per-entry values */
typedef struct lphq_plane_block {
    uint64_t abs;
    uint32_t w[14];
} lphq_plane_block;

#endif
