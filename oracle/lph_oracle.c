/*
 * lph_oracle.c -- TEST INFRASTRUCTURE ONLY (see lph_oracle.h).
 *
 * Plain-C, single-threaded restatement of the LPHash `query-p` / `query-u` path and of the
 * build path that writes the `.lph` files.  Each function cites the reference file:line it
 * follows (paths relative to /root/reference).  Structures are kept in their SERIALIZED
 * shape (rank9 pairs, darray inventories, Elias-Fano high/low) and walked exactly the way the
 * reference walks them -- on purpose different from the product's re-packed device image, so
 * that oracle-vs-product agreement is a real check.
 *
 * PARITY UNPINNED: PTHash (external/pthash, empty submodule) is restated from its published
 * v1 algorithm (Pibiri & Trani, SIGIR'21; jermp/pthash v1 include/{single_phf,utils/bucketers,
 * encoders/...,fastmod/fastmod}.hpp as recalled in SURVEY.md Appendix A); there are no golden
 * vectors in the reference to pin absolute hash codes.
 */
#include "lph_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

static __thread char g_err[512];
const char* orc_last_error(void) { return g_err; }
#define FAIL(...)                                  \
    do {                                           \
        snprintf(g_err, sizeof g_err, __VA_ARGS__); \
    } while (0)

/* ------------------------------------------------------------------------------------------
 * PTHash hashing:  MurmurHash2_64 == public MurmurHash64A (Austin Appleby), SURVEY Appendix A.
 * Call sites: include/partitioned_mphf.hpp:116, include/mphf_utils.hpp:128, include/constants.hpp:56-70
 * ---------------------------------------------------------------------------------------- */
uint64_t orc_murmur64a(const void* key, size_t len, uint64_t seed) {
    const uint64_t m = 0xc6a4a7935bd1e995ULL;
    const int r = 47;
    uint64_t h = seed ^ (len * m);
    const uint8_t* data = (const uint8_t*)key;
    const uint8_t* end = data + (len / 8) * 8;
    while (data != end) {
        uint64_t k;
        memcpy(&k, data, 8);
        data += 8;
        k *= m;
        k ^= k >> r;
        k *= m;
        h ^= k;
        h *= m;
    }
    switch (len & 7) {
        case 7: h ^= (uint64_t)data[6] << 48; /* fallthrough */
        case 6: h ^= (uint64_t)data[5] << 40; /* fallthrough */
        case 5: h ^= (uint64_t)data[4] << 32; /* fallthrough */
        case 4: h ^= (uint64_t)data[3] << 24; /* fallthrough */
        case 3: h ^= (uint64_t)data[2] << 16; /* fallthrough */
        case 2: h ^= (uint64_t)data[1] << 8;  /* fallthrough */
        case 1: h ^= (uint64_t)data[0]; h *= m;
    }
    h ^= h >> r;
    h *= m;
    h ^= h >> r;
    return h;
}

/* pthash::murmurhash2_64::hash(uint64_t val, seed) ; pthash::default_hash64(val, seed) */
static inline uint64_t hash_u64(uint64_t v, uint64_t seed) { return orc_murmur64a(&v, 8, seed); }

/* fallback_hasher::hash(kmer_t, seed)   include/constants.hpp:56-70 */
static inline uint64_t fallback_hash(u128 kmer, int kmer_bits, uint64_t seed) {
    if (kmer_bits == 64) {
        uint64_t v = (uint64_t)kmer;
        return orc_murmur64a(&v, 8, seed);
    }
    uint64_t low = (uint64_t)kmer, high = (uint64_t)(kmer >> 64);
    return orc_murmur64a(&low, 8, seed) ^ orc_murmur64a(&high, 8, ~seed);
}

/* ------------------------------------------------------------------------------------------
 * pthash fastmod (Lemire), util::{popcount,msb,lsb,select_in_word}
 * ---------------------------------------------------------------------------------------- */
static u128 computeM_u64(uint64_t d) {
    if (d == 0) return 0;
    u128 M = ~(u128)0;
    M /= d;
    M += 1;
    return M;
}
static uint64_t mul128_u64(u128 lowbits, uint64_t d) {
    u128 bottom_half = (lowbits & (u128)0xFFFFFFFFFFFFFFFFULL) * d;
    bottom_half >>= 64;
    u128 top_half = (lowbits >> 64) * d;
    u128 both = bottom_half + top_half;
    both >>= 64;
    return (uint64_t)both;
}
static uint64_t fastmod_u64(uint64_t a, u128 M, uint64_t d) {
    u128 lowbits = M * a;
    return mul128_u64(lowbits, d);
}
static inline unsigned popcnt(uint64_t x) { return (unsigned)__builtin_popcountll(x); }
static inline unsigned msb64(uint64_t x) { return 63u - (unsigned)__builtin_clzll(x); }
static inline unsigned select_in_word(uint64_t x, unsigned k) { /* position of the k-th (0-based) set bit */
    for (unsigned i = 0; i < k; ++i) x &= x - 1;
    return (unsigned)__builtin_ctzll(x);
}
static inline uint64_t words_for(uint64_t bits) { return (bits + 63) / 64; }

/* ------------------------------------------------------------------------------------------
 * Serialized shapes (essentials visit order)
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t size, nwords; uint64_t* bits; } bitvec;
typedef struct { uint64_t size, width, mask, nwords; uint64_t* bits; } cvec;
typedef struct {
    uint64_t positions;
    uint64_t nblk; int64_t* blk;
    uint64_t nsub; uint16_t* sub;
    uint64_t novf; uint64_t* ovf;
} darray1;
typedef struct { bitvec high; darray1 d1; cvec low; } efseq;
typedef struct { bitvec bv; uint64_t npairs; uint64_t* pairs; uint64_t nhints; uint64_t* hints; } rsbv;
typedef struct { uint64_t num_dense, num_sparse; u128 M_dense, M_sparse; } bucketer;
typedef struct { cvec ranks, dict; } dictionary;
typedef struct {
    uint64_t seed, num_keys, table_size;
    u128 M;
    bucketer bk;
    dictionary front, back;
    efseq free_slots;
} phf;

struct orc_index {
    int kind, kmer_bits;
    uint8_t k, m;
    uint64_t mm_seed, nkmers, distinct_minimizers;
    /* partitioned (include/partitioned_mphf.hpp:43-52) */
    uint64_t n_maximal, right_coll_sizes_start, none_sizes_start, none_pos_start;
    rsbv root, left_right, max_none; /* quartet_wtree */
    efseq sizes_and_positions;
    /* unpartitioned (include/unpartitioned_mphf.hpp:41-50) */
    uint64_t num_kmers_in_main_index;
    efseq positions, sizes;
    phf minimizer_order, fallback_kmer_order;
};

static void bitvec_free(bitvec* b) { free(b->bits); }
static void cvec_free(cvec* c) { free(c->bits); }
static void darray_free(darray1* d) { free(d->blk); free(d->sub); free(d->ovf); }
static void ef_free(efseq* e) { bitvec_free(&e->high); darray_free(&e->d1); cvec_free(&e->low); }
static void rs_free(rsbv* r) { bitvec_free(&r->bv); free(r->pairs); free(r->hints); }
static void phf_free(phf* f) {
    cvec_free(&f->front.ranks); cvec_free(&f->front.dict);
    cvec_free(&f->back.ranks); cvec_free(&f->back.dict);
    ef_free(&f->free_slots);
}
void orc_free(orc_index* idx) {
    if (!idx) return;
    rs_free(&idx->root); rs_free(&idx->left_right); rs_free(&idx->max_none);
    ef_free(&idx->sizes_and_positions); ef_free(&idx->positions); ef_free(&idx->sizes);
    phf_free(&idx->minimizer_order); phf_free(&idx->fallback_kmer_order);
    free(idx);
}

/* ---------------- essentials::loader (POD raw; vector = u64 n + n*sizeof(T) raw) ------------ */
typedef struct { const uint8_t* p; size_t len, off; int bad; } rd_t;
static void rd_raw(rd_t* r, void* dst, size_t n) {
    if (r->bad || n > r->len - r->off) { r->bad = 1; memset(dst, 0, n); return; }
    memcpy(dst, r->p + r->off, n);
    r->off += n;
}
static uint64_t rd_u64(rd_t* r) { uint64_t v; rd_raw(r, &v, 8); return v; }
static void* rd_vec(rd_t* r, size_t elem, uint64_t* n_out) {
    uint64_t n = rd_u64(r);
    *n_out = 0;
    if (r->bad) return NULL;
    if (n > (r->len - r->off) / elem) { r->bad = 1; return NULL; }
    void* v = malloc(n ? n * elem : 1);
    rd_raw(r, v, n * elem);
    *n_out = n;
    return v;
}
static void rd_bitvec(rd_t* r, bitvec* b) { b->size = rd_u64(r); b->bits = rd_vec(r, 8, &b->nwords); }
static void rd_cvec(rd_t* r, cvec* c) {
    c->size = rd_u64(r); c->width = rd_u64(r); c->mask = rd_u64(r);
    c->bits = rd_vec(r, 8, &c->nwords);
}
static void rd_darray(rd_t* r, darray1* d) {
    d->positions = rd_u64(r);
    d->blk = rd_vec(r, 8, &d->nblk);
    d->sub = rd_vec(r, 2, &d->nsub);
    d->ovf = rd_vec(r, 8, &d->novf);
}
static void rd_ef(rd_t* r, efseq* e) { rd_bitvec(r, &e->high); rd_darray(r, &e->d1); rd_cvec(r, &e->low); }
/* rs_bit_vector::visit  include/rs_bit_vector.hpp:91-96 */
static void rd_rs(rd_t* r, rsbv* v) {
    rd_bitvec(r, &v->bv);
    v->pairs = rd_vec(r, 8, &v->npairs);
    v->hints = rd_vec(r, 8, &v->nhints);
}
static void rd_dict(rd_t* r, dictionary* d) { rd_cvec(r, &d->ranks); rd_cvec(r, &d->dict); }
static void rd_phf(rd_t* r, phf* f) {
    f->seed = rd_u64(r); f->num_keys = rd_u64(r); f->table_size = rd_u64(r);
    rd_raw(r, &f->M, 16);
    f->bk.num_dense = rd_u64(r); f->bk.num_sparse = rd_u64(r);
    rd_raw(r, &f->bk.M_dense, 16); rd_raw(r, &f->bk.M_sparse, 16);
    rd_dict(r, &f->front); rd_dict(r, &f->back);
    rd_ef(r, &f->free_slots);
}

orc_index* orc_load_mem(const uint8_t* buf, size_t len, int kind, int kmer_bits) {
    if (kmer_bits != 64 && kmer_bits != 128) { FAIL("kmer_bits must be 64 or 128"); return NULL; }
    orc_index* x = calloc(1, sizeof *x);
    rd_t r = {buf, len, 0, 0};
    x->kind = kind; x->kmer_bits = kmer_bits;
    rd_raw(&r, &x->k, 1); rd_raw(&r, &x->m, 1);
    x->mm_seed = rd_u64(&r); x->nkmers = rd_u64(&r); x->distinct_minimizers = rd_u64(&r);
    if (kind == ORC_PARTITIONED) { /* include/partitioned_mphf.hpp:204-219 */
        x->n_maximal = rd_u64(&r); x->right_coll_sizes_start = rd_u64(&r);
        x->none_sizes_start = rd_u64(&r); x->none_pos_start = rd_u64(&r);
        rd_phf(&r, &x->minimizer_order);
        rd_rs(&r, &x->root); rd_rs(&r, &x->left_right); rd_rs(&r, &x->max_none); /* quartet_wtree.hpp:43-48 */
        rd_ef(&r, &x->sizes_and_positions);
        rd_phf(&r, &x->fallback_kmer_order);
    } else { /* include/unpartitioned_mphf.hpp:199-211 */
        x->num_kmers_in_main_index = rd_u64(&r);
        rd_phf(&r, &x->minimizer_order);
        rd_ef(&r, &x->positions); rd_ef(&r, &x->sizes);
        rd_phf(&r, &x->fallback_kmer_order);
    }
    if (r.bad || r.off != r.len) {
        FAIL("malformed .lph (bad=%d, consumed %zu of %zu bytes)", r.bad, r.off, r.len);
        orc_free(x);
        return NULL;
    }
    if (x->k > (unsigned)(kmer_bits / 2 - 1) || x->m > x->k || x->m > 32 || x->m == 0) {
        FAIL("k=%u m=%u out of range for kmer_bits=%d", x->k, x->m, kmer_bits);
        orc_free(x);
        return NULL;
    }
    return x;
}

orc_index* orc_load(const char* path, int kind, int kmer_bits) {
    FILE* f = fopen(path, "rb");
    if (!f) { FAIL("cannot open %s", path); return NULL; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* buf = malloc(n > 0 ? (size_t)n : 1);
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); FAIL("short read"); return NULL; }
    fclose(f);
    orc_index* x = orc_load_mem(buf, (size_t)n, kind, kmer_bits);
    free(buf);
    return x;
}

uint64_t orc_k(const orc_index* x) { return x->k; }
size_t orc_kmer_words(const orc_index* x) { return x->kmer_bits == 64 ? 1 : 2; }
uint64_t orc_m(const orc_index* x) { return x->m; }
uint64_t orc_mm_seed(const orc_index* x) { return x->mm_seed; }
uint64_t orc_kmer_count(const orc_index* x) { return x->nkmers; }
uint64_t orc_minimizer_L0(const orc_index* x) { return x->distinct_minimizers; }
uint64_t orc_fallback_keys(const orc_index* x) { return x->fallback_kmer_order.num_keys; }

/* ------------------------------------------------------------------------------------------
 * pthash encoders, read side
 * ---------------------------------------------------------------------------------------- */
static inline int bv_get(const bitvec* b, uint64_t i) { return (int)((b->bits[i >> 6] >> (i & 63)) & 1); }

/* compact_vector::access: bits [i*w, i*w+w) of the little-endian bit stream */
static inline uint64_t cv_access(const cvec* c, uint64_t i) {
    if (c->width == 0) return 0;
    uint64_t pos = i * c->width, blk = pos >> 6, sh = pos & 63;
    uint64_t v = c->bits[blk] >> sh;
    if (sh + c->width > 64) v |= c->bits[blk + 1] << (64 - sh);
    return v & c->mask;
}

/* darray1::select (succinct darray: 1024-one blocks, 32-one sub-blocks, overflow table) */
static uint64_t d1_select(const darray1* d, const bitvec* bv, uint64_t idx) {
    uint64_t block = idx / 1024;
    int64_t block_pos = d->blk[block];
    if (block_pos < 0) {
        uint64_t overflow_pos = (uint64_t)(-block_pos - 1);
        return d->ovf[overflow_pos + (idx & 1023)];
    }
    uint64_t subblock = idx / 32;
    uint64_t start_pos = (uint64_t)block_pos + d->sub[subblock];
    uint64_t reminder = idx & 31;
    if (!reminder) return start_pos;
    uint64_t word_idx = start_pos >> 6, word_shift = start_pos & 63;
    uint64_t word = bv->bits[word_idx] & (~(uint64_t)0 << word_shift);
    for (;;) {
        unsigned pc = popcnt(word);
        if (reminder < pc) break;
        reminder -= pc;
        word = bv->bits[++word_idx];
    }
    return (word_idx << 6) + select_in_word(word, (unsigned)reminder);
}
/* bit_vector::unary_iterator(bv,pos).next(): next set bit at or after pos */
static uint64_t bv_next_one(const bitvec* b, uint64_t pos) {
    uint64_t w = pos >> 6;
    uint64_t buf = b->bits[w] & (~(uint64_t)0 << (pos & 63));
    while (!buf) buf = b->bits[++w];
    return (w << 6) + (unsigned)__builtin_ctzll(buf);
}

/* ef access: identical for pthash::ef_sequence<false>::access and lphash ef_sequence::access
 * (include/ef_sequence.hpp:77-81) */
static uint64_t ef_access(const efseq* e, uint64_t i) {
    return ((d1_select(&e->d1, &e->high, i) - i) << e->low.width) | cv_access(&e->low, i);
}
/* include/ef_sequence.hpp:83-94 */
static void ef_pair(const efseq* e, uint64_t i, uint64_t* v1, uint64_t* v2) {
    uint64_t low1 = cv_access(&e->low, i);
    uint64_t low2 = cv_access(&e->low, i + 1);
    uint64_t l = e->low.width;
    uint64_t pos = d1_select(&e->d1, &e->high, i);
    uint64_t h1 = pos - i;
    uint64_t h2 = bv_next_one(&e->high, pos + 1) - i - 1;
    *v1 = (h1 << l) | low1;
    *v2 = (h2 << l) | low2;
}
/* include/ef_sequence.hpp:96-99 */
static uint64_t ef_diff(const efseq* e, uint64_t i) { uint64_t a, b; ef_pair(e, i, &a, &b); return b - a; }
static inline uint64_t ef_size(const efseq* e) { return e->low.size; } /* ef_sequence.hpp:101 */

/* skew_bucketer::bucket, T = uint64_t(0.6 * UINT64_MAX) */
static inline uint64_t bucket_of(const bucketer* b, uint64_t hash) {
    static const uint64_t T = 0x9999999999999800ULL;
    return (hash < T) ? fastmod_u64(hash, b->M_dense, b->num_dense)
                      : b->num_dense + fastmod_u64(hash, b->M_sparse, b->num_sparse);
}
/* dual<dictionary,dictionary>::access ; dictionary::access = dict[ranks[i]] */
static inline uint64_t pilots_access(const phf* f, uint64_t i) {
    if (i < f->front.ranks.size) return cv_access(&f->front.dict, cv_access(&f->front.ranks, i));
    i -= f->front.ranks.size;
    return cv_access(&f->back.dict, cv_access(&f->back.ranks, i));
}
/* single_phf<...,true>::position(hash): the bucket comes from hash.first(), the table position from hash.second() */
static uint64_t phf_position2(const phf* f, uint64_t first, uint64_t second) {
    uint64_t bucket = bucket_of(&f->bk, first);
    uint64_t pilot = pilots_access(f, bucket);
    uint64_t hashed_pilot = hash_u64(pilot, f->seed);
    uint64_t p = fastmod_u64(second ^ hashed_pilot, f->M, f->table_size);
    if (p < f->num_keys) return p;
    return ef_access(&f->free_slots, p - f->num_keys);
}
/* hash64 (the hashers of LPHash's own functions): first() == second() */
static uint64_t phf_position(const phf* f, uint64_t hash) { return phf_position2(f, hash, hash); }
uint64_t orc_minimizer_order(const orc_index* x, uint64_t mm) {
    return phf_position(&x->minimizer_order, hash_u64(mm, x->minimizer_order.seed));
}
static inline uint64_t fallback_order(const orc_index* x, u128 kmer) {
    const phf* f = &x->fallback_kmer_order;
    return phf_position(f, fallback_hash(kmer, x->kmer_bits, f->seed));
}

/* ------------------------------------------------------------------------------------------
 * rs_bit_vector (include/rs_bit_vector.hpp:27-38,99-114) and quartet_wtree::rank_of
 * (src/quartet_wtree.cpp:84-106)
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t rs_num_ones(const rsbv* v) { return v->pairs[v->npairs - 2]; }
static inline uint64_t rs_sub_block_rank(const rsbv* v, uint64_t sub_block) {
    uint64_t block = sub_block / 8;
    uint64_t r = v->pairs[block * 2];
    uint64_t left = sub_block % 8;
    r += v->pairs[block * 2 + 1] >> ((7 - left) * 9) & 0x1FF;
    return r;
}
static uint64_t rs_rank(const rsbv* v, uint64_t pos) {
    if (pos == v->bv.size) return rs_num_ones(v);
    uint64_t sub_block = pos / 64;
    uint64_t r = rs_sub_block_rank(v, sub_block);
    uint64_t sub_left = pos % 64;
    if (sub_left) r += popcnt(v->bv.bits[sub_block] << (64 - sub_left));
    return r;
}
static inline uint64_t rs_rank0(const rsbv* v, uint64_t pos) { return pos - rs_rank(v, pos); }
static inline uint64_t rank_switch(int type, const rsbv* v, uint64_t idx) { return type ? rs_rank(v, idx) : rs_rank0(v, idx); }

enum { T_LEFT = 0, T_RIGHT_OR_COLLISION = 1, T_MAXIMAL = 2, T_NONE = 3 }; /* include/quartet_wtree.hpp:7 */

static void wtree_rank_of(const orc_index* x, uint64_t idx, int* type, uint64_t* rank) {
    int msb = bv_get(&x->root.bv, idx), lsb;
    uint64_t r = rank_switch(msb, &x->root, idx);
    if (msb) { lsb = bv_get(&x->max_none.bv, r); *rank = rank_switch(lsb, &x->max_none, r); }
    else     { lsb = bv_get(&x->left_right.bv, r); *rank = rank_switch(lsb, &x->left_right, r); }
    *type = (msb << 1) | lsb;
}

/* ------------------------------------------------------------------------------------------
 * mphf::query (src/partitioned_mphf.cpp:292-339), mphf_alt::query (src/unpartitioned_mphf.cpp:191-206)
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t hval, global_rank, local_rank; uint8_t type; int collision; } mm_context_t;

static mm_context_t query_p(const orc_index* x, u128 kmer, uint64_t minimizer, uint32_t position) {
    mm_context_t res; memset(&res, 0, sizeof res);
    const uint64_t k = x->k, m = x->m;
    uint64_t mp_hash = orc_minimizer_order(x, minimizer);
    int mm_type; uint64_t mm_type_rank;
    wtree_rank_of(x, mp_hash, &mm_type, &mm_type_rank);
    switch (mm_type) {
        case T_LEFT:
            res.global_rank = ef_access(&x->sizes_and_positions, mm_type_rank) + (k - m + 1) * x->n_maximal;
            res.local_rank = position;
            res.type = T_LEFT;
            break;
        case T_RIGHT_OR_COLLISION: {
            uint64_t val1, val2;
            ef_pair(&x->sizes_and_positions, x->right_coll_sizes_start + mm_type_rank, &val1, &val2);
            uint64_t sk_size = val2 - val1;
            if (sk_size == 0) {
                res.global_rank = ef_access(&x->sizes_and_positions, x->none_pos_start) + (k - m + 1) * x->n_maximal;
                res.local_rank = fallback_order(x, kmer);
                res.type = T_NONE + 1;
            } else {
                res.global_rank = val1 + (k - m + 1) * x->n_maximal;
                res.local_rank = k - m - position;
                res.type = T_RIGHT_OR_COLLISION;
            }
        } break;
        case T_MAXIMAL:
            res.global_rank = (k - m + 1) * mm_type_rank;
            res.local_rank = position;
            res.type = T_MAXIMAL;
            break;
        case T_NONE: {
            res.global_rank = ef_access(&x->sizes_and_positions, x->none_sizes_start + mm_type_rank) + (k - m + 1) * x->n_maximal;
            uint64_t sk_size = ef_diff(&x->sizes_and_positions, x->none_pos_start + mm_type_rank);
            res.local_rank = sk_size - position;
            res.type = T_NONE;
        } break;
    }
    res.hval = res.global_rank + res.local_rank;
    return res;
}

static mm_context_t query_u(const orc_index* x, u128 kmer, uint64_t minimizer, uint32_t position) {
    mm_context_t res; memset(&res, 0, sizeof res);
    uint64_t index = orc_minimizer_order(x, minimizer);
    uint64_t val1, val2;
    ef_pair(&x->sizes, index, &val1, &val2);
    uint64_t size = val2 - val1;
    if (size == 0) {
        res.hval = x->num_kmers_in_main_index + fallback_order(x, kmer);
        res.collision = 1;
        return res;
    }
    uint64_t p1 = ef_diff(&x->positions, index);
    res.hval = val1 + p1 - position;
    res.collision = 0;
    return res;
}
static inline mm_context_t query_any(const orc_index* x, u128 kmer, uint64_t mm, uint32_t p) {
    return x->kind == ORC_PARTITIONED ? query_p(x, kmer, mm, p) : query_u(x, kmer, mm, p);
}

/* ------------------------------------------------------------------------------------------
 * constants::seq_nt4_table  (src/constants.cpp:5-13): A/a=0 C/c=1 G/g=2 T/t/U/u=3 else 4
 * ---------------------------------------------------------------------------------------- */
static uint8_t nt4(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}

/* debug::compute_minimizer_triplet  include/mphf_utils.hpp:118-137 (`<=`: leftmost minimum wins) */
static void minimizer_triplet(u128 kmer, uint64_t k, uint64_t m, uint64_t seed, uint64_t* mm_out, uint64_t* p_out) {
    uint64_t min_hash = ~(uint64_t)0;
    u128 minimizer = ~(u128)0;
    u128 mask = (((u128)1) << (2 * m)) - 1;
    uint64_t pos = 0;
    for (uint64_t i = 0; i != k - m + 1; ++i) {
        u128 mmer = kmer & mask;
        uint64_t hash = hash_u64((uint64_t)mmer, seed); /* kmer_t narrows to uint64_t at the call */
        if (hash <= min_hash) { min_hash = hash; minimizer = mmer; pos = i; }
        kmer >>= 2;
    }
    *mm_out = (uint64_t)minimizer;
    *p_out = k - (pos + m);
}

void orc_query_packed(const orc_index* x, const uint64_t* kmers, size_t n, uint64_t* out) {
    for (size_t i = 0; i < n; ++i) {
        u128 kmer = x->kmer_bits == 64 ? (u128)kmers[i] : ((u128)kmers[2 * i] | ((u128)kmers[2 * i + 1] << 64));
        uint64_t mm, p;
        minimizer_triplet(kmer, x->k, x->m, x->mm_seed, &mm, &p);
        out[i] = query_any(x, kmer, mm, (uint32_t)p).hval;
    }
}

/* mm_quartet_t  include/mm_quartet.hpp:7-26 */
typedef struct { uint64_t hash, id, itself; uint8_t p1, size; } mm_quartet_t;

/* MPHF::operator()  include/partitioned_mphf.hpp:73-197, include/unpartitioned_mphf.hpp:72-192 */
int64_t orc_query(const orc_index* x, const char* contig, size_t length, int streaming, uint64_t* out, size_t cap) {
    const uint64_t k = x->k, m = x->m, W = k - m + 1;
    uint64_t nres = 0;
#define PUSH(v) do { if (nres < cap) out[nres] = (v); ++nres; } while (0)
    if (length < k) return 0;
    if (streaming) {
        uint64_t shift = 2 * (m - 1);
        uint64_t mask = m == 32 ? ~(uint64_t)0 : ((1ULL << (2 * m)) - 1);
        uint64_t mm[2] = {0, 0};
        uint64_t km_shift = 2 * (k - 1);
        u128 km_mask = (((u128)1) << (2 * k)) - 1;
        u128 km[2] = {0, 0};
        uint64_t nbases_since_last_break = 0;
        uint8_t find_brand_new_min = 0;
        uint32_t p1 = 0;
        mm_quartet_t* buffer = calloc(W, sizeof *buffer);
        size_t buf_pos = 0, min_pos = W;
        mm_quartet_t current; memset(&current, 0, sizeof current);
        mm_context_t mm_ctx; memset(&mm_ctx, 0, sizeof mm_ctx);
        const uint8_t z = 0;
        for (uint64_t i = 0; i < length; ++i) {
            int c = nt4((uint8_t)contig[i]);
            if (c < 4) {
                mm[0] = (mm[0] << 2 | (uint64_t)c) & mask;
                mm[1] = (mm[1] >> 2) | (3ULL ^ (uint64_t)c) << shift;
                km[0] = (km[0] << 2 | (u128)c) & km_mask;
                km[1] = (km[1] >> 2) | (((u128)3 ^ (u128)c) << km_shift);
                ++nbases_since_last_break;
                if (nbases_since_last_break >= m) {
                    current.itself = mm[z];
                    current.hash = hash_u64(mm[z], x->mm_seed);
                    if (buf_pos == min_pos) find_brand_new_min = 1;
                    buffer[buf_pos] = current;
                    if (nbases_since_last_break == k) {
                        find_brand_new_min = 1;
                    } else if (nbases_since_last_break > k) {
                        if (buffer[min_pos].hash > buffer[buf_pos].hash) {
                            p1 = (uint32_t)(k - m);
                            min_pos = buf_pos;
                            find_brand_new_min = 2;
                        }
                    }
                    switch (find_brand_new_min) {
                        case 0:
                            if (nbases_since_last_break >= k) {
                                if (x->kind == ORC_PARTITIONED) {
                                    if (mm_ctx.type == (T_NONE + 1)) mm_ctx.local_rank = fallback_order(x, km[z]);
                                    else if (mm_ctx.type == T_RIGHT_OR_COLLISION || mm_ctx.type == T_NONE) ++mm_ctx.local_rank;
                                    else --mm_ctx.local_rank;
                                    mm_ctx.hval = mm_ctx.global_rank + mm_ctx.local_rank;
                                } else { /* unpartitioned_mphf.hpp:130-139 */
                                    if (mm_ctx.collision) mm_ctx.hval = fallback_order(x, km[z]) + ef_access(&x->sizes, ef_size(&x->sizes) - 1);
                                    else ++mm_ctx.hval;
                                }
                                PUSH(mm_ctx.hval);
                            }
                            break;
                        case 1: {
                            min_pos = (buf_pos + 1) % W;
                            p1 = 0;
                            uint32_t tmp = 1;
                            for (size_t j = (buf_pos + 2) % W; j < W; ++j) {
                                if (buffer[min_pos].hash > buffer[j].hash) { min_pos = j; p1 = tmp; }
                                ++tmp;
                            }
                            for (size_t j = 0; j <= (buf_pos + 2) % W; ++j) {
                                if (buffer[min_pos].hash > buffer[j].hash) { min_pos = j; p1 = tmp; }
                                ++tmp;
                            }
                        } /* fallthrough */
                        case 2:
                            find_brand_new_min = 0;
                            mm_ctx = query_any(x, km[z], buffer[min_pos].itself, p1);
                            PUSH(mm_ctx.hval);
                            break;
                    }
                    buf_pos = (buf_pos + 1) % W;
                }
            } else {
                nbases_since_last_break = 0;
                buf_pos = 0;
            }
        }
        free(buffer);
    } else {
        for (size_t i = 0; i < length - k + 1; ++i) {
            u128 kmer = 0; /* debug::string_to_integer_no_reverse  mphf_utils.hpp:108-116 */
            for (uint64_t j = 0; j != k; ++j) kmer = (kmer << 2) | (u128)(nt4((uint8_t)contig[i + j]) & 3);
            uint64_t mmv, p;
            minimizer_triplet(kmer, k, m, x->mm_seed, &mmv, &p);
            PUSH(query_any(x, kmer, mmv, (uint32_t)p).hval);
        }
    }
#undef PUSH
    return (int64_t)nres;
}

/* ==========================================================================================
 * BUILD PATH
 * ======================================================================================== */

/* ---------------- essentials::saver ---------------- */
typedef struct { uint8_t* p; size_t len, cap; } wr_t;
static void wr_raw(wr_t* w, const void* src, size_t n) {
    if (w->len + n > w->cap) {
        size_t nc = w->cap ? w->cap * 2 : 4096;
        while (nc < w->len + n) nc *= 2;
        w->p = realloc(w->p, nc);
        w->cap = nc;
    }
    memcpy(w->p + w->len, src, n);
    w->len += n;
}
static void wr_u64(wr_t* w, uint64_t v) { wr_raw(w, &v, 8); }
static void wr_vec(wr_t* w, const void* d, size_t elem, uint64_t n) { wr_u64(w, n); if (n) wr_raw(w, d, n * elem); }
static void wr_bitvec(wr_t* w, const bitvec* b) { wr_u64(w, b->size); wr_vec(w, b->bits, 8, b->nwords); }
static void wr_cvec(wr_t* w, const cvec* c) { wr_u64(w, c->size); wr_u64(w, c->width); wr_u64(w, c->mask); wr_vec(w, c->bits, 8, c->nwords); }
static void wr_darray(wr_t* w, const darray1* d) {
    wr_u64(w, d->positions);
    wr_vec(w, d->blk, 8, d->nblk); wr_vec(w, d->sub, 2, d->nsub); wr_vec(w, d->ovf, 8, d->novf);
}
static void wr_ef(wr_t* w, const efseq* e) { wr_bitvec(w, &e->high); wr_darray(w, &e->d1); wr_cvec(w, &e->low); }
static void wr_rs(wr_t* w, const rsbv* v) { wr_bitvec(w, &v->bv); wr_vec(w, v->pairs, 8, v->npairs); wr_vec(w, v->hints, 8, v->nhints); }
static void wr_phf(wr_t* w, const phf* f) {
    wr_u64(w, f->seed); wr_u64(w, f->num_keys); wr_u64(w, f->table_size); wr_raw(w, &f->M, 16);
    wr_u64(w, f->bk.num_dense); wr_u64(w, f->bk.num_sparse); wr_raw(w, &f->bk.M_dense, 16); wr_raw(w, &f->bk.M_sparse, 16);
    wr_cvec(w, &f->front.ranks); wr_cvec(w, &f->front.dict); wr_cvec(w, &f->back.ranks); wr_cvec(w, &f->back.dict);
    wr_ef(w, &f->free_slots);
}

/* ---------------- pthash encoders, build side ---------------- */
static void bitvec_init(bitvec* b, uint64_t size) { b->size = size; b->nwords = words_for(size); b->bits = calloc(b->nwords ? b->nwords : 1, 8); }
static inline void bv_set(bitvec* b, uint64_t i) { b->bits[i >> 6] |= 1ULL << (i & 63); }

/* compact_vector::builder(n, w): words_for(n*w)+1 words (one padding word for safe access) */
static void cvec_init(cvec* c, uint64_t n, uint64_t w) {
    c->size = n; c->width = w;
    c->mask = (w == 64) ? ~(uint64_t)0 : ((1ULL << w) - 1);
    c->nwords = words_for(n * w) + 1;
    c->bits = calloc(c->nwords, 8);
}
static void cvec_set(cvec* c, uint64_t i, uint64_t v) {
    if (!c->width) return;
    uint64_t pos = i * c->width, blk = pos >> 6, sh = pos & 63;
    c->bits[blk] |= v << sh;
    if (sh + c->width > 64) c->bits[blk + 1] |= v >> (64 - sh);
}
/* compact_vector::build(begin, n): width = max==0 ? 1 : ceil(log2(max+1)) */
static void cvec_build(cvec* c, const uint64_t* v, uint64_t n) {
    uint64_t mx = 0;
    for (uint64_t i = 0; i < n; ++i) if (v[i] > mx) mx = v[i];
    uint64_t w = mx == 0 ? 1 : msb64(mx) + 1;
    cvec_init(c, n, w);
    for (uint64_t i = 0; i < n; ++i) cvec_set(c, i, v[i]);
}

/* darray1::build + flush_cur_block */
static void d1_flush(darray1* d, const uint64_t* cur, uint64_t ncur) {
    if (cur[ncur - 1] - cur[0] < (1u << 16)) {
        d->blk[d->nblk++] = (int64_t)cur[0];
        for (uint64_t i = 0; i < ncur; i += 32) d->sub[d->nsub++] = (uint16_t)(cur[i] - cur[0]);
    } else {
        d->blk[d->nblk++] = -(int64_t)d->novf - 1;
        for (uint64_t i = 0; i < ncur; ++i) d->ovf[d->novf++] = cur[i];
        for (uint64_t i = 0; i < ncur; i += 32) d->sub[d->nsub++] = (uint16_t)-1;
    }
}
static void d1_build(darray1* d, const bitvec* bv) {
    memset(d, 0, sizeof *d);
    uint64_t ones = 0;
    for (uint64_t w = 0; w < bv->nwords; ++w) ones += popcnt(bv->bits[w]);
    uint64_t nblk = (ones + 1023) / 1024;
    d->blk = malloc((nblk ? nblk : 1) * 8);
    d->sub = malloc((nblk ? nblk * 32 : 1) * 2);
    d->ovf = malloc((ones ? ones : 1) * 8);
    uint64_t* cur = malloc(1024 * 8);
    uint64_t ncur = 0;
    for (uint64_t w = 0; w < bv->nwords; ++w) {
        uint64_t word = bv->bits[w];
        while (word) {
            uint64_t pos = (w << 6) + (unsigned)__builtin_ctzll(word);
            word &= word - 1;
            if (pos >= bv->size) break;
            cur[ncur++] = pos;
            d->positions += 1;
            if (ncur == 1024) { d1_flush(d, cur, ncur); ncur = 0; }
        }
    }
    if (ncur) d1_flush(d, cur, ncur);
    free(cur);
}

/* Elias-Fano encode of a non-decreasing sequence vals[0..n) with universe u.
 *  leading_zero = 0: pthash::ef_sequence<false>::encode (free slots), u = last value
 *  leading_zero = 1: lphash ef_sequence::encode, include/ef_sequence.hpp:37-75 (vals are the
 *                    cumulative sums produced by cumulative_iterator, :9-33; a 0 is prepended) */
static void ef_encode(efseq* e, const uint64_t* vals, uint64_t n, uint64_t u, int leading_zero) {
    memset(e, 0, sizeof *e);
    if (n == 0) return;
    uint64_t nn = n + (leading_zero ? 1 : 0);
    uint64_t l = (nn && u / nn) ? msb64(u / nn) : 0;
    bitvec_init(&e->high, nn + (u >> l) + 1);
    cvec_init(&e->low, nn, l);
    uint64_t low_mask = (1ULL << l) - 1;
    uint64_t at = 0;
    if (leading_zero) { bv_set(&e->high, 0); at = 1; } /* low value 0 already there */
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t v = vals[i];
        if (l) cvec_set(&e->low, i + at, v & low_mask);
        bv_set(&e->high, (v >> l) + i + at);
    }
    d1_build(&e->d1, &e->high);
}

/* rs_bit_vector::build_indices(false)  include/rs_bit_vector.hpp:120-157 */
static void rs_build(rsbv* v) {
    uint64_t nw = v->bv.nwords;
    uint64_t* prs = malloc((2 * (nw / 8 + 2) + 2) * 8);
    uint64_t np = 0;
    uint64_t next_rank = 0, cur_subrank = 0, subranks = 0;
    prs[np++] = 0;
    for (uint64_t i = 0; i < nw; ++i) {
        uint64_t word_pop = popcnt(v->bv.bits[i]);
        uint64_t shift = i % 8;
        if (shift) { subranks <<= 9; subranks |= cur_subrank; }
        next_rank += word_pop;
        cur_subrank += word_pop;
        if (shift == 7) { prs[np++] = subranks; prs[np++] = next_rank; subranks = 0; cur_subrank = 0; }
    }
    uint64_t left = 8 - nw % 8;
    for (uint64_t i = 0; i < left; ++i) { subranks <<= 9; subranks |= cur_subrank; }
    prs[np++] = subranks;
    if (nw % 8) { prs[np++] = next_rank; prs[np++] = 0; }
    v->pairs = prs; v->npairs = np;
    v->hints = NULL; v->nhints = 0;
}

/* ---------------- PTHash single_phf build (internal-memory search, first valid pilot) -------- */
typedef struct { uint64_t bucket, hash; } bh_t;
static int cmp_bh(const void* a, const void* b) {
    const bh_t *x = a, *y = b;
    if (x->bucket != y->bucket) return x->bucket < y->bucket ? -1 : 1;
    return x->hash < y->hash ? -1 : (x->hash > y->hash);
}
typedef struct { uint64_t val, freq; } vf_t;
static int cmp_vf(const void* a, const void* b) {
    const vf_t *x = a, *y = b;
    if (x->freq != y->freq) return x->freq > y->freq ? -1 : 1;
    return x->val < y->val ? -1 : (x->val > y->val);
}
static int cmp_vf_val(const void* a, const void* b) {
    const vf_t *x = a, *y = b;
    return x->val < y->val ? -1 : (x->val > y->val);
}
static int cmp_u64(const void* a, const void* b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : (x > y); }

/* pthash dictionary::encode: codewords by non-increasing frequency (ties: smaller value first) */
static void dict_encode(dictionary* d, const uint64_t* v, uint64_t n) {
    uint64_t* s = malloc((n ? n : 1) * 8);
    memcpy(s, v, n * 8);
    qsort(s, n, 8, cmp_u64);
    vf_t* vf = malloc((n ? n : 1) * sizeof *vf);
    uint64_t nd = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i;
        while (j < n && s[j] == s[i]) ++j;
        vf[nd].val = s[i]; vf[nd].freq = j - i; ++nd;
        i = j;
    }
    qsort(vf, nd, sizeof *vf, cmp_vf);
    uint64_t* dict = malloc((nd ? nd : 1) * 8);
    for (uint64_t i = 0; i < nd; ++i) dict[i] = vf[i].val;
    /* value -> codeword: binary search in a by-value sorted copy */
    vf_t* byval = malloc((nd ? nd : 1) * sizeof *byval);
    for (uint64_t i = 0; i < nd; ++i) { byval[i].val = vf[i].val; byval[i].freq = i; }
    qsort(byval, nd, sizeof *byval, cmp_vf_val);
    uint64_t* ranks = malloc((n ? n : 1) * 8);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t lo = 0, hi = nd;
        while (hi - lo > 1) { uint64_t mid = (lo + hi) / 2; if (byval[mid].val <= v[i]) lo = mid; else hi = mid; }
        ranks[i] = byval[lo].freq;
    }
    cvec_build(&d->ranks, ranks, n);
    cvec_build(&d->dict, dict, nd);
    free(s); free(vf); free(dict); free(byval); free(ranks);
}

static int phf_build2(phf* f, const uint64_t* firsts, const uint64_t* hashes, uint64_t n, uint64_t seed, double c, double alpha);
static int phf_build(phf* f, const uint64_t* hashes, uint64_t n, uint64_t seed, double c, double alpha) {
    return phf_build2(f, hashes, hashes, n, seed, c, alpha);
}
/* firsts[i] = hash.first() (bucket), hashes[i] = hash.second() (payload: table position, duplicate check) of key i */
static int phf_build2(phf* f, const uint64_t* firsts, const uint64_t* hashes, uint64_t n, uint64_t seed, double c, double alpha) {
    memset(f, 0, sizeof *f);
    f->seed = seed;
    f->num_keys = n;
    if (n == 0) return 0; /* nothing to hash: all-zero structure (the reference never builds one) */
    uint64_t table_size = (uint64_t)((double)n / alpha);
    if ((table_size & (table_size - 1)) == 0) table_size += 1;
    /* PTHash: num_buckets = ceil(c*n/log2(n)); log2(1)=0 is undefined there, use the n=2 value */
    uint64_t num_buckets = n < 2 ? 6 : (uint64_t)ceil((c * (double)n) / log2((double)n));
    f->table_size = table_size;
    f->M = computeM_u64(table_size);
    f->bk.num_dense = (uint64_t)(0.3 * (double)num_buckets);
    f->bk.num_sparse = num_buckets - f->bk.num_dense;
    f->bk.M_dense = computeM_u64(f->bk.num_dense);
    f->bk.M_sparse = computeM_u64(f->bk.num_sparse);

    bh_t* bh = malloc(n * sizeof *bh);
    for (uint64_t i = 0; i < n; ++i) { bh[i].hash = hashes[i]; bh[i].bucket = bucket_of(&f->bk, firsts[i]); }
    qsort(bh, n, sizeof *bh, cmp_bh);
    uint64_t* bstart = calloc(num_buckets + 1, 8);
    for (uint64_t i = 0; i < n; ++i) bstart[bh[i].bucket + 1]++;
    uint64_t max_size = 0;
    for (uint64_t b = 0; b < num_buckets; ++b) { if (bstart[b + 1] > max_size) max_size = bstart[b + 1]; bstart[b + 1] += bstart[b]; }
    for (uint64_t i = 1; i < n; ++i)
        if (bh[i].bucket == bh[i - 1].bucket && bh[i].hash == bh[i - 1].hash) {
            FAIL("pthash seed failure: duplicate 64-bit hash in a bucket (duplicate key in the input?)");
            free(bh); free(bstart);
            return -1;
        }
    /* buckets by non-increasing size, then increasing id */
    uint64_t* order = malloc(num_buckets * 8);
    {
        uint64_t* cnt = calloc(max_size + 2, 8);
        for (uint64_t b = 0; b < num_buckets; ++b) cnt[max_size - (bstart[b + 1] - bstart[b]) + 1]++;
        for (uint64_t s = 0; s <= max_size; ++s) cnt[s + 1] += cnt[s];
        for (uint64_t b = 0; b < num_buckets; ++b) order[cnt[max_size - (bstart[b + 1] - bstart[b])]++] = b;
        free(cnt);
    }
    uint64_t* pilots = calloc(num_buckets, 8);
    uint64_t* taken = calloc(words_for(table_size) + 1, 8);
    uint64_t* pos = malloc((max_size ? max_size : 1) * 8);
    enum { HP_CACHE = 1 << 16 };
    uint64_t* hp_cache = malloc(HP_CACHE * 8);
    for (uint64_t p = 0; p < HP_CACHE; ++p) hp_cache[p] = hash_u64(p, seed);
    for (uint64_t oi = 0; oi < num_buckets; ++oi) {
        uint64_t b = order[oi];
        uint64_t s = bstart[b + 1] - bstart[b];
        if (s == 0) break;
        const bh_t* keys = bh + bstart[b];
        for (uint64_t pilot = 0;; ++pilot) {
            if (pilot >> 28) { /* unplaceable bucket: PTHash itself would not terminate (see lphash_amd/csrc/builder.cpp) */
                FAIL("pthash search: no pilot below 2^28 for some bucket (degenerate tiny key set)");
                free(bh); free(bstart); free(order); free(pilots); free(taken); free(pos); free(hp_cache);
                return -1;
            }
            uint64_t hp = pilot < HP_CACHE ? hp_cache[pilot] : hash_u64(pilot, seed);
            uint64_t j = 0;
            for (; j < s; ++j) {
                uint64_t p = (keys[j].hash ^ hp) % table_size; /* == fastmod_u64(.., M, table_size) */
                if ((taken[p >> 6] >> (p & 63)) & 1) break;
                pos[j] = p;
            }
            if (j < s) continue;
            int dup = 0;
            for (uint64_t a = 1; a < s && !dup; ++a)
                for (uint64_t bb = 0; bb < a; ++bb)
                    if (pos[a] == pos[bb]) { dup = 1; break; }
            if (dup) continue;
            for (j = 0; j < s; ++j) taken[pos[j] >> 6] |= 1ULL << (pos[j] & 63);
            pilots[b] = pilot;
            break;
        }
    }
    /* dual<dictionary,dictionary>::encode: front = first 0.3*n buckets */
    uint64_t front_size = (uint64_t)((double)num_buckets * 0.3);
    dict_encode(&f->front, pilots, front_size);
    dict_encode(&f->back, pilots + front_size, num_buckets - front_size);
    /* minimal: fill_free_slots + ef_sequence<false>::encode */
    if (n < table_size) {
        uint64_t nfree = table_size - n;
        uint64_t* fs = malloc(nfree * 8);
        uint64_t nf = 0;
        uint64_t next_used_slot = n, last_free_slot = 0, last_valid_free_slot = 0;
#define TAKEN(i) ((taken[(i) >> 6] >> ((i) & 63)) & 1)
        for (;;) {
            while (last_free_slot < n && TAKEN(last_free_slot)) ++last_free_slot;
            if (last_free_slot == n) break;
            while (!TAKEN(next_used_slot)) { fs[nf++] = last_free_slot; ++next_used_slot; }
            fs[nf++] = last_free_slot;
            last_valid_free_slot = last_free_slot;
            ++next_used_slot;
            ++last_free_slot;
        }
        while (next_used_slot != table_size) { fs[nf++] = last_valid_free_slot; ++next_used_slot; }
#undef TAKEN
        ef_encode(&f->free_slots, fs, nfree, fs[nfree - 1], 0);
        free(fs);
    }
    free(bh); free(bstart); free(order); free(pilots); free(taken); free(pos); free(hp_cache);
    return 0;
}

/* ---------------- minimizer::from_string  include/minimizer.hpp:11-170 ---------------- */
typedef struct { uint64_t itself, id; uint8_t p1, size; } mm_record_t;
typedef struct { mm_record_t* v; size_t n, cap; } recvec;
static void rec_push(recvec* a, mm_record_t r) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 1024; a->v = realloc(a->v, a->cap * sizeof *a->v); }
    a->v[a->n++] = r;
}
static void quartet_clear(mm_quartet_t* q) { q->itself = 0; q->hash = ~(uint64_t)0; q->p1 = 0xFF; q->size = 0xFF; }
static void update_output(recvec* acc, const mm_quartet_t* added) {
    mm_record_t r = {added->itself, added->id, added->p1, added->size};
    rec_push(acc, r);
}

static uint64_t from_string(const char* contig, size_t contig_size, uint32_t k, uint32_t m, uint64_t seed,
                            uint64_t* mm_count, recvec* accumulator) {
    const size_t W = k - m + 1;
    size_t buf_pos, min_pos;
    mm_quartet_t current; memset(&current, 0, sizeof current);
    uint64_t shift = 2 * (m - 1);
    uint64_t mask = m == 32 ? ~(uint64_t)0 : ((1ULL << (2 * m)) - 1);
    uint64_t mm[2] = {0, 0};
    uint64_t nbases_since_last_break = 0;
    uint32_t sks = 0, p1 = 0;
    uint64_t kmer_count;
    mm_quartet_t* buffer = calloc(W, sizeof *buffer);
    const uint8_t z = 0;
    int find_brand_new_min = 0;

    buf_pos = 0;
    min_pos = W;
    kmer_count = 0;
    for (uint64_t i = 0; i < contig_size; ++i) {
        int c = nt4((uint8_t)contig[i]);
        quartet_clear(&current);
        if (c < 4) {
            mm[0] = (mm[0] << 2 | (uint64_t)c) & mask;
            mm[1] = (mm[1] >> 2) | (3ULL ^ (uint64_t)c) << shift;
            ++nbases_since_last_break;
            if (nbases_since_last_break >= m) {
                current.itself = mm[z];
                current.hash = hash_u64(mm[z], seed);
                current.p1 = (uint8_t)(i - m + 1);
                current.id = (*mm_count)++;
                if (nbases_since_last_break == k) ++kmer_count;
                if (nbases_since_last_break == k + 1) {
                    min_pos = p1 = 0;
                    for (size_t j = 0; j < W; ++j) {
                        if (buffer[j].hash < buffer[min_pos].hash) { min_pos = j; p1 = (uint32_t)min_pos; }
                    }
                    sks = 1;
                }
                if (nbases_since_last_break >= k + 1) {
                    if ((buf_pos % W) == min_pos) {
                        buffer[min_pos].p1 = (uint8_t)p1;
                        buffer[min_pos].size = (uint8_t)sks;
                        update_output(accumulator, &buffer[min_pos]);
                        sks = 0;
                        find_brand_new_min = 1;
                    } else if (current.hash < buffer[min_pos].hash) {
                        buffer[min_pos].p1 = (uint8_t)p1;
                        buffer[min_pos].size = (uint8_t)sks;
                        update_output(accumulator, &buffer[min_pos]);
                        sks = 0;
                        p1 = k - m;
                        min_pos = buf_pos;
                    }
                    ++sks;
                    ++kmer_count;
                }
                buffer[buf_pos++] = current;
                buf_pos %= W;
                if (find_brand_new_min) {
                    find_brand_new_min = 0;
                    min_pos = buf_pos;
                    p1 = 0;
                    uint32_t tmp = 1;
                    for (size_t j = (buf_pos + 1) % W; j < W; ++j) {
                        if (buffer[min_pos].hash > buffer[j].hash) { min_pos = j; p1 = tmp; }
                        ++tmp;
                    }
                    for (size_t j = 0; j <= buf_pos; ++j) {
                        if (buffer[min_pos].hash > buffer[j].hash) { min_pos = j; p1 = tmp; }
                        ++tmp;
                    }
                }
            }
        } else {
            nbases_since_last_break = 0;
            if (min_pos < W) {
                buffer[min_pos].p1 = (uint8_t)p1;
                buffer[min_pos].size = (uint8_t)sks;
                update_output(accumulator, &buffer[min_pos]);
            }
            sks = 0;
            min_pos = W;
            buf_pos = 0;
        }
    }
    if (nbases_since_last_break == k) {
        min_pos = p1 = 0;
        sks = 1;
        for (size_t j = 0; j < W; ++j) {
            if (buffer[j].hash < buffer[min_pos].hash) { min_pos = j; p1 = (uint32_t)min_pos; }
        }
    }
    if (min_pos < W) {
        buffer[min_pos].p1 = (uint8_t)p1;
        buffer[min_pos].size = (uint8_t)sks;
        update_output(accumulator, &buffer[min_pos]);
        sks = 1;
    }
    free(buffer);
    return kmer_count;
}

/* ---------------- minimizer::get_colliding_kmers  include/minimizer.hpp:172-319 -------------- */
typedef struct { u128* v; size_t n, cap; } kmvec;
static void km_push(kmvec* a, u128 x) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 1024; a->v = realloc(a->v, a->cap * sizeof *a->v); }
    a->v[a->n++] = x;
}
static void get_colliding_kmers(const char* contig, size_t contig_size, uint32_t k, uint32_t m, uint64_t seed,
                                const uint64_t* ids, size_t* itr, size_t stop, uint64_t* mm_count, kmvec* accumulator) {
    const size_t W = k - m + 1;
    mm_record_t* mm_buffer = calloc(W, sizeof *mm_buffer);
    kmvec km_buffer = {0, 0, 0};
    size_t mm_buf_pos = 0, min_pos = W;
    mm_record_t current; memset(&current, 0, sizeof current);
    uint64_t mm_shift = 2 * (m - 1);
    uint64_t mm_mask = m == 32 ? ~(uint64_t)0 : ((1ULL << (2 * m)) - 1);
    uint64_t km_shift = 2 * (k - 1);
    u128 km_mask = (((u128)1) << (2 * k)) - 1;
    uint64_t mm[2] = {0, 0};
    u128 km[2] = {0, 0};
    uint64_t nbases_since_last_break = 0;
    uint32_t sks = 0;
    const uint8_t z = 0;
    int find_brand_new_min = 0;
#define FLUSH_IF_COLLIDING()                                            \
    do {                                                                \
        if (*itr != stop && ids[*itr] == mm_buffer[min_pos].id) {       \
            for (size_t q = 0; q < km_buffer.n; ++q) km_push(accumulator, km_buffer.v[q]); \
            ++*itr;                                                     \
        }                                                               \
    } while (0)
    for (uint64_t i = 0; i < contig_size; ++i) {
        int c = nt4((uint8_t)contig[i]);
        if (c < 4) {
            mm[0] = (mm[0] << 2 | (uint64_t)c) & mm_mask;
            mm[1] = (mm[1] >> 2) | (3ULL ^ (uint64_t)c) << mm_shift;
            km[0] = (km[0] << 2 | (u128)c) & km_mask;
            km[1] = (km[1] >> 2) | (((u128)3 ^ (u128)c) << km_shift);
            ++nbases_since_last_break;
            if (nbases_since_last_break >= m) {
                current.itself = hash_u64(mm[z], seed);
                current.id = (*mm_count)++;
                if (nbases_since_last_break == k + 1) {
                    min_pos = 0;
                    for (size_t j = 0; j < W; ++j)
                        if (mm_buffer[j].itself < mm_buffer[min_pos].itself) min_pos = j;
                    sks = 1;
                }
                if (nbases_since_last_break >= k + 1) {
                    if ((mm_buf_pos % W) == min_pos || current.itself < mm_buffer[min_pos].itself) {
                        FLUSH_IF_COLLIDING();
                        km_buffer.n = 0;
                        if ((mm_buf_pos % W) == min_pos) find_brand_new_min = 1;
                        else if (current.itself < mm_buffer[min_pos].itself) min_pos = mm_buf_pos;
                        sks = 0;
                    }
                    ++sks;
                }
                mm_buffer[mm_buf_pos++] = current;
                mm_buf_pos %= W;
                if (nbases_since_last_break >= k) km_push(&km_buffer, km[z]);
                if (find_brand_new_min) {
                    find_brand_new_min = 0;
                    min_pos = mm_buf_pos;
                    for (size_t j = (mm_buf_pos + 1) % W; j < W; ++j)
                        if (mm_buffer[min_pos].itself > mm_buffer[j].itself) min_pos = j;
                    for (size_t j = 0; j <= mm_buf_pos; ++j)
                        if (mm_buffer[min_pos].itself > mm_buffer[j].itself) min_pos = j;
                }
            }
        } else {
            nbases_since_last_break = 0;
            if (min_pos < W) FLUSH_IF_COLLIDING();
            km_buffer.n = 0;
            min_pos = W;
            sks = 0;
            mm_buf_pos = 0;
        }
    }
    if (nbases_since_last_break == k) {
        min_pos = 0;
        sks = 1;
        for (size_t j = 0; j < W; ++j)
            if (mm_buffer[j].itself < mm_buffer[min_pos].itself) min_pos = j;
    }
    if (min_pos < W) FLUSH_IF_COLLIDING();
#undef FLUSH_IF_COLLIDING
    (void)sks;
    free(mm_buffer);
    free(km_buffer.v);
}

/* ---------------- minimizer::classify  src/minimizer.cpp:5-50 ---------------- */
typedef struct { uint64_t itself; uint8_t p1, size; } mm_triplet_t;
static int cmp_rec(const void* a, const void* b) {
    const mm_record_t *x = a, *y = b;
    if (x->itself != y->itself) return x->itself < y->itself ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id); /* ties are unordered in the reference; ids make it stable */
}
static int cmp_trip(const void* a, const void* b) {
    const mm_triplet_t *x = a, *y = b;
    return x->itself < y->itself ? -1 : (x->itself > y->itself);
}
static void classify(recvec* minimizers, mm_triplet_t** uniq_out, size_t* nuniq_out, uint64_t** coll_out, size_t* ncoll_out) {
    qsort(minimizers->v, minimizers->n, sizeof *minimizers->v, cmp_rec);
    mm_triplet_t* uniq = malloc((minimizers->n ? minimizers->n : 1) * sizeof *uniq);
    uint64_t* coll = malloc((minimizers->n ? minimizers->n : 1) * 8);
    size_t nu = 0, nc = 0;
    size_t start = 0, stop = minimizers->n;
    mm_record_t prev; memset(&prev, 0, sizeof prev);
    while (start != stop) {
        if (prev.size != 0) {
            if (prev.itself == minimizers->v[start].itself) {
                prev.p1 = prev.size = 0;
                uniq[nu].itself = prev.itself; uniq[nu].p1 = 0; uniq[nu].size = 0; ++nu;
                coll[nc++] = prev.id;
                while (start != stop && minimizers->v[start].itself == prev.itself) { coll[nc++] = minimizers->v[start].id; ++start; }
            } else {
                uniq[nu].itself = prev.itself; uniq[nu].p1 = prev.p1; uniq[nu].size = prev.size; ++nu;
                prev = minimizers->v[start];
                ++start;
            }
        } else {
            prev = minimizers->v[start];
            ++start;
        }
    }
    if (prev.size) { uniq[nu].itself = prev.itself; uniq[nu].p1 = prev.p1; uniq[nu].size = prev.size; ++nu; }
    qsort(coll, nc, 8, cmp_u64);
    *uniq_out = uniq; *nuniq_out = nu; *coll_out = coll; *ncoll_out = nc;
}

/* quartet_wtree_builder::push_back  src/quartet_wtree.cpp:12-41 */
typedef struct { uint8_t* root; uint8_t* lr; uint8_t* mn; size_t nroot, nlr, nmn; } wtb_t;
static void wtb_push(wtb_t* b, int type) {
    switch (type) {
        case T_MAXIMAL: b->root[b->nroot++] = 1; b->mn[b->nmn++] = 0; break;
        case T_LEFT: b->root[b->nroot++] = 0; b->lr[b->nlr++] = 0; break;
        case T_RIGHT_OR_COLLISION: b->root[b->nroot++] = 0; b->lr[b->nlr++] = 1; break;
        case T_NONE: b->root[b->nroot++] = 1; b->mn[b->nmn++] = 1; break;
    }
}
static void rs_from_bytes(rsbv* v, const uint8_t* bits, size_t n) {
    memset(v, 0, sizeof *v);
    bitvec_init(&v->bv, n);
    for (size_t i = 0; i < n; ++i) if (bits[i]) bv_set(&v->bv, i);
    rs_build(v);
}

int orc_build(const char* const* contigs, const size_t* lengths, size_t n_contigs, int k_, int m_, uint64_t mm_seed,
              double c, int kind, int kmer_bits, const char* out_path) {
    const uint32_t k = (uint32_t)k_, m = (uint32_t)m_;
    if (kmer_bits != 64 && kmer_bits != 128) { FAIL("kmer_bits must be 64 or 128"); return -1; }
    if (k > (uint32_t)(kmer_bits / 2 - 1) || m > k || m > 32 || m == 0) { FAIL("bad k/m"); return -1; }
    const double alpha = 0.94;       /* src/partitioned_mphf.cpp:25,48 */
    const uint64_t pthash_seed = 1;  /* constants::default_pthash_seed, include/constants.hpp:17 */
    orc_index* x = calloc(1, sizeof *x);
    x->kind = kind; x->kmer_bits = kmer_bits; x->k = (uint8_t)k; x->m = (uint8_t)m; x->mm_seed = mm_seed;
    int rc = -1;

    /* Part 1  src/partitioned_mphf.cpp:61-83 */
    recvec all = {0, 0, 0};
    uint64_t id = 0;
    for (size_t i = 0; i < n_contigs; ++i) x->nkmers += from_string(contigs[i], lengths[i], k, m, mm_seed, &id, &all);

    /* Part 2  :85-91 */
    mm_triplet_t* uniq; size_t nuniq; uint64_t* coll_ids; size_t ncoll;
    classify(&all, &uniq, &nuniq, &coll_ids, &ncoll);
    free(all.v);
    x->distinct_minimizers = nuniq;
    uint64_t* hashes = malloc((nuniq ? nuniq : 1) * 8);
    for (size_t i = 0; i < nuniq; ++i) hashes[i] = hash_u64(uniq[i].itself, pthash_seed);
    if (phf_build(&x->minimizer_order, hashes, nuniq, pthash_seed, c, alpha)) { free(hashes); goto out; }
    free(hashes);

    /* Part 3  :93-108 */
    for (size_t i = 0; i < nuniq; ++i) uniq[i].itself = orc_minimizer_order(x, uniq[i].itself);
    qsort(uniq, nuniq, sizeof *uniq, cmp_trip);

    if (kind == ORC_PARTITIONED) { /* build_inverted_index  src/partitioned_mphf.cpp:163-268 */
        wtb_t wtb;
        wtb.root = calloc(nuniq + 1, 1); wtb.lr = calloc(nuniq + 1, 1); wtb.mn = calloc(nuniq + 1, 1);
        wtb.nroot = wtb.nlr = wtb.nmn = 0;
        uint64_t *left_positions = malloc((nuniq + 1) * 8), *right_or_collision_sizes = malloc((nuniq + 1) * 8);
        uint64_t *none_sizes = malloc((nuniq + 1) * 8), *none_positions = malloc((nuniq + 1) * 8);
        size_t nl = 0, nr = 0, nn = 0;
        uint64_t universe = 0;
        for (size_t i = 0; i < nuniq; ++i) {
            mm_triplet_t mmv = uniq[i];
            if (mmv.size == 0) {
                wtb_push(&wtb, T_RIGHT_OR_COLLISION);
                right_or_collision_sizes[nr++] = 0;
            } else if (mmv.p1 == k - m) {
                if (mmv.size == k - m + 1) { wtb_push(&wtb, T_MAXIMAL); ++x->n_maximal; }
                else { wtb_push(&wtb, T_RIGHT_OR_COLLISION); right_or_collision_sizes[nr++] = mmv.size; universe += mmv.size; }
            } else if (mmv.p1 == mmv.size - 1) {
                wtb_push(&wtb, T_LEFT);
                left_positions[nl++] = (uint64_t)mmv.p1 + 1;
                universe += (uint64_t)mmv.p1 + 1;
            } else {
                wtb_push(&wtb, T_NONE);
                none_positions[nn] = mmv.p1;
                none_sizes[nn] = mmv.size;
                ++nn;
                universe += (uint64_t)mmv.p1 + mmv.size;
            }
        }
        rs_from_bytes(&x->root, wtb.root, wtb.nroot);
        rs_from_bytes(&x->left_right, wtb.lr, wtb.nlr);
        rs_from_bytes(&x->max_none, wtb.mn, wtb.nmn);
        x->right_coll_sizes_start = nl;
        x->none_sizes_start = x->right_coll_sizes_start + nr;
        x->none_pos_start = x->none_sizes_start + nn;
        /* append_iterator + cumulative_iterator: running sum over the four arrays in order */
        uint64_t total = x->none_pos_start + nn;
        uint64_t* cum = malloc((total ? total : 1) * 8);
        uint64_t s = 0, t = 0;
        for (size_t i = 0; i < nl; ++i) { s += left_positions[i]; cum[t++] = s; }
        for (size_t i = 0; i < nr; ++i) { s += right_or_collision_sizes[i]; cum[t++] = s; }
        for (size_t i = 0; i < nn; ++i) { s += none_sizes[i]; cum[t++] = s; }
        for (size_t i = 0; i < nn; ++i) { s += none_positions[i]; cum[t++] = s; }
        ef_encode(&x->sizes_and_positions, cum, total, universe, 1);
        free(cum); free(left_positions); free(right_or_collision_sizes); free(none_sizes); free(none_positions);
        free(wtb.root); free(wtb.lr); free(wtb.mn);
    } else { /* build_pos_index / build_size_index  src/unpartitioned_mphf.cpp:93-112,158-176 */
        uint64_t pos_sum = 0, size_sum = 0;
        uint64_t *cp = malloc((nuniq ? nuniq : 1) * 8), *cs = malloc((nuniq ? nuniq : 1) * 8);
        for (size_t i = 0; i < nuniq; ++i) { pos_sum += uniq[i].p1; size_sum += uniq[i].size; cp[i] = pos_sum; cs[i] = size_sum; }
        ef_encode(&x->positions, cp, nuniq, pos_sum, 1);
        ef_encode(&x->sizes, cs, nuniq, size_sum, 1);
        x->num_kmers_in_main_index = size_sum; /* == sizes.access(sizes.size()-1) */
        free(cp); free(cs);
    }

    /* Part 4  src/partitioned_mphf.cpp:110-137 */
    {
        kmvec unbucketable = {0, 0, 0};
        size_t itr = 0;
        id = 0;
        for (size_t i = 0; i < n_contigs; ++i)
            get_colliding_kmers(contigs[i], lengths[i], k, m, mm_seed, coll_ids, &itr, ncoll, &id, &unbucketable);
        uint64_t* fh = malloc((unbucketable.n ? unbucketable.n : 1) * 8);
        for (size_t i = 0; i < unbucketable.n; ++i) fh[i] = fallback_hash(unbucketable.v[i], kmer_bits, pthash_seed);
        int brc = phf_build(&x->fallback_kmer_order, fh, unbucketable.n, pthash_seed, c, alpha);
        free(fh); free(unbucketable.v);
        if (brc) goto out;
    }

    /* essentials::save  src/build.cpp:28 */
    {
        wr_t w = {0, 0, 0};
        wr_raw(&w, &x->k, 1); wr_raw(&w, &x->m, 1);
        wr_u64(&w, x->mm_seed); wr_u64(&w, x->nkmers); wr_u64(&w, x->distinct_minimizers);
        if (kind == ORC_PARTITIONED) {
            wr_u64(&w, x->n_maximal); wr_u64(&w, x->right_coll_sizes_start); wr_u64(&w, x->none_sizes_start); wr_u64(&w, x->none_pos_start);
            wr_phf(&w, &x->minimizer_order);
            wr_rs(&w, &x->root); wr_rs(&w, &x->left_right); wr_rs(&w, &x->max_none);
            wr_ef(&w, &x->sizes_and_positions);
            wr_phf(&w, &x->fallback_kmer_order);
        } else {
            wr_u64(&w, x->num_kmers_in_main_index);
            wr_phf(&w, &x->minimizer_order);
            wr_ef(&w, &x->positions); wr_ef(&w, &x->sizes);
            wr_phf(&w, &x->fallback_kmer_order);
        }
        FILE* f = fopen(out_path, "wb");
        if (!f) { FAIL("cannot write %s", out_path); free(w.p); goto out; }
        fwrite(w.p, 1, w.len, f);
        fclose(f);
        free(w.p);
    }
    rc = 0;
out:
    free(uniq); free(coll_ids);
    orc_free(x);
    return rc;
}


/* ==========================================================================================
 * Competitor harness: a plain PTHash MPHF over whole k-mers, src/ptbb/ptbb.hpp:16-34,
 * src/ptbb/ptbb_build.cpp:83-99, src/ptbb/ptbb_query.cpp:40-55.  (The BBHash half of the reference's harness is not restated.)
 * ======================================================================================== */
struct orc_ptbb { phf f; uint32_t k; int kmer_bits; };

/* ptbb::PTHasher::hash (ptbb.hpp:16-23): hash128{Murmur(&val, sizeof val, seed), Murmur(&val, sizeof val, ~seed)} */
static void ptbb_hash(u128 kmer, int kmer_bits, uint64_t seed, uint64_t* first, uint64_t* second) {
    if (kmer_bits == 64) {
        uint64_t v = (uint64_t)kmer;
        *first = orc_murmur64a(&v, 8, seed);
        *second = orc_murmur64a(&v, 8, ~seed);
    } else {
        *first = orc_murmur64a(&kmer, 16, seed);
        *second = orc_murmur64a(&kmer, 16, ~seed);
    }
}

/* the k-mers ptbb_file_itr yields for one record (ptbb.hpp:108-131): every window of k valid bases, forward strand.
 * Defined divergence: when a record's LAST stretch of valid bases is shorter than k and follows a non-ACGT byte, the reference's operator++ runs
 * into the record's end with nbases_since_last_break < k, returns with has_next() still true, and the caller consumes *itr once more -- a chimeric k-mer (the km
 * register is not reset at a break: the last bases before the break + the tail).  Neither this restatement nor the product yields that key; ACGT-only input, and
 * any input whose records end in a stretch of >= k valid bases, is unaffected. */
typedef void (*kmer_fn)(u128 kmer, void* ctx);
static uint64_t ptbb_each_kmer(const char* s, size_t len, uint32_t k, kmer_fn fn, void* ctx) {
    const u128 km_mask = (((u128)1) << (2 * k)) - 1;
    u128 km = 0;
    uint64_t since_break = 0, n = 0;
    for (size_t i = 0; i < len; ++i) {
        int c = nt4((uint8_t)s[i]);
        if (c < 4) {
            km = ((km << 2) | (u128)c) & km_mask;
            if (++since_break >= k) { if (fn) fn(km, ctx); ++n; }
        } else {
            since_break = 0;
        }
    }
    return n;
}
typedef struct { uint64_t* first; uint64_t* second; uint64_t n; int kmer_bits; uint64_t seed; } ptbb_acc;
static void ptbb_collect(u128 kmer, void* ctx) {
    ptbb_acc* a = ctx;
    ptbb_hash(kmer, a->kmer_bits, a->seed, &a->first[a->n], &a->second[a->n]);
    ++a->n;
}

/* ptbb_build -i <fasta> -k k -p <out> [-c c]  (ptbb_build.cpp:83-99: seed = default_pthash_seed = 1, alpha = 0.94, minimal) */
int orc_ptbb_build(const char* const* contigs, const size_t* lengths, size_t n_contigs, int k, int kmer_bits, double c, const char* out_path) {
    if ((kmer_bits != 64 && kmer_bits != 128) || k < 1 || k > kmer_bits / 2 - 1) { FAIL("ptbb: k out of range for kmer_bits"); return -1; }
    uint64_t total = 0;
    for (size_t i = 0; i < n_contigs; ++i) total += ptbb_each_kmer(contigs[i], lengths[i], (uint32_t)k, NULL, NULL);
    ptbb_acc a = {malloc((total ? total : 1) * 8), malloc((total ? total : 1) * 8), 0, kmer_bits, 1};
    for (size_t i = 0; i < n_contigs; ++i) ptbb_each_kmer(contigs[i], lengths[i], (uint32_t)k, ptbb_collect, &a);
    phf f;
    int rc = phf_build2(&f, a.first, a.second, a.n, 1, c, 0.94);
    free(a.first); free(a.second);
    if (rc) return -1;
    wr_t w = {0, 0, 0};
    wr_phf(&w, &f);
    FILE* fp = fopen(out_path, "wb");
    if (!fp || fwrite(w.p, 1, w.len, fp) != w.len) { if (fp) fclose(fp); free(w.p); phf_free(&f); FAIL("cannot write %s", out_path); return -1; }
    fclose(fp);
    free(w.p);
    phf_free(&f);
    return 0;
}

orc_ptbb* orc_ptbb_load(const char* path, int k, int kmer_bits) {
    FILE* fp = fopen(path, "rb");
    if (!fp) { FAIL("cannot open %s", path); return NULL; }
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    uint8_t* buf = malloc(n > 0 ? (size_t)n : 1);
    if (fread(buf, 1, (size_t)n, fp) != (size_t)n) { fclose(fp); free(buf); FAIL("short read"); return NULL; }
    fclose(fp);
    orc_ptbb* h = calloc(1, sizeof *h);
    h->k = (uint32_t)k; h->kmer_bits = kmer_bits;
    rd_t r = {buf, (size_t)n, 0, 0};
    rd_phf(&r, &h->f);
    const int bad = r.bad || r.off != r.len;
    free(buf);
    if (bad) { FAIL("malformed single_phf file"); phf_free(&h->f); free(h); return NULL; }
    return h;
}
void orc_ptbb_free(orc_ptbb* h) { if (h) { phf_free(&h->f); free(h); } }
uint64_t orc_ptbb_num_keys(const orc_ptbb* h) { return h->f.num_keys; }

/* pthash_mphf(*kmer_itr)  (ptbb_query.cpp:47) on packed k-mers */
void orc_ptbb_query_packed(const orc_ptbb* h, const uint64_t* kmers, size_t n, uint64_t* out) {
    for (size_t i = 0; i < n; ++i) {
        u128 kmer = h->kmer_bits == 64 ? (u128)kmers[i] : ((u128)kmers[2 * i] | ((u128)kmers[2 * i + 1] << 64));
        uint64_t a, b;
        ptbb_hash(kmer, h->kmer_bits, h->f.seed, &a, &b);
        out[i] = phf_position2(&h->f, a, b);
    }
}
typedef struct { const orc_ptbb* h; uint64_t* out; size_t cap, n; } ptbb_q;
static void ptbb_query_one(u128 kmer, void* ctx) {
    ptbb_q* q = ctx;
    uint64_t a, b;
    ptbb_hash(kmer, q->h->kmer_bits, q->h->f.seed, &a, &b);
    if (q->n < q->cap) q->out[q->n] = phf_position2(&q->h->f, a, b);
    ++q->n;
}
/* the loop of ptbb_query.cpp:46-51 over one record: one code per window of k valid bases; returns how many */
int64_t orc_ptbb_query_contig(const orc_ptbb* h, const char* contig, size_t length, uint64_t* out, size_t cap) {
    ptbb_q q = {h, out, cap, 0};
    ptbb_each_kmer(contig, length, h->k, ptbb_query_one, &q);
    return (int64_t)q.n;
}
