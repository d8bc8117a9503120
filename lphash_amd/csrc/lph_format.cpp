// lph_format.cpp -- parser for the reference's serialized mphf / mphf_alt (see lph_format.hpp).
#include "lph_format.hpp"

#include <cstdio>
#include <cstring>

#include "../../include/lphq.h"

namespace lphq {

u128 compute_M_u64(uint64_t d) {  // pthash fastmod::computeM_u64
    if (d == 0) return 0;
    u128 M = ~(u128)0;
    M /= d;
    M += 1;
    return M;
}

uint64_t murmur64a_u64(uint64_t v, uint64_t seed) {  // MurmurHash64A, len = 8
    const uint64_t m = 0xc6a4a7935bd1e995ULL;
    uint64_t h = seed ^ (8 * m);
    uint64_t k = v * m;
    k ^= k >> 47;
    k *= m;
    h ^= k;
    h *= m;
    h ^= h >> 47;
    h *= m;
    h ^= h >> 47;
    return h;
}

namespace {

inline uint64_t words_for(uint64_t bits) { return (bits + 63) / 64; }

struct Reader {
    const uint8_t* p;
    size_t len, off = 0;
    bool bad = false;
    void raw(void* dst, size_t n) {
        if (n == 0) return;  // (an empty vector's data() may be null: not a valid memcpy argument even for 0 bytes)
        if (bad || n > len - off) {
            bad = true;
            memset(dst, 0, n);
            return;
        }
        memcpy(dst, p + off, n);
        off += n;
    }
    uint64_t u64() {
        uint64_t v;
        raw(&v, 8);
        return v;
    }
    template <typename T>
    void vec(std::vector<T>& v) {
        uint64_t n = u64();
        if (bad || n > (len - off) / sizeof(T)) {
            bad = true;
            return;
        }
        v.resize(n);
        raw(v.data(), n * sizeof(T));
    }
    void read(BitVector& b) {
        b.size = u64();
        vec(b.bits);
    }
    void read(CompactVector& c) {
        c.size = u64();
        c.width = u64();
        c.mask = u64();
        vec(c.bits);
    }
    void read(DArray1& d) {
        d.positions = u64();
        vec(d.block_inventory);
        vec(d.subblock_inventory);
        vec(d.overflow_positions);
    }
    void read(EliasFano& e) {
        read(e.high);
        read(e.d1);
        read(e.low);
    }
    void read(RsBitVector& r) {
        read(r.bv);
        vec(r.block_rank_pairs);
        vec(r.select_hints);
    }
    void read(Dictionary& d) {
        read(d.ranks);
        read(d.dict);
    }
    void read(SinglePhf& f) {
        f.seed = u64();
        f.num_keys = u64();
        f.table_size = u64();
        raw(&f.M, 16);
        f.num_dense = u64();
        f.num_sparse = u64();
        raw(&f.M_dense, 16);
        raw(&f.M_sparse, 16);
        read(f.front);
        read(f.back);
        read(f.free_slots);
    }
};

#define CHECK(cond, ...)                              \
    do {                                              \
        if (!(cond)) {                                \
            char b_[256];                             \
            snprintf(b_, sizeof b_, __VA_ARGS__);     \
            err = std::string(what) + ": " + b_;      \
            return false;                             \
        }                                             \
    } while (0)

bool check_bv(const BitVector& b, const char* what, std::string& err) {
    CHECK(b.bits.size() >= words_for(b.size), "bit_vector of %llu bits has only %zu words", (unsigned long long)b.size, b.bits.size());
    return true;
}
bool check_cv(const CompactVector& c, const char* what, std::string& err) {
    CHECK(c.width <= 64, "compact_vector width %llu", (unsigned long long)c.width);
    uint64_t mask = c.width == 64 ? ~0ULL : ((1ULL << c.width) - 1);
    CHECK(c.mask == mask, "compact_vector mask does not match width %llu", (unsigned long long)c.width);
    CHECK(c.width == 0 || c.size <= (uint64_t)c.bits.size() * 64 / c.width, "compact_vector too short");
    return true;
}
bool check_ef(const EliasFano& e, const char* what, std::string& err) {
    if (!check_bv(e.high, what, err) || !check_cv(e.low, what, err)) return false;
    CHECK(e.low.width < 64, "elias-fano low width %llu", (unsigned long long)e.low.width);
    CHECK(e.low.size <= e.high.size, "elias-fano: more values than high bits");
    return true;
}
uint64_t popcount_bv(const BitVector& b) {
    uint64_t n = 0, full = b.size / 64;
    for (uint64_t i = 0; i < full; ++i) n += __builtin_popcountll(b.bits[i]);
    if (b.size % 64) n += __builtin_popcountll(b.bits[full] & ((1ULL << (b.size % 64)) - 1));
    return n;
}
bool check_rs(const RsBitVector& r, const char* what, std::string& err) {
    if (!check_bv(r.bv, what, err)) return false;
    CHECK(r.block_rank_pairs.size() >= 2, "rank index missing");
    // rs_bit_vector.hpp:24  num_ones() = *(m_block_rank_pairs.end() - 2)
    CHECK(r.block_rank_pairs[r.block_rank_pairs.size() - 2] == popcount_bv(r.bv), "rank index disagrees with the bits");
    return true;
}
bool check_phf(const SinglePhf& f, const char* what, std::string& err) {
    if (!check_cv(f.front.ranks, what, err) || !check_cv(f.front.dict, what, err) || !check_cv(f.back.ranks, what, err) ||
        !check_cv(f.back.dict, what, err) || !check_ef(f.free_slots, what, err))
        return false;
    if (f.num_keys == 0) {
        CHECK(f.table_size == 0 || f.front.ranks.size + f.back.ranks.size == f.num_dense + f.num_sparse, "empty function with inconsistent buckets");
        return true;
    }
    CHECK(f.table_size >= f.num_keys, "table_size < num_keys");
    CHECK(f.M == compute_M_u64(f.table_size), "m_M != computeM(table_size)");
    CHECK(f.num_dense > 0 && f.num_sparse > 0, "empty bucket class");
    CHECK(f.M_dense == compute_M_u64(f.num_dense) && f.M_sparse == compute_M_u64(f.num_sparse), "bucketer M mismatch");
    CHECK(f.front.ranks.size + f.back.ranks.size == f.num_dense + f.num_sparse, "pilots cover %llu buckets, bucketer has %llu",
          (unsigned long long)(f.front.ranks.size + f.back.ranks.size), (unsigned long long)(f.num_dense + f.num_sparse));
    CHECK(f.free_slots.size() == f.table_size - f.num_keys, "free slots: %llu entries for %llu holes", (unsigned long long)f.free_slots.size(),
          (unsigned long long)(f.table_size - f.num_keys));
    for (const Dictionary* d : {&f.front, &f.back}) {
        CHECK(d->ranks.size == 0 || (d->ranks.width >= 1 && d->dict.width >= 1), "zero-width pilot dictionary");
        for (uint64_t i = 0; i < d->ranks.size; ++i) CHECK(d->ranks.access(i) < d->dict.size, "pilot codeword out of range");
    }
    return true;
}

}  // namespace

bool EliasFano::decode(std::vector<uint64_t>& out, std::string& err) const {
    const uint64_t n = low.size, l = low.width;
    out.clear();
    out.reserve(n);
    uint64_t i = 0;
    for (uint64_t w = 0; w < high.bits.size() && i < n; ++w) {
        uint64_t word = high.bits[w];
        while (word && i < n) {
            uint64_t pos = (w << 6) + (uint64_t)__builtin_ctzll(word);
            word &= word - 1;
            if (pos >= high.size) break;
            out.push_back(((pos - i) << l) | low.access(i));
            ++i;
        }
    }
    if (i != n) {
        err = "elias-fano: fewer set high bits than stored values";
        return false;
    }
    for (uint64_t j = 1; j < n; ++j)
        if (out[j] < out[j - 1]) {
            err = "elias-fano: sequence not monotone";
            return false;
        }
    return true;
}

bool parse_lph(const uint8_t* buf, size_t len, int kind, int kmer_bits, LphFile& x, std::string& err) {
    const char* what = "lph";
    CHECK(kind == LPHQ_PARTITIONED || kind == LPHQ_UNPARTITIONED, "kind must be LPHQ_PARTITIONED or LPHQ_UNPARTITIONED");
    CHECK(kmer_bits == 64 || kmer_bits == 128, "kmer_bits must be 64 or 128");
    Reader r{buf, len};
    x.kind = kind;
    x.kmer_bits = kmer_bits;
    r.raw(&x.k, 1);
    r.raw(&x.m, 1);
    x.mm_seed = r.u64();
    x.nkmers = r.u64();
    x.distinct_minimizers = r.u64();
    if (kind == LPHQ_PARTITIONED) {
        x.n_maximal = r.u64();
        x.right_coll_sizes_start = r.u64();
        x.none_sizes_start = r.u64();
        x.none_pos_start = r.u64();
        size_t at = r.off;
        auto part = [&](int i) {
            x.part_bytes[i] += r.off - at;
            at = r.off;
        };
        r.read(x.minimizer_order);
        part(0);
        r.read(x.root);
        r.read(x.left_right);
        r.read(x.max_none);
        part(1);
        r.read(x.sizes_and_positions);
        part(2);
        x.part_bytes[2] += 4 * 8;  // n_maximal and the three region starts are counted with the arrays (partitioned_mphf.cpp:344-347)
        r.read(x.fallback_kmer_order);
        part(3);
    } else {
        x.num_kmers_in_main_index = r.u64();
        size_t at = r.off;
        auto part = [&](int i) {
            x.part_bytes[i] += r.off - at;
            at = r.off;
        };
        r.read(x.minimizer_order);
        part(0);
        r.read(x.positions);
        r.read(x.sizes);
        part(2);
        r.read(x.fallback_kmer_order);
        part(3);
    }
    CHECK(!r.bad, "truncated file (needed more than %zu bytes)", len);
    CHECK(r.off == len, "%zu trailing bytes: wrong kind (query-p vs query-u) or not an LPHash file", len - r.off);
    x.bytes_read = r.off;
    // constants::max_k = sizeof(kmer_t)*8/2 - 1 (include/constants.hpp:16); m <= 32 (mphf_utils.hpp:119)
    CHECK(x.k >= 1 && x.k <= kmer_bits / 2 - 1, "k=%u does not fit a %d-bit kmer_t (use kmer_bits=128 for k>31)", x.k, kmer_bits);
    CHECK(x.m >= 1 && x.m <= x.k && x.m <= 32, "m=%u out of range for k=%u", x.m, x.k);

    what = "minimizer_order";
    if (!check_phf(x.minimizer_order, what, err)) return false;
    CHECK(x.minimizer_order.num_keys == x.distinct_minimizers, "num_keys != distinct_minimizers");
    what = "fallback_kmer_order";
    if (!check_phf(x.fallback_kmer_order, what, err)) return false;

    const uint64_t W = (uint64_t)x.k - x.m + 1;
    if (kind == LPHQ_PARTITIONED) {
        what = "quartet_wtree";
        if (!check_rs(x.root, what, err) || !check_rs(x.left_right, what, err) || !check_rs(x.max_none, what, err)) return false;
        CHECK(x.root.bv.size == x.distinct_minimizers, "root has %llu symbols for %llu minimizers", (unsigned long long)x.root.bv.size,
              (unsigned long long)x.distinct_minimizers);
        uint64_t ones = popcount_bv(x.root.bv);
        CHECK(x.max_none.bv.size == ones && x.left_right.bv.size == x.root.bv.size - ones, "child sizes do not match the root");
        uint64_t n_none = popcount_bv(x.max_none.bv), n_right = popcount_bv(x.left_right.bv);
        uint64_t n_left = x.left_right.bv.size - n_right;
        CHECK(x.n_maximal == ones - n_none, "n_maximal mismatch");
        CHECK(x.right_coll_sizes_start == n_left && x.none_sizes_start == n_left + n_right && x.none_pos_start == n_left + n_right + n_none,
              "region starts do not match the wavelet tree census");
        what = "sizes_and_positions";
        if (!check_ef(x.sizes_and_positions, what, err)) return false;
        uint64_t expect = x.none_pos_start + n_none;
        CHECK(x.sizes_and_positions.size() == (expect ? expect + 1 : 0), "%llu entries, expected %llu", (unsigned long long)x.sizes_and_positions.size(),
              (unsigned long long)(expect ? expect + 1 : 0));
        (void)W;
    } else {
        what = "positions/sizes";
        if (!check_ef(x.positions, what, err) || !check_ef(x.sizes, what, err)) return false;
        uint64_t expect = x.distinct_minimizers ? x.distinct_minimizers + 1 : 0;
        CHECK(x.positions.size() == expect && x.sizes.size() == expect, "entry count mismatch");
    }
    return true;
}

bool parse_pthash(const uint8_t* buf, size_t len, int k, int kmer_bits, LphFile& x, std::string& err) {
    const char* what = "pthash";
    CHECK(kmer_bits == 64 || kmer_bits == 128, "kmer_bits must be 64 or 128");
    CHECK(k >= 1 && k <= kmer_bits / 2 - 1, "k=%d does not fit a %d-bit kmer_t", k, kmer_bits);
    Reader r{buf, len};
    x = LphFile();
    x.kind = 2;
    x.kmer_bits = kmer_bits;
    x.k = x.m = (uint8_t)k;
    r.read(x.minimizer_order);
    CHECK(!r.bad, "truncated file (needed more than %zu bytes)", len);
    CHECK(r.off == len, "%zu trailing bytes: not a bare single_phf file", len - r.off);
    x.bytes_read = r.off;
    if (!check_phf(x.minimizer_order, what, err)) return false;
    x.nkmers = x.distinct_minimizers = x.minimizer_order.num_keys;
    return true;
}

}  // namespace lphq
