// builder.cpp -- in-memory, multi-threaded construction of a partitioned / unpartitioned LP-MPHF in the
// reference's serialization (lphq_build in include/lphq.h).
//
// What it reproduces (paths relative to the reference tree):
//   Part 1  super-k-mer records           include/minimizer.hpp:11-170  (from_string)
//   Part 2  classify + PTHash over minimizers  src/minimizer.cpp:5-50, src/partitioned_mphf.cpp:85-91,147-154
//   Part 3  inverted index                src/partitioned_mphf.cpp:163-268, src/unpartitioned_mphf.cpp:156-176
//   Part 4  fallback PTHash over k-mers of colliding minimizers  include/minimizer.hpp:172-319
//   save    essentials::save walk over visit()   src/build.cpp:28
// It is a position-parallel re-design (records are cut into chunks, super-k-mers are runs of windows
// that share the position of their leftmost minimum), not a translation of the reference's streaming
// state machines.  PTHash search = first valid pilot per bucket, buckets by non-increasing size
// (PTHash v1), run speculatively in parallel with in-order commit, so the result
// does not depend on the thread count.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "../../include/lphq.h"
#include "external_memory.hpp"
#include "extract.hpp"
#include "host_cpus.hpp"
#include "lph_format.hpp"
#include "search.hpp"

namespace lphq {
namespace {

typedef unsigned __int128 u128;

inline double resident_mb(bool anonymous_only = true) {  // (verbose builds only: where the memory goes; a memory-mapped input counts as file pages)
    double anon_kb = 0, file_kb = 0;
    FILE* f = fopen("/proc/self/status", "r");
    if (f) {
        char line[256];
        while (fgets(line, sizeof line, f)) {
            if (!strncmp(line, "RssAnon:", 8)) anon_kb = atof(line + 8);
            if (!strncmp(line, "RssFile:", 8)) file_kb = atof(line + 8);
        }
        fclose(f);
    }
    return (anonymous_only ? anon_kb : anon_kb + file_kb) / 1024.0;
}

inline uint64_t words_for(uint64_t bits) { return (bits + 63) / 64; }
inline unsigned msb64(uint64_t x) { return 63u - (unsigned)__builtin_clzll(x); }

inline uint8_t nt4(uint8_t c) {  // constants::seq_nt4_table, src/constants.cpp:5-13
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}

inline uint64_t mul128_u64(u128 lowbits, uint64_t d) {
    u128 bottom = ((lowbits & (u128)0xFFFFFFFFFFFFFFFFULL) * d) >> 64;
    u128 top = (lowbits >> 64) * d;
    return (uint64_t)((bottom + top) >> 64);
}
inline uint64_t fastmod_u64(uint64_t a, u128 M, uint64_t d) { return mul128_u64(M * a, d); }

// ---------------------------------------------------------------------------------------------
// serialized building blocks (writer side of lph_format.hpp)
// ---------------------------------------------------------------------------------------------
void bv_init(BitVector& b, uint64_t size) {
    b.size = size;
    b.bits.assign(words_for(size), 0);
}
inline void bv_set(BitVector& b, uint64_t i) { b.bits[i >> 6] |= 1ULL << (i & 63); }

void cv_init(CompactVector& c, uint64_t n, uint64_t w) {  // compact_vector::builder(n, w)
    c.size = n;
    c.width = w;
    c.mask = w == 64 ? ~0ULL : ((1ULL << w) - 1);
    c.bits.assign(words_for(n * w) + 1, 0);
}
inline void cv_set(CompactVector& c, uint64_t i, uint64_t v) {
    if (!c.width) return;
    uint64_t pos = i * c.width, blk = pos >> 6, sh = pos & 63;
    c.bits[blk] |= v << sh;
    if (sh + c.width > 64) c.bits[blk + 1] |= v >> (64 - sh);
}
void cv_build(CompactVector& c, const std::vector<uint64_t>& v) {  // compact_vector::build(begin, n)
    uint64_t mx = 0;
    for (uint64_t x : v) mx = std::max(mx, x);
    cv_init(c, v.size(), mx == 0 ? 1 : msb64(mx) + 1);
    for (size_t i = 0; i < v.size(); ++i) cv_set(c, i, v[i]);
}

void d1_build(DArray1& d, const BitVector& bv) {  // pthash::darray1::build
    d = DArray1();
    std::vector<uint64_t> cur;
    cur.reserve(1024);
    auto flush = [&] {
        if (cur.back() - cur.front() < (1u << 16)) {
            d.block_inventory.push_back((int64_t)cur.front());
            for (size_t i = 0; i < cur.size(); i += 32) d.subblock_inventory.push_back((uint16_t)(cur[i] - cur.front()));
        } else {
            d.block_inventory.push_back(-(int64_t)d.overflow_positions.size() - 1);
            for (uint64_t p : cur) d.overflow_positions.push_back(p);
            for (size_t i = 0; i < cur.size(); i += 32) d.subblock_inventory.push_back((uint16_t)-1);
        }
        cur.clear();
    };
    for (uint64_t w = 0; w < bv.bits.size(); ++w) {
        uint64_t word = bv.bits[w];
        while (word) {
            uint64_t pos = (w << 6) + (uint64_t)__builtin_ctzll(word);
            word &= word - 1;
            if (pos >= bv.size) break;
            cur.push_back(pos);
            d.positions += 1;
            if (cur.size() == 1024) flush();
        }
    }
    if (!cur.empty()) flush();
}

// Elias-Fano over a non-decreasing sequence given by get(i), i < n, universe u.
// leading_zero: lphash ef_sequence::encode (ef_sequence.hpp:37-75) prepends a 0; otherwise pthash::ef_sequence<false>.
template <typename Get>
void ef_encode(EliasFano& e, uint64_t n, uint64_t u, bool leading_zero, Get get, int threads = 1) {
    e = EliasFano();
    if (n == 0) return;
    const uint64_t nn = n + (leading_zero ? 1 : 0);
    const uint64_t l = (nn && u / nn) ? msb64(u / nn) : 0;
    bv_init(e.high, nn + (u >> l) + 1);
    cv_init(e.low, nn, l);
    const uint64_t low_mask = (1ULL << l) - 1;
    uint64_t at = 0;
    if (leading_zero) {
        bv_set(e.high, 0);
        at = 1;
    }
    if (threads <= 1 || n < (1u << 16)) {
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t v = get(i);
            if (l) cv_set(e.low, i + at, v & low_mask);
            bv_set(e.high, (v >> l) + i + at);
        }
    } else {  // elements are independent; neighbours share words, hence the atomic ORs
        uint64_t* high = e.high.bits.data();
        uint64_t* low = e.low.bits.data();
        const uint64_t chunk = 1u << 16;
        parallel_for((size_t)((n + chunk - 1) / chunk), threads, [&](size_t c) {
            for (uint64_t i = c * chunk, end = std::min(n, i + chunk); i < end; ++i) {
                const uint64_t v = get(i);
                if (l) {
                    const uint64_t pos = (i + at) * l, blk = pos >> 6, sh = pos & 63, x = v & low_mask;
                    __atomic_fetch_or(&low[blk], x << sh, __ATOMIC_RELAXED);
                    if (sh + l > 64) __atomic_fetch_or(&low[blk + 1], x >> (64 - sh), __ATOMIC_RELAXED);
                }
                const uint64_t hp = (v >> l) + i + at;
                __atomic_fetch_or(&high[hp >> 6], 1ULL << (hp & 63), __ATOMIC_RELAXED);
            }
        });
    }
    d1_build(e.d1, e.high);
}

// The same encoding when the values are cheaper to produce a chunk at a time than one by one (running sums): produce(c, emit) calls
// emit(i, value) for every i of chunk c = [c * chunk, min(n, (c + 1) * chunk)); chunks run in parallel.
template <typename Produce>
void ef_encode_chunked(EliasFano& e, uint64_t n, uint64_t u, bool leading_zero, uint64_t chunk, int threads, Produce produce) {
    e = EliasFano();
    if (n == 0) return;
    const uint64_t nn = n + (leading_zero ? 1 : 0);
    const uint64_t l = (nn && u / nn) ? msb64(u / nn) : 0;
    bv_init(e.high, nn + (u >> l) + 1);
    cv_init(e.low, nn, l);
    const uint64_t low_mask = (1ULL << l) - 1;
    uint64_t at = 0;
    if (leading_zero) {
        bv_set(e.high, 0);
        at = 1;
    }
    uint64_t* high = e.high.bits.data();
    uint64_t* low = e.low.bits.data();
    const std::function<void(uint64_t, uint64_t)> emit = [&](uint64_t i, uint64_t v) {  // neighbours share words, hence the atomic ORs
        if (l) {
            const uint64_t pos = (i + at) * l, blk = pos >> 6, sh = pos & 63, x = v & low_mask;
            __atomic_fetch_or(&low[blk], x << sh, __ATOMIC_RELAXED);
            if (sh + l > 64) __atomic_fetch_or(&low[blk + 1], x >> (64 - sh), __ATOMIC_RELAXED);
        }
        const uint64_t hp = (v >> l) + i + at;
        __atomic_fetch_or(&high[hp >> 6], 1ULL << (hp & 63), __ATOMIC_RELAXED);
    };
    parallel_for((size_t)((n + chunk - 1) / chunk), threads, [&](size_t c) { produce(c, emit); });
    d1_build(e.d1, e.high);
}

// Rank index of an rs_bit_vector, written from the layout the reader expects (the accessor side is
// include/rs_bit_vector.hpp:27-36,108): one (u64, u64) pair per 512-bit block -- the number of ones before the block, then seven
// 9-bit in-block prefix counts with the count before word j (j = 1..7) at bit (7 - j) * 9 -- and a closing (total ones, 0) pair.
// A trailing partial block is treated as zero-padded.  Three passes: per-block popcounts, an exclusive scan over the blocks,
// field packing; the first and last are independent per block.
void rs_build(RsBitVector& r, int threads = 1) {
    const std::vector<uint64_t>& words = r.bv.bits;
    const size_t nblocks = (words.size() + 7) / 8;
    std::vector<uint64_t>& pairs = r.block_rank_pairs;
    pairs.assign(2 * (nblocks + 1), 0);
    const size_t bchunk = 1 << 12;
    parallel_for((nblocks + bchunk - 1) / bchunk, threads, [&](size_t c) {
        for (size_t b = c * bchunk, e = std::min(nblocks, b + bchunk); b < e; ++b) {  // pass 1: ones inside block b, parked in the slot of its absolute rank
            const size_t end = std::min(words.size(), 8 * b + 8);
            uint64_t ones = 0;
            for (size_t w = 8 * b; w < end; ++w) ones += (uint64_t)__builtin_popcountll(words[w]);
            pairs[2 * b] = ones;
        }
    });
    uint64_t before = 0;
    for (size_t b = 0; b <= nblocks; ++b) {  // pass 2: exclusive scan; entry nblocks becomes the closing total
        const uint64_t ones = pairs[2 * b];
        pairs[2 * b] = before;
        before += ones;
    }
    parallel_for((nblocks + bchunk - 1) / bchunk, threads, [&](size_t c) {
        for (size_t b = c * bchunk, e = std::min(nblocks, b + bchunk); b < e; ++b) {  // pass 3: the seven in-block prefix counts
            uint64_t packed = 0, inside = 0;
            for (size_t j = 1; j < 8; ++j) {
                const size_t w = 8 * b + j - 1;
                if (w < words.size()) inside += (uint64_t)__builtin_popcountll(words[w]);
                packed |= inside << ((7 - j) * 9);
            }
            pairs[2 * b + 1] = packed;
        }
    });
    r.select_hints.clear();
}

// ---------------------------------------------------------------------------------------------
// essentials::saver
// ---------------------------------------------------------------------------------------------
struct Writer {  // straight to the file: no second copy of the index in memory
    FILE* f = nullptr;
    uint64_t bytes = 0;
    bool ok = true;
    ~Writer() {
        if (f) fclose(f);
    }
    bool open(const char* path) { return (f = fopen(path, "wb")) != nullptr; }
    bool close() {  // true: every byte went out
        if (f && fclose(f) != 0) ok = false;
        f = nullptr;
        return ok;
    }
    void raw(const void* p, size_t n) {
        if (n && fwrite(p, 1, n, f) != n) ok = false;
        bytes += n;
    }
    void u64(uint64_t v) { raw(&v, 8); }
    template <typename T>
    void vec(const std::vector<T>& v) {
        u64(v.size());
        if (!v.empty()) raw(v.data(), v.size() * sizeof(T));
    }
    void put(const BitVector& b) {
        u64(b.size);
        vec(b.bits);
    }
    void put(const CompactVector& c) {
        u64(c.size);
        u64(c.width);
        u64(c.mask);
        vec(c.bits);
    }
    void put(const DArray1& d) {
        u64(d.positions);
        vec(d.block_inventory);
        vec(d.subblock_inventory);
        vec(d.overflow_positions);
    }
    void put(const EliasFano& e) {
        put(e.high);
        put(e.d1);
        put(e.low);
    }
    void put(const RsBitVector& r) {
        put(r.bv);
        vec(r.block_rank_pairs);
        vec(r.select_hints);
    }
    void put(const SinglePhf& f) {
        u64(f.seed);
        u64(f.num_keys);
        u64(f.table_size);
        raw(&f.M, 16);
        u64(f.num_dense);
        u64(f.num_sparse);
        raw(&f.M_dense, 16);
        raw(&f.M_sparse, 16);
        put(f.front.ranks);
        put(f.front.dict);
        put(f.back.ranks);
        put(f.back.dict);
        put(f.free_slots);
    }
};

// ---------------------------------------------------------------------------------------------
// PTHash single_phf build
// ---------------------------------------------------------------------------------------------
inline uint64_t bucket_of(const SinglePhf& f, uint64_t hash) {  // skew_bucketer::bucket
    static const uint64_t T = 0x9999999999999800ULL;             // uint64_t(0.6 * UINT64_MAX)
    return hash < T ? fastmod_u64(hash, f.M_dense, f.num_dense) : f.num_dense + fastmod_u64(hash, f.M_sparse, f.num_sparse);
}

void dict_encode(Dictionary& d, const uint64_t* v, uint64_t n) {  // pthash::dictionary::encode
    std::vector<std::pair<uint64_t, uint64_t>> vf;  // (value, freq), ascending by value
    uint64_t vmax = 0;
    for (uint64_t i = 0; i < n; ++i) vmax = std::max(vmax, v[i]);
    const bool dense = vmax < (1ull << 26) && vmax <= 64 * n + 1024;  // pilots are small: count them in place of sorting them
    std::vector<uint32_t> count_or_rank;
    if (dense) {
        count_or_rank.assign(vmax + 1, 0);
        for (uint64_t i = 0; i < n; ++i) ++count_or_rank[v[i]];
        for (uint64_t x = 0; x <= vmax; ++x)
            if (count_or_rank[x]) vf.emplace_back(x, count_or_rank[x]);
    } else {
        std::vector<uint64_t> s(v, v + n);
        std::sort(s.begin(), s.end());
        for (uint64_t i = 0; i < n;) {
            uint64_t j = i;
            while (j < n && s[j] == s[i]) ++j;
            vf.emplace_back(s[i], j - i);
            i = j;
        }
    }
    // codewords by non-increasing frequency; ties (unordered in PTHash) by increasing value
    std::sort(vf.begin(), vf.end(), [](const auto& a, const auto& b) { return a.second != b.second ? a.second > b.second : a.first < b.first; });
    std::vector<uint64_t> dict(vf.size());
    for (size_t i = 0; i < vf.size(); ++i) dict[i] = vf[i].first;
    std::vector<uint64_t> ranks(n);
    if (dense) {
        for (size_t i = 0; i < vf.size(); ++i) count_or_rank[vf[i].first] = (uint32_t)i;
        for (uint64_t i = 0; i < n; ++i) ranks[i] = count_or_rank[v[i]];
    } else {
        std::vector<std::pair<uint64_t, uint64_t>> byval(vf.size());
        for (size_t i = 0; i < vf.size(); ++i) byval[i] = {vf[i].first, i};
        std::sort(byval.begin(), byval.end());
        for (uint64_t i = 0; i < n; ++i) ranks[i] = std::lower_bound(byval.begin(), byval.end(), std::make_pair(v[i], (uint64_t)0))->second;
    }
    cv_build(d.ranks, ranks);
    cv_build(d.dict, dict);
}

// Parameters of a single_phf over n keys (single_phf::build_in_internal_memory, skew_bucketer::init); false = too large for this builder.
bool phf_setup(SinglePhf& f, uint64_t n, uint64_t seed, double c, double alpha, uint64_t& num_buckets, std::string& err) {
    f = SinglePhf();
    f.seed = seed;
    f.num_keys = n;
    num_buckets = 0;
    if (n == 0) return true;  // all-zero function; the reference never builds one
    uint64_t table_size = (uint64_t)((double)n / alpha);
    if ((table_size & (table_size - 1)) == 0) table_size += 1;
    num_buckets = n < 2 ? 6 : (uint64_t)std::ceil((c * (double)n) / std::log2((double)n));
    f.table_size = table_size;
    f.M = compute_M_u64(table_size);
    f.num_dense = (uint64_t)(0.3 * (double)num_buckets);
    f.num_sparse = num_buckets - f.num_dense;
    f.M_dense = compute_M_u64(f.num_dense);
    f.M_sparse = compute_M_u64(f.num_sparse);
    if (num_buckets > 0xFFFFFFFFull || n > 0xFFFFFFFFull) {
        err = "pthash build: more than 2^32-1 keys/buckets is not supported by this builder";
        return false;
    }
    return true;
}

// First-fit pilot search, one bucket after the other in the order given: the first pilot whose table positions are free and distinct.
// `taken` (bit p of word p/64, words_for(table_size) + 1 words) is carried from call to call, so that a search may come in pieces.
// The CPU search is sequential: a speculative in-order-commit pool never beat one thread (1.05 us/key on 64 threads = sequential
// speed), and a CPU port of the GPU's window/claim/commit scheme got slower with every thread added (profiles/r01_builder_stages.txt).
// The parallel search is the GPU's (search.hip).
struct CpuPilotSearch {
    static constexpr uint64_t kCache = 1ull << 20;
    // XOR displacement keeps the low bits of a hash: with a tiny table whose size has a power-of-two factor a bucket can
    // be unplaceable for every pilot (PTHash itself would spin forever there); give up instead of hanging
    static constexpr uint64_t kPilotCap = 1ull << 28;
    std::vector<uint64_t> hp_cache;
    uint64_t seed = 0;
    void prepare(uint64_t seed_, int threads) {
        if (!hp_cache.empty() && seed == seed_) return;
        seed = seed_;
        hp_cache.resize(kCache);
        parallel_for(kCache >> 12, threads, [&](size_t t) {
            for (uint64_t p = t << 12; p < (t + 1) << 12; ++p) hp_cache[p] = murmur64a_u64(p, seed);
        });
    }
    bool run(const SinglePhf& f, const uint64_t* hs, const uint64_t* bstart, const uint32_t* order, uint64_t n_order, std::vector<uint64_t>& taken,
             uint64_t* pilots_by_bucket, std::string& err) const {
        const u128 M = f.M;
        const uint64_t table_size = f.table_size;
        std::vector<uint64_t> pos(16);
        for (uint64_t oi = 0; oi < n_order; ++oi) {
            const uint32_t b = order[oi];
            const uint64_t* keys = hs + bstart[b];
            const uint64_t s = bstart[b + 1] - bstart[b];
            if (s == 0) continue;
            if (pos.size() < s) pos.resize(s);
            uint64_t pilot = 0;
            for (;; ++pilot) {
                if (pilot >= kPilotCap) {
                    err = "pthash search: no pilot below 2^28 for some bucket (degenerate tiny key set: XOR displacement cannot separate keys that share "
                          "their low bits when the table size has a power-of-two factor; the reference's PTHash does not terminate on it either)";
                    return false;
                }
                const uint64_t hp = pilot < kCache ? hp_cache[pilot] : murmur64a_u64(pilot, seed);
                uint64_t j = 0;
                for (; j < s; ++j) {
                    const uint64_t p = fastmod_u64(keys[j] ^ hp, M, table_size);
                    if ((taken[p >> 6] >> (p & 63)) & 1) break;
                    pos[j] = p;
                }
                if (j < s) continue;
                bool d2 = false;
                for (uint64_t a = 1; a < s && !d2; ++a)
                    for (uint64_t q = 0; q < a; ++q)
                        if (pos[a] == pos[q]) {
                            d2 = true;
                            break;
                        }
                if (!d2) break;
            }
            for (uint64_t j = 0; j < s; ++j) taken[pos[j] >> 6] |= 1ULL << (pos[j] & 63);
            pilots_by_bucket[b] = pilot;
        }
        return true;
    }
};

// What follows the search: dual<dictionary, dictionary>::encode of the pilots, and for a minimal function fill_free_slots + ef_sequence<false>.
bool phf_finalize(SinglePhf& f, uint64_t num_buckets, const std::vector<uint64_t>& pilots, const std::vector<uint64_t>& taken_plain, std::string& err) {
    const uint64_t n = f.num_keys, table_size = f.table_size;
    auto is_taken = [&](uint64_t p) { return (taken_plain[p >> 6] >> (p & 63)) & 1; };
    {  // the search must have placed every key exactly once: everything below relies on it (free-slot fill, minimality)
        uint64_t ones = 0;
        for (uint64_t p = 0; p < table_size; ++p) ones += is_taken(p);
        if (ones != n) {
            err = "pthash search: internal error (occupied slots != keys)";
            return false;
        }
    }
    // front = first 0.3 * num_buckets pilots
    const uint64_t front_size = (uint64_t)((double)num_buckets * 0.3);
    dict_encode(f.front, pilots.data(), front_size);
    dict_encode(f.back, pilots.data() + front_size, num_buckets - front_size);
    if (n < table_size) {
        std::vector<uint64_t> fs;
        fs.reserve(table_size - n);
        uint64_t next_used_slot = n, last_free_slot = 0, last_valid_free_slot = 0;
        for (;;) {
            while (last_free_slot < n && is_taken(last_free_slot)) ++last_free_slot;
            if (last_free_slot == n) break;
            while (next_used_slot < table_size && !is_taken(next_used_slot)) {
                fs.push_back(last_free_slot);
                ++next_used_slot;
            }
            if (next_used_slot >= table_size) {  // cannot happen when exactly n slots are occupied (checked above)
                err = "pthash: inconsistent slot bitmap";
                return false;
            }
            fs.push_back(last_free_slot);
            last_valid_free_slot = last_free_slot;
            ++next_used_slot;
            ++last_free_slot;
        }
        while (next_used_slot != table_size) {
            fs.push_back(last_valid_free_slot);
            ++next_used_slot;
        }
        ef_encode(f.free_slots, fs.size(), fs.back(), false, [&](uint64_t i) { return fs[i]; });
    }
    return true;
}

// hashes[i] = hash.second() of key i (table position, duplicate check); firsts = hash.first() (bucket) for hash128 hashers,
// nullptr for hash64 hashers, whose first() == second()
bool phf_build(SinglePhf& f, const std::vector<uint64_t>& hashes, uint64_t seed, double c, double alpha, int threads, int device, std::string& err,
               const std::vector<uint64_t>* firsts = nullptr) {
    const bool verbose = getenv("LPHQ_BUILD_VERBOSE") != nullptr;
    auto tprev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        auto now = std::chrono::steady_clock::now();
        if (verbose) fprintf(stderr, "[lphq_build]   pthash: %-34s %8.3f s\n", what, std::chrono::duration<double>(now - tprev).count());
        tprev = now;
    };
    const uint64_t n = hashes.size();
    uint64_t num_buckets = 0;
    if (!phf_setup(f, n, seed, c, alpha, num_buckets, err)) return false;
    if (n == 0) return true;
    const uint64_t table_size = f.table_size;

    // group hashes by bucket (counting sort), hashes sorted inside each bucket
    std::vector<uint32_t> bucket_id(n);
    std::vector<std::atomic<uint32_t>> counts(num_buckets + 1);
    for (auto& x : counts) x.store(0, std::memory_order_relaxed);
    const size_t chunk = 1 << 16;
    const size_t nchunks = (n + chunk - 1) / chunk;
    parallel_for(nchunks, threads, [&](size_t t) {
        for (size_t i = t * chunk, e = std::min<size_t>(n, i + chunk); i < e; ++i) {
            uint32_t b = (uint32_t)bucket_of(f, firsts ? (*firsts)[i] : hashes[i]);
            bucket_id[i] = b;
            counts[b + 1].fetch_add(1, std::memory_order_relaxed);
        }
    });
    std::vector<uint64_t> bstart(num_buckets + 1, 0);
    uint64_t max_size = 0;
    for (uint64_t b = 0; b < num_buckets; ++b) {
        uint64_t sz = counts[b + 1].load(std::memory_order_relaxed);
        max_size = std::max(max_size, sz);
        bstart[b + 1] = bstart[b] + sz;
    }
    std::vector<uint64_t> hs(n);
    {
        std::vector<std::atomic<uint32_t>> cursor(num_buckets);
        for (auto& x : cursor) x.store(0, std::memory_order_relaxed);
        parallel_for(nchunks, threads, [&](size_t t) {
            for (size_t i = t * chunk, e = std::min<size_t>(n, i + chunk); i < e; ++i) {
                uint32_t b = bucket_id[i];
                hs[bstart[b] + cursor[b].fetch_add(1, std::memory_order_relaxed)] = hashes[i];
            }
        });
    }
    std::vector<uint32_t>().swap(bucket_id);
    std::atomic<bool> dup{false};
    const size_t bchunk = 1 << 12;
    parallel_for((num_buckets + bchunk - 1) / bchunk, threads, [&](size_t t) {
        for (uint64_t b = t * bchunk, e = std::min<uint64_t>(num_buckets, b + bchunk); b < e; ++b) {
            std::sort(hs.begin() + bstart[b], hs.begin() + bstart[b + 1]);
            for (uint64_t i = bstart[b] + 1; i < bstart[b + 1]; ++i)
                if (hs[i] == hs[i - 1]) dup.store(true);
        }
    });
    if (dup.load()) {
        err = "pthash seed failure: duplicate 64-bit hash inside a bucket (duplicate key in the input?)";
        return false;
    }
    // buckets by non-increasing size, then increasing id
    std::vector<uint32_t> order(num_buckets);
    {
        std::vector<uint64_t> cnt(max_size + 2, 0);
        for (uint64_t b = 0; b < num_buckets; ++b) cnt[max_size - (bstart[b + 1] - bstart[b]) + 1]++;
        for (uint64_t s = 0; s <= max_size; ++s) cnt[s + 1] += cnt[s];
        for (uint64_t b = 0; b < num_buckets; ++b) order[cnt[max_size - (bstart[b + 1] - bstart[b])]++] = (uint32_t)b;
    }
    uint64_t n_nonempty = num_buckets;
    while (n_nonempty && bstart[order[n_nonempty - 1] + 1] == bstart[order[n_nonempty - 1]]) --n_nonempty;

    lap("group hashes by bucket, order");
    // search
    std::vector<uint64_t> pilots(num_buckets, 0);
    std::vector<uint64_t> taken_plain;  // bit p of word p/64: table position p is occupied
    bool searched = false;
    if (device >= 0 && n >= 4096) {  // pilot search on the GPU (search.hip): same first-fit result, ~16x faster
        searched = gpu_pilot_search(device, hs, bstart, order, n_nonempty, table_size, seed, pilots, taken_plain, err);
        if (!searched) {  // construction is not the query path: finish on the CPU, loudly
            fprintf(stderr, "[lphq_build] WARNING: GPU pilot search failed (%s); falling back to the sequential CPU search\n", err.c_str());
            std::fill(pilots.begin(), pilots.end(), 0);
            taken_plain.clear();
            err.clear();
        }
    }
    const bool used_gpu = searched;
    if (!searched) {
        taken_plain.assign(words_for(table_size) + 1, 0);
        CpuPilotSearch cpu;
        cpu.prepare(seed, threads);
        if (!cpu.run(f, hs.data(), bstart.data(), order.data(), n_nonempty, taken_plain, pilots.data(), err)) return false;
    }
    lap(used_gpu ? "pilot search (GPU)" : "pilot search (CPU)");
    if (!phf_finalize(f, num_buckets, pilots, taken_plain, err)) return false;
    lap("dictionary encode, free slots");
    return true;
}

uint64_t phf_eval(const SinglePhf& f, uint64_t hash, const std::vector<uint64_t>* free_table) {
    uint64_t b = bucket_of(f, hash);
    uint64_t pilot = b < f.front.ranks.size ? f.front.dict.access(f.front.ranks.access(b))
                                            : f.back.dict.access(f.back.ranks.access(b - f.front.ranks.size));
    uint64_t p = fastmod_u64(hash ^ murmur64a_u64(pilot, f.seed), f.M, f.table_size);
    if (p < f.num_keys) return p;
    return (*free_table)[p - f.num_keys];
}

// ---------------------------------------------------------------------------------------------
// super-k-mers
// ---------------------------------------------------------------------------------------------
typedef SuperKmerRun Run;  // extract.hpp: {mm, first, q, size, pad}, batch coordinates are global
typedef SuperKmerRuns RunVec;  // (a vector whose resize() does not touch the new elements)

// Windows [wa, wb) of one record of rec_len bases (record base pointer s, window w covers bases [w, w+k)).
// Windows holding a non-ACGT base are skipped and cut runs (the reference resets there, minimizer.hpp:150-164).
// One window is valid and still in no run, because the reference leaves it out: the only window of a stretch of EXACTLY k valid bases that a non-ACGT base
// closes inside the record.  from_string looks for the minimum of a stretch's first window when base k + 1 arrives (minimizer.hpp:61-66); a break after base k
// comes first, finds min_pos still unset and pushes nothing (minimizer.hpp:139-152; get_colliding_kmers, :287-301, alike) -- while the k-mer was counted
// (:60).  Such a window adds to `lost` (it is part of nkmers) and to no super-k-mer, so the file is the reference's byte for byte on such input too.  (A stretch
// of k bases that ends WITH the record is kept: minimizer.hpp:154-163.)
inline bool lost_window(const char* s, uint64_t rec_len, uint64_t w, uint32_t k) {
    return (w == 0 || nt4((uint8_t)s[w - 1]) > 3) && w + k < rec_len && nt4((uint8_t)s[w + k]) > 3;
}
void extract_runs(const char* s, uint64_t rec_len, uint64_t rec_off, uint64_t wa, uint64_t wb, uint32_t k, uint32_t m, uint64_t seed, RunVec& out, uint64_t& lost) {
    if (wa >= wb) return;
    const uint32_t W = k - m + 1;
    const uint64_t mask = m >= 32 ? ~0ULL : ((1ULL << (2 * m)) - 1);
    std::vector<uint64_t> hbuf(W), vbuf(W);  // ring over m-mer positions
    uint64_t mmer = 0;
    uint64_t valid_run = 0;  // consecutive valid bases ending at the current base
    // m-mer positions needed: wa .. wb-1+W-1 ; bases wa .. wb+k-2
    uint64_t cur_q = ~0ULL, cur_hash = 0;  // current minimum (leftmost) among the window's m-mers
    Run run{0, 0, ~0ULL, 0, 0};
    auto flush = [&] {
        if (run.size) out.push_back(run);
        run.size = 0;
        run.q = ~0ULL;
    };
    for (uint64_t i = wa; i < wb + k - 1; ++i) {
        const uint8_t c = nt4((uint8_t)s[i]);
        if (c > 3) {
            valid_run = 0;
            mmer = 0;
            flush();
            cur_q = ~0ULL;
            continue;
        }
        mmer = ((mmer << 2) | c) & mask;
        ++valid_run;
        if (valid_run < m) continue;
        const uint64_t pq = i + 1 - m;  // position of this m-mer
        const uint64_t h = murmur64a_u64(mmer, seed);
        hbuf[pq % W] = h;
        vbuf[pq % W] = mmer;
        if (valid_run < k) continue;
        const uint64_t w = i + 1 - k;  // window ending at base i
        if (w < wa) continue;
        if (valid_run == k && lost_window(s, rec_len, w, k)) {  // (valid_run == k: the first whole window since the last break, or since the chunk began)
            ++lost;
            continue;  // the base after it is the break: the state is reset there
        }
        // leftmost minimum over m-mer positions [w, w+W-1] (pq == w+W-1)
        if (cur_q == ~0ULL || cur_q < w || valid_run == k) {
            // (re)scan the whole window
            cur_q = w;
            cur_hash = hbuf[w % W];
            for (uint64_t t = w + 1; t <= pq; ++t)
                if (hbuf[t % W] < cur_hash) {
                    cur_hash = hbuf[t % W];
                    cur_q = t;
                }
        } else if (h < cur_hash) {
            cur_hash = h;
            cur_q = pq;
        }
        if (run.size && run.q == rec_off + cur_q) {
            ++run.size;
        } else {
            flush();
            run.mm = vbuf[cur_q % W];
            run.first = rec_off + w;
            run.q = rec_off + cur_q;
            run.size = 1;
        }
    }
    flush();
}

struct Triplet {  // what is left of mm_triplet_t (include/constants.hpp:37-43) once the minimizer itself has become a position: (p1, size), 0 / 0 for a
                  // repeated minimizer; both at most k - m + 1 <= 63, so 0xFF / 0xFF marks a slot nothing was placed in yet
    uint8_t p1, size;
};

// ---------------------------------------------------------------------------------------------
// the input of a build: batches of whole records, read twice (src/partitioned_mphf.cpp:58-81 and :110-130)
// ---------------------------------------------------------------------------------------------
struct RecordSource {
    virtual ~RecordSource() {}
    // next batch of pass `pass` (0, then 1; the first call of pass 1 restarts the input): 1 = a batch, 0 = end of input, < 0 = error
    virtual int next(int pass, const char* const*& recs, const size_t*& lens, size_t& n) = 0;
    virtual bool one_batch() const { return false; }  // the first batch is the whole input
};
struct MemorySource : RecordSource {  // lphq_build: the whole input is one batch
    bool one_batch() const override { return true; }
    const char* const* recs_;
    const size_t* lens_;
    size_t n_;
    int served_ = -1;
    MemorySource(const char* const* r, const size_t* l, size_t n) : recs_(r), lens_(l), n_(n) {}
    int next(int pass, const char* const*& recs, const size_t*& lens, size_t& n) override {
        if (served_ == pass) return 0;
        served_ = pass;
        recs = recs_;
        lens = lens_;
        n = n_;
        return 1;
    }
};
struct CallbackSource : RecordSource {  // lphq_build_stream
    lphq_record_source fn_;
    void* ctx_;
    CallbackSource(lphq_record_source fn, void* ctx) : fn_(fn), ctx_(ctx) {}
    int next(int pass, const char* const*& recs, const size_t*& lens, size_t& n) override {
        const char* const* r = nullptr;
        const size_t* l = nullptr;
        size_t cnt = 0;
        const int rc = fn_(ctx_, pass, &r, &l, &cnt);
        if (rc <= 0) return rc;
        if (cnt && (!r || !l)) return -1;
        recs = r;
        lens = l;
        n = cnt;
        return 1;
    }
};

struct RunKey {
    uint64_t operator()(const Run& r) const { return r.mm; }
};
typedef Spill<Run, RunVec, RunKey> RunSpill;

// the per-minimizer vectors of a build with a memory limit (external_memory_vector<mm_triplet_t> and <uint64_t>, src/partitioned_mphf.cpp:88-130)
struct UniqRec {  // one distinct minimizer, in minimizer order: p1 = size = 0 for a repeated one
    uint64_t mm;
    uint8_t p1, size;
};
struct OrderRec {  // the same, keyed by its value under the minimizer MPHF
    uint64_t order;
    uint8_t p1, size;
};
struct OrderKey {
    uint64_t operator()(const OrderRec& r) const { return r.order; }
};
typedef std::vector<OrderRec, DefaultInitAllocator<OrderRec>> OrderVec;
struct CollRec {  // (first window, size) of a super-k-mer of a repeated minimizer
    uint64_t first;
    uint32_t size, pad;
};
struct CollKey {
    uint64_t operator()(const CollRec& r) const { return r.first; }
};
typedef std::vector<CollRec, DefaultInitAllocator<CollRec>> CollVec;

// ---------------------------------------------------------------------------------------------
// PTHash construction with a memory limit (single_phf::build_in_external_memory, what src/partitioned_mphf.cpp:143-160 calls):
// the (bucket, hash) pairs are sorted in pieces on disk; their merge writes every bucket into the file of its size; the search
// reads those files from the largest size down -- the order of the in-memory search (non-increasing size, then increasing id) --
// a piece at a time, with the table of taken positions carried from piece to piece.  In memory: the pilots (8 B per bucket), the
// table's bitmap, one piece.  Same function, bit for bit, as phf_build.
// ---------------------------------------------------------------------------------------------
struct BucketRec {
    uint64_t hash;
    uint32_t bucket, pad;
};
struct BucketKey {
    uint64_t operator()(const BucketRec& r) const { return r.bucket; }
};
typedef std::vector<BucketRec, DefaultInitAllocator<BucketRec>> BucketVec;

class PhfExternalBuilder {
  public:
    uint64_t big_bucket_ = 256;  // buckets of at least this many keys (unheard of for hashed keys) share one file, sorted in memory
                          // (LPHQ_EXT_BIG_BUCKET lowers the threshold: a test hook, the result does not depend on it)
    bool begin(SinglePhf& f, uint64_t n, uint64_t seed, double c, double alpha, const std::string& dir, uint64_t budget_bytes, int threads, int device,
               std::string& err) {
        f_ = &f;
        threads_ = threads;
        device_ = device;
        dir_ = dir;
        if (!phf_setup(f, n, seed, c, alpha, num_buckets_, err)) return false;
        spill_.dir = dir;
        spill_.kind = "buckets";
        if (const char* e = getenv("LPHQ_EXT_BIG_BUCKET")) big_bucket_ = std::max<uint64_t>(2, std::min<uint64_t>(256, strtoull(e, nullptr, 10)));
        cap_ = std::max<size_t>(4096, (size_t)(budget_bytes / 4 / sizeof(BucketRec)));
        merge_buf_ = std::max<size_t>(256, std::min<size_t>(1 << 16, cap_ / (2 * Spill<BucketRec, BucketVec, BucketKey>::kMaxPieces)));
        piece_keys_ = std::max<size_t>(4096, (size_t)(budget_bytes / 4 / 8));
        buf_.reserve(std::min<uint64_t>(cap_, n));
        bucket_bits_ = msb64(num_buckets_ | 1) + 1;
        pushed_ = 0;
        return true;
    }
    bool push(uint64_t hash, std::string& err) {
        if (buf_.size() >= cap_ && !spill_.flush(buf_, threads_, bucket_bits_, merge_buf_, err)) return false;
        buf_.push_back(BucketRec{hash, (uint32_t)bucket_of(*f_, hash), 0});
        ++pushed_;
        return true;
    }
    bool finish(std::string& err) {
        SinglePhf& f = *f_;
        const uint64_t n = f.num_keys;
        if (pushed_ != n) {
            err = "pthash build: internal error (keys announced != keys given)";
            return false;
        }
        if (n == 0) return true;
        const bool verbose = getenv("LPHQ_BUILD_VERBOSE") != nullptr;
        auto tprev = std::chrono::steady_clock::now();
        auto lap = [&](const std::string& what) {
            auto now = std::chrono::steady_clock::now();
            if (verbose) fprintf(stderr, "[lphq_build]   pthash: %-50s %8.3f s\n", what.c_str(), std::chrono::duration<double>(now - tprev).count());
            tprev = now;
        };
        const size_t n_pair_pieces = spill_.files.size() + (spill_.files.empty() ? 0 : 1);
        // 1. buckets, one after the other by id, into the file of their size
        std::vector<std::unique_ptr<RecordFile<uint64_t>>> by_size(big_bucket_ + 1);  // [s] for s < big_bucket_: (id, s hashes)...; [big_bucket_]: (size, id, hashes)...
        uint64_t max_size = 0;
        {
            std::vector<uint64_t> keys;
            uint32_t cur = 0;
            bool have = false, ok = true;
            auto close_bucket = [&]() {
                if (!have) return;
                std::sort(keys.begin(), keys.end());
                for (size_t i = 1; i < keys.size(); ++i)
                    if (keys[i] == keys[i - 1]) {
                        err = "pthash seed failure: duplicate 64-bit hash inside a bucket (duplicate key in the input?)";
                        ok = false;
                        return;
                    }
                const uint64_t s = keys.size(), slot = std::min<uint64_t>(s, big_bucket_);
                max_size = std::max(max_size, s);
                if (!by_size[slot]) {
                    by_size[slot].reset(new RecordFile<uint64_t>());
                    if (!by_size[slot]->create(tmp_name(dir_, "size" + std::to_string(slot)), 1 << 12)) {
                        err = "cannot create the temporary file " + by_size[slot]->path;
                        ok = false;
                        return;
                    }
                }
                RecordFile<uint64_t>& out = *by_size[slot];
                bool w = true;
                if (slot == big_bucket_) w = out.put(s);
                w = w && out.put(cur);
                for (uint64_t h : keys) w = w && out.put(h);
                if (!w) {
                    err = "short write on the temporary file " + out.path;
                    ok = false;
                }
                keys.clear();
                have = false;
            };
            auto feed = [&](const BucketRec& r) {
                if (have && r.bucket != cur) close_bucket();
                if (!ok) return false;
                cur = r.bucket;
                have = true;
                keys.push_back(r.hash);
                return true;
            };
            if (spill_.files.empty()) {  // everything still in memory: one sort, no file of pairs
                parallel_sort_by_key(buf_, threads_, bucket_bits_, [](const BucketRec& r) { return (uint64_t)r.bucket; });
                for (const BucketRec& r : buf_)
                    if (!feed(r)) return false;
            } else {
                if (!spill_.flush(buf_, threads_, bucket_bits_, merge_buf_, err)) return false;
                if (!Spill<BucketRec, BucketVec, BucketKey>::merge(spill_.files, merge_buf_, err, feed)) return false;
            }
            close_bucket();
            if (!ok) return false;
            BucketVec().swap(buf_);
            spill_.drop();
        }
        lap(std::to_string(n_pair_pieces) + " sorted pieces of (bucket, hash) -> files by size");
        // 2. the search, a piece of the order at a time
        uint64_t n_search_pieces = 0, n_on_gpu = 0;
        std::vector<uint64_t> pilots(num_buckets_, 0);
        std::vector<uint64_t> taken(words_for(f.table_size) + 1, 0);
        std::vector<uint64_t> hs, bstart(1, 0);
        std::vector<uint32_t> ids;
        CpuPilotSearch cpu;
        uint64_t placed = 0;
        auto search_piece = [&]() -> bool {
            if (ids.empty()) return true;
            const uint64_t nb = ids.size();
            std::vector<uint32_t> order(nb);
            for (uint64_t i = 0; i < nb; ++i) order[i] = (uint32_t)i;
            std::vector<uint64_t> local(nb, 0);
            bool searched = false;
            if (device_ >= 0 && hs.size() >= 4096) {
                std::vector<uint64_t> t = taken;  // (left as it was if the device fails half-way)
                std::string gerr;
                searched = gpu_pilot_search(device_, hs, bstart, order, nb, f.table_size, f.seed, local, t, gerr);
                if (searched) {
                    taken.swap(t);
                    ++n_on_gpu;
                } else {
                    fprintf(stderr, "[lphq_build] WARNING: GPU pilot search failed (%s); falling back to the sequential CPU search\n", gerr.c_str());
                    std::fill(local.begin(), local.end(), 0);
                }
            }
            if (!searched) {
                cpu.prepare(f.seed, threads_);
                if (!cpu.run(f, hs.data(), bstart.data(), order.data(), nb, taken, local.data(), err)) return false;
            }
            for (uint64_t i = 0; i < nb; ++i) pilots[ids[i]] = local[i];
            ++n_search_pieces;
            placed += hs.size();
            hs.clear();
            bstart.assign(1, 0);
            ids.clear();
            return true;
        };
        auto add_bucket = [&](uint32_t id, const uint64_t* keys, uint64_t s) -> bool {
            if (!hs.empty() && hs.size() + s > piece_keys_ && !search_piece()) return false;
            ids.push_back(id);
            hs.insert(hs.end(), keys, keys + s);
            bstart.push_back(hs.size());
            return true;
        };
        if (by_size[big_bucket_]) {  // (size, id, hashes) records: all in memory, by non-increasing size then id
            RecordFile<uint64_t>& in = *by_size[big_bucket_];
            if (!in.rewind_for_reading()) return io_error(in, err);
            struct Big {
                uint64_t size, id;
                std::vector<uint64_t> keys;
            };
            std::vector<Big> big;
            uint64_t s = 0, id = 0, h = 0;
            while (in.next(s)) {
                if (!in.next(id)) return io_error(in, err);
                big.push_back(Big{s, id, {}});
                big.back().keys.reserve(s);
                for (uint64_t j = 0; j < s; ++j) {
                    if (!in.next(h)) return io_error(in, err);
                    big.back().keys.push_back(h);
                }
            }
            if (in.failed) return io_error(in, err);
            std::sort(big.begin(), big.end(), [](const Big& a, const Big& b) { return a.size != b.size ? a.size > b.size : a.id < b.id; });
            for (const Big& b : big)
                if (!add_bucket((uint32_t)b.id, b.keys.data(), b.size)) return false;
            by_size[big_bucket_].reset();
        }
        std::vector<uint64_t> keys;
        for (uint64_t s = std::min<uint64_t>(max_size, big_bucket_ - 1); s >= 1; --s) {
            if (!by_size[s]) continue;
            RecordFile<uint64_t>& in = *by_size[s];
            if (!in.rewind_for_reading()) return io_error(in, err);
            keys.resize(s);
            uint64_t id = 0;
            while (in.next(id)) {
                for (uint64_t j = 0; j < s; ++j)
                    if (!in.next(keys[j])) return io_error(in, err);
                if (!add_bucket((uint32_t)id, keys.data(), s)) return false;
            }
            if (in.failed) return io_error(in, err);
            by_size[s].reset();
        }
        if (!search_piece()) return false;
        if (placed != n) {
            err = "pthash build: internal error (keys searched != keys given)";
            return false;
        }
        lap("pilot search in " + std::to_string(n_search_pieces) + " pieces (" + std::to_string(n_on_gpu) + " on the GPU)");
        const bool ok = phf_finalize(f, num_buckets_, pilots, taken, err);
        lap("dictionary encode, free slots");
        return ok;
    }

  private:
    static bool io_error(const RecordFile<uint64_t>& f, std::string& err) {
        err = "read error on the temporary file " + f.path;
        return false;
    }
    SinglePhf* f_ = nullptr;
    uint64_t num_buckets_ = 0, pushed_ = 0;
    int threads_ = 1, device_ = -1;
    std::string dir_;
    Spill<BucketRec, BucketVec, BucketKey> spill_;
    BucketVec buf_;
    size_t cap_ = 0, merge_buf_ = 0, piece_keys_ = 0;
    uint32_t bucket_bits_ = 32;
};

}  // namespace
}  // namespace lphq

using namespace lphq;

namespace lphq { void set_last_error(const std::string& s); }  // capi.cpp

static int build_impl(RecordSource& src, int k_, int m_, uint64_t mm_seed, double c, int kind, int kmer_bits, int threads, int device,
                      const char* tmp_dir, uint64_t max_memory_bytes, const char* out_path, lphq_build_stats* stats) {
    auto fail = [&](int code, const std::string& msg) {
        set_last_error(msg);
        return code;
    };
    if (!out_path) return fail(LPHQ_ERR_ARG, "null argument");
    if (kind != LPHQ_PARTITIONED && kind != LPHQ_UNPARTITIONED) return fail(LPHQ_ERR_ARG, "kind must be LPHQ_PARTITIONED or LPHQ_UNPARTITIONED");
    if (kmer_bits != 64 && kmer_bits != 128) return fail(LPHQ_ERR_ARG, "kmer_bits must be 64 or 128");
    // parser_build.cpp:64-68 with constants::max_k of the chosen kmer_t (constants.hpp:16)
    if (k_ < 1 || k_ > kmer_bits / 2 - 1) return fail(LPHQ_ERR_ARG, "k cannot be larger than " + std::to_string(kmer_bits / 2 - 1));
    if (m_ < 1 || m_ > k_ || m_ > 32) return fail(LPHQ_ERR_ARG, "m cannot be larger than k (and must be <= 32)");
    if (c < 3.0 || c > 10.0) return fail(LPHQ_ERR_ARG, "3.0 <= c <= 10.0");  // parser_build.cpp:78-81
    if (threads <= 0) threads = usable_cpus();
    const uint32_t k = (uint32_t)k_, m = (uint32_t)m_, W = k - m + 1;
    const bool verbose = getenv("LPHQ_BUILD_VERBOSE") != nullptr;
    auto tprev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        auto now = std::chrono::steady_clock::now();
        if (verbose) fprintf(stderr, "[lphq_build] %-44s %8.3f s   rss %6.0f MB anonymous, %6.0f MB with file pages\n", what, std::chrono::duration<double>(now - tprev).count(), resident_mb(), resident_mb(false));
        tprev = now;
    };
    const double alpha = 0.94;       // partitioned_mphf.cpp:48
    const uint64_t pthash_seed = 1;  // constants::default_pthash_seed

    // ---- Part 1: super-k-mers, batch by batch, chunked over windows -------------------------------------
    // Coordinates are global: base g of the input = base g - (bases of the earlier records) of its record.
    // With a memory limit the records are sorted and spilled in pieces (external_memory_vector<mm_record_t>, partitioned_mphf.cpp:62-65).
    const size_t run_cap = max_memory_bytes ? std::max<size_t>(4096, (size_t)(max_memory_bytes / 4 / sizeof(Run))) : ~(size_t)0;
    const size_t merge_buf = max_memory_bytes ? std::max<size_t>(256, std::min<size_t>(1 << 16, run_cap / (2 * RunSpill::kMaxPieces))) : 0;
    RunSpill spill;
    spill.dir = (tmp_dir && *tmp_dir) ? tmp_dir : ".";
    if (max_memory_bytes) (void)::mkdir(spill.dir.c_str(), 0777);  // essentials::create_directory, parser_build.cpp:76
    std::string err;
    RunVec runs;
    uint64_t nkmers = 0, n_super_kmers = 0, n_records = 0, base = 0;
    int extract_device = getenv("LPHQ_BUILD_HOST_EXTRACT") ? -1 : device;  // (the variable exists for A/B timing; the result is the same)
    ClassifiedMinimizers on_device;  // the whole input in one batch and a device: extraction, sort AND classification there (extract.hip)
    bool classified_on_device = false;
    bool runs_sorted = false;  // `runs` ascending by minimizer already (one batch, extracted and sorted on the GPU)
    uint64_t gpu_batches = 0;
    {
        const char* const* records = nullptr;
        const size_t* lengths = nullptr;
        size_t nb = 0;
        int rc;
        std::vector<uint64_t> rec_off;
        struct Chunk { size_t rec; uint64_t wa, wb; };
        std::vector<Chunk> chunks;
        std::vector<RunVec> chunk_runs;
        std::vector<uint64_t> chunk_lost;
        while ((rc = src.next(0, records, lengths, nb)) > 0) {
            rec_off.assign(nb + 1, base);
            for (size_t r = 0; r < nb; ++r) rec_off[r + 1] = rec_off[r] + lengths[r];
            chunks.clear();
            const uint64_t CH = 1u << 20;
            for (size_t r = 0; r < nb; ++r) {
                if (lengths[r] < k) continue;
                const uint64_t nw = lengths[r] - k + 1;
                for (uint64_t a = 0; a < nw; a += CH) chunks.push_back({r, a, std::min(nw, a + CH)});
            }
            if (extract_device >= 0 && src.one_batch() && !max_memory_bytes && !getenv("LPHQ_BUILD_HOST_CLASSIFY")) {
                std::string gerr;
                if (gpu_extract_classified(extract_device, records, lengths, nb, k, m, mm_seed, threads, verbose, on_device, gerr)) {
                    classified_on_device = true;
                    nkmers = on_device.num_kmers;
                    n_super_kmers = on_device.num_super_kmers;
                    ++gpu_batches;
                    base = rec_off[nb];
                    n_records += nb;
                    continue;
                }
                fprintf(stderr, "[lphq_build] GPU super-k-mer extraction unavailable (%s): the host extracts\n", gerr.c_str());
                extract_device = -1;
            }
            if (extract_device >= 0) {  // super-k-mers of the batch from the GPU, already sorted by minimizer
                std::string gerr;
                const bool was_empty = runs.empty() && spill.files.empty();
                uint64_t lost = 0;
                if (gpu_extract_sorted_runs(extract_device, records, lengths, nb, base, k, m, mm_seed, threads, verbose, runs, lost, gerr)) {
                    nkmers += lost;  // (counted like the reference counts them, in no super-k-mer: extract_runs)
                    runs_sorted = was_empty;  // still sorted as a whole if this is all there is
                    if (runs.size() >= run_cap) {  // over the memory limit: one more sorted piece
                        for (const Run& x : runs) nkmers += x.size;
                        n_super_kmers += runs.size();
                        if (!spill.flush(runs, threads, 2 * m, merge_buf, err)) return fail(LPHQ_ERR_IO, err);
                        runs_sorted = false;
                    }
                    ++gpu_batches;
                    base = rec_off[nb];
                    n_records += nb;
                    continue;
                }
                fprintf(stderr, "[lphq_build] GPU super-k-mer extraction unavailable (%s): the host extracts\n", gerr.c_str());
                extract_device = -1;
            }
            runs_sorted = false;
            chunk_runs.assign(chunks.size(), RunVec());
            chunk_lost.assign(chunks.size(), 0);
            parallel_for(chunks.size(), threads, [&](size_t ci) {
                const Chunk& ch = chunks[ci];
                chunk_runs[ci].reserve((ch.wb - ch.wa) * 2 / (W + 1) + 16);
                extract_runs(records[ch.rec], lengths[ch.rec], rec_off[ch.rec], ch.wa, ch.wb, k, m, mm_seed, chunk_runs[ci], chunk_lost[ci]);
            });
            for (uint64_t x : chunk_lost) nkmers += x;
            size_t total = runs.size();
            for (auto& v : chunk_runs) total += v.size();
            if (runs.capacity() < total) runs.reserve(std::min(run_cap, std::max(total, runs.capacity() * 2)));
            for (size_t ci = 0; ci < chunks.size(); ++ci) {
                bool first_of_chunk = true;
                for (const Run& r : chunk_runs[ci]) {
                    // a super-k-mer cut by a chunk boundary: same minimizer position, contiguous windows (a chunk never begins a
                    // batch's first record elsewhere than at its first window, so the piece before is still in `runs`)
                    if (first_of_chunk && !runs.empty() && runs.back().q == r.q && runs.back().first + runs.back().size == r.first && ci > 0 &&
                        chunks[ci].rec == chunks[ci - 1].rec) {
                        runs.back().size += r.size;
                    } else {
                        if (runs.size() >= run_cap) {  // (never between the two pieces of a cut super-k-mer: that case is the branch above)
                            const Run keep = runs.back();  // the last one may still grow at the next chunk boundary
                            runs.pop_back();
                            for (const Run& x : runs) nkmers += x.size;
                            n_super_kmers += runs.size();
                            if (!spill.flush(runs, threads, 2 * m, merge_buf, err)) return fail(LPHQ_ERR_IO, err);
                            runs.push_back(keep);
                        }
                        runs.push_back(r);
                    }
                    first_of_chunk = false;
                }
                RunVec().swap(chunk_runs[ci]);
            }
            base = rec_off[nb];
            n_records += nb;
        }
        if (rc < 0) return fail(LPHQ_ERR_IO, "the record source failed during the first pass");
    }
    for (const Run& x : runs) nkmers += x.size;
    n_super_kmers += runs.size();
    if (verbose && gpu_batches) fprintf(stderr, "[lphq_build] super-k-mers of %llu batch(es) extracted and sorted on GPU %d\n", (unsigned long long)gpu_batches, device);
    lap("part 1: super-k-mer runs");

    // ---- Part 2: classify (src/minimizer.cpp:5-50) + PTHash over the distinct minimizers -------------
    typedef std::vector<Triplet, DefaultInitAllocator<Triplet>> TripletVec;  // (resize() leaves the new elements to the parallel copy that follows)
    typedef std::vector<uint64_t, DefaultInitAllocator<uint64_t>> WordVec;
    TripletVec uniq;
    WordVec uniq_mm;
    std::vector<std::pair<uint64_t, uint32_t>> colliding;  // (first window, size) of every super-k-mer of a repeated minimizer
    // ... and their homes when there is a memory limit
    const bool external = max_memory_bytes != 0;
    const size_t ext_cap = std::max<size_t>(4096, (size_t)(max_memory_bytes / 4 / 16));  // records of 16 bytes per sorted piece
    const size_t ext_merge_buf = std::max<size_t>(256, std::min<size_t>(1 << 16, ext_cap / (2 * RunSpill::kMaxPieces)));
    const uint32_t coord_bits = msb64(base | 1) + 1;  // input coordinates
    RecordFile<UniqRec> uniq_file;
    Spill<CollRec, CollVec, CollKey> coll_spill;
    coll_spill.dir = spill.dir;
    coll_spill.kind = "repeated";
    CollVec coll_buf;
    uint64_t n_colliding = 0, n_colliding_kmers = 0;
    {
        // records arrive ascending by minimizer; a minimizer seen once keeps (offset, size), one seen more often sends all its super-k-mers
        // to the fallback
        struct Classifier {
            TripletVec uniq;
            WordVec uniq_mm;
            std::vector<std::pair<uint64_t, uint32_t>> colliding;
            Run prev{0, 0, 0, 0, 0};
            bool have = false, repeated = false;
            void close_group() {
                if (!have) return;
                uniq_mm.push_back(prev.mm);
                if (repeated) uniq.push_back({0, 0});
                else uniq.push_back({(uint8_t)(prev.q - prev.first), (uint8_t)prev.size});
                have = false;
            }
            void feed(const Run& r) {
                if (have && r.mm == prev.mm) {
                    if (!repeated) colliding.emplace_back(prev.first, prev.size);
                    repeated = true;
                    colliding.emplace_back(r.first, r.size);
                } else {
                    close_group();
                    prev = r;
                    have = true;
                    repeated = false;
                }
            }
        };
        if (classified_on_device) {  // only the unpacking is left
            const size_t nu = on_device.minimizer.size(), nc = on_device.colliding_first.size();
            uniq_mm.swap(on_device.minimizer);
            uniq.resize(nu);
            colliding.resize(nc);
            parallel_for((nu + 65535) / 65536, threads, [&](size_t t) {
                for (size_t i = t * 65536, e = std::min(nu, i + 65536); i < e; ++i) {
                    const uint16_t ps = on_device.offset_size[i];
                    uniq[i] = Triplet{(uint8_t)(ps & 0xFF), (uint8_t)(ps >> 8)};
                }
            });
            parallel_for((nc + 65535) / 65536, threads, [&](size_t t) {
                for (size_t i = t * 65536, e = std::min(nc, i + 65536); i < e; ++i) colliding[i] = {on_device.colliding_first[i], on_device.colliding_size[i]};
            });
            on_device = ClassifiedMinimizers();
        } else if (!external) {
            if (!runs_sorted) parallel_sort_by_key(runs, threads, 2 * m, [](const Run& r) { return r.mm; });
            lap("part 2: sort runs by minimizer");
            // pieces that begin at the first record of a minimizer, one classifier each, joined in order
            const size_t n = runs.size();
            const size_t pieces = std::max<size_t>(1, std::min<size_t>((size_t)threads, n / (1u << 16)));
            std::vector<size_t> cut(pieces + 1, n);
            cut[0] = 0;
            for (size_t t = 1; t < pieces; ++t) {
                size_t c = std::max(cut[t - 1], n * t / pieces);
                while (c < n && c > 0 && runs[c].mm == runs[c - 1].mm) ++c;
                cut[t] = c;
            }
            std::vector<Classifier> part(pieces);
            parallel_for(pieces, threads, [&](size_t t) {
                Classifier& c = part[t];
                c.uniq.reserve(cut[t + 1] - cut[t]);
                c.uniq_mm.reserve(cut[t + 1] - cut[t]);
                for (size_t i = cut[t]; i < cut[t + 1]; ++i) c.feed(runs[i]);
                c.close_group();
            });
            std::vector<size_t> uo(pieces + 1, 0), co(pieces + 1, 0);
            for (size_t t = 0; t < pieces; ++t) {
                uo[t + 1] = uo[t] + part[t].uniq.size();
                co[t + 1] = co[t] + part[t].colliding.size();
            }
            uniq.resize(uo[pieces]);
            uniq_mm.resize(uo[pieces]);
            colliding.resize(co[pieces]);
            parallel_for(pieces, threads, [&](size_t t) {
                std::copy(part[t].uniq.begin(), part[t].uniq.end(), uniq.begin() + uo[t]);
                std::copy(part[t].uniq_mm.begin(), part[t].uniq_mm.end(), uniq_mm.begin() + uo[t]);
                std::copy(part[t].colliding.begin(), part[t].colliding.end(), colliding.begin() + co[t]);
                Classifier().uniq.swap(part[t].uniq);
            });
        } else {  // with a memory limit: the distinct minimizers go to a file as the merge delivers them, the repeated ones' super-k-mers to sorted pieces
            if (!spill.flush(runs, threads, 2 * m, merge_buf, err, runs_sorted)) return fail(LPHQ_ERR_IO, err);
            RunVec().swap(runs);
            if (verbose) fprintf(stderr, "[lphq_build] %zu sorted pieces of super-k-mer records in %s\n", spill.files.size(), spill.dir.c_str());
            if (!uniq_file.create(tmp_name(spill.dir, "minimizers"), 1 << 14))
                return fail(LPHQ_ERR_IO, "cannot create the temporary file " + uniq_file.path);
            Classifier c;
            bool wrote = true;
            auto drain = [&]() {
                for (size_t i = 0; i < c.uniq.size(); ++i) wrote = uniq_file.put(UniqRec{c.uniq_mm[i], c.uniq[i].p1, c.uniq[i].size}) && wrote;
                c.uniq.clear();
                c.uniq_mm.clear();
                for (const auto& cw : c.colliding) {
                    if (coll_buf.size() >= ext_cap && !coll_spill.flush(coll_buf, threads, coord_bits, ext_merge_buf, err)) return false;
                    coll_buf.push_back(CollRec{cw.first, cw.second, 0});
                    n_colliding_kmers += cw.second;
                    ++n_colliding;
                }
                c.colliding.clear();
                return wrote;
            };
            if (!RunSpill::merge(spill.files, merge_buf, err, [&](const Run& r) {
                    c.feed(r);
                    // (both lists are bounded: one heavily repeated minimizer -- millions of super-k-mers on a low-complexity genome -- grows `colliding`
                    //  alone; draining in the middle of a group is fine, the group's state lives in prev / have / repeated)
                    return (c.uniq.size() < 4096 && c.colliding.size() < 4096) || drain();
                }))
                return fail(LPHQ_ERR_IO, err.empty() ? "short write on the temporary file " + uniq_file.path : err);
            c.close_group();
            if (!drain()) return fail(LPHQ_ERR_IO, err.empty() ? "short write on the temporary file " + uniq_file.path : err);
            spill.drop();
        }
    }
    RunVec().swap(runs);
    if (!external) {
        n_colliding = colliding.size();
        for (const auto& cw : colliding) n_colliding_kmers += cw.second;
    }
    const uint64_t D = external ? uniq_file.count : uniq.size();
    lap("part 2: classify");

    LphFile x;
    x.kind = kind;
    x.kmer_bits = kmer_bits;
    x.k = (uint8_t)k;
    x.m = (uint8_t)m;
    x.mm_seed = mm_seed;
    x.nkmers = nkmers;
    x.distinct_minimizers = D;
    if (!external) {
        std::vector<uint64_t> hashes(D);
        parallel_for((D + 65535) / 65536, threads, [&](size_t t) {
            for (size_t i = t * 65536, e = std::min<size_t>(D, i + 65536); i < e; ++i) hashes[i] = murmur64a_u64(uniq_mm[i], pthash_seed);
        });
        if (!phf_build(x.minimizer_order, hashes, pthash_seed, c, alpha, threads, device, err)) return fail(LPHQ_ERR_FORMAT, err);
    } else {  // build_minimizers_mphf -> build_in_external_memory (src/partitioned_mphf.cpp:143-150)
        PhfExternalBuilder pb;
        if (!pb.begin(x.minimizer_order, D, pthash_seed, c, alpha, spill.dir, max_memory_bytes, threads, device, err)) return fail(LPHQ_ERR_FORMAT, err);
        if (!uniq_file.rewind_for_reading()) return fail(LPHQ_ERR_IO, "cannot reopen the temporary file " + uniq_file.path);
        UniqRec u;
        while (uniq_file.next(u))
            if (!pb.push(murmur64a_u64(u.mm, pthash_seed), err)) return fail(LPHQ_ERR_IO, err);
        if (uniq_file.failed) return fail(LPHQ_ERR_IO, "read error on the temporary file " + uniq_file.path);
        if (!pb.finish(err)) return fail(err.find("temporary file") != std::string::npos ? LPHQ_ERR_IO : LPHQ_ERR_FORMAT, err);
    }
    lap("part 2: PTHash over minimizers");

    // ---- Part 3: order by the minimizer MPHF (a bijection onto [0, D)), then the inverted index over that order ----------------
    // The index is written from stretches of the order: the whole of it at once (direct placement in memory), or -- with a memory limit --
    // what a merge of sorted pieces delivers, a stretch at a time (external_memory_vector<mm_triplet_t> mm_sorted_by_mphf,
    // src/partitioned_mphf.cpp:92-104).  A stretch begins at a multiple of 64 (its pieces own whole words of the root bit vector).
    std::vector<uint8_t> lr, mn, left_positions, right_or_collision_sizes, none_sizes, none_positions;  // build_inverted_index, src/partitioned_mphf.cpp:163-268
    std::vector<uint8_t> u_pos, u_size;                                                                  // build_pos_index / build_size_index, src/unpartitioned_mphf.cpp:156-176
    uint64_t universe = 0;
    if (kind == LPHQ_PARTITIONED) bv_init(x.root.bv, D);
    auto index_stretch = [&](uint64_t first, const Triplet* tr, uint64_t count) {
        if (kind != LPHQ_PARTITIONED) {
            const size_t old = u_pos.size();
            u_pos.resize(old + count);
            u_size.resize(old + count);
            parallel_for((size_t)((count + 65535) / 65536), threads, [&](size_t t) {
                for (uint64_t i = t * 65536, e = std::min<uint64_t>(count, i + 65536); i < e; ++i) {
                    u_pos[old + i] = tr[i].p1;
                    u_size[old + i] = tr[i].size;
                }
            });
            return;
        }
        // pieces of a multiple of 64 minimizers, one set of output arrays each, joined in order
        struct Piece {
            std::vector<uint8_t> lr, mn, left_positions, right_or_collision_sizes, none_sizes, none_positions;
            uint64_t universe = 0, n_maximal = 0;
        };
        const uint64_t per = std::max<uint64_t>(1u << 16, ((count + (uint64_t)threads - 1) / (uint64_t)threads + 63) & ~63ull);
        const size_t pieces = (size_t)((count + per - 1) / per);
        std::vector<Piece> part(pieces);
        parallel_for(pieces, threads, [&](size_t pi) {
            Piece& o = part[pi];
            const uint64_t lo = pi * per, hi = std::min<uint64_t>(count, lo + per);
            o.lr.reserve(hi - lo);
            o.mn.reserve(hi - lo);
            for (uint64_t i = lo; i < hi; ++i) {
                const Triplet& t = tr[i];
                if (t.size == 0) {  // RIGHT_OR_COLLISION, size 0 marks a repeated minimizer
                    o.lr.push_back(1);
                    o.right_or_collision_sizes.push_back(0);
                } else if (t.p1 == k - m) {
                    if (t.size == W) {  // MAXIMAL
                        bv_set(x.root.bv, first + i);
                        o.mn.push_back(0);
                        ++o.n_maximal;
                    } else {  // RIGHT
                        o.lr.push_back(1);
                        o.right_or_collision_sizes.push_back(t.size);
                        o.universe += t.size;
                    }
                } else if (t.p1 == t.size - 1) {  // LEFT: stores p1 + 1
                    o.lr.push_back(0);
                    o.left_positions.push_back((uint8_t)(t.p1 + 1));
                    o.universe += (uint64_t)t.p1 + 1;
                } else {  // NONE
                    bv_set(x.root.bv, first + i);
                    o.mn.push_back(1);
                    o.none_positions.push_back(t.p1);
                    o.none_sizes.push_back(t.size);
                    o.universe += (uint64_t)t.p1 + t.size;
                }
            }
        });
        auto join = [&](std::vector<uint8_t>& out, std::vector<uint8_t> Piece::*field) {
            std::vector<size_t> at(pieces + 1, out.size());
            for (size_t pi = 0; pi < pieces; ++pi) at[pi + 1] = at[pi] + (part[pi].*field).size();
            out.resize(at[pieces]);
            parallel_for(pieces, threads, [&](size_t pi) {
                std::copy((part[pi].*field).begin(), (part[pi].*field).end(), out.begin() + at[pi]);
                std::vector<uint8_t>().swap(part[pi].*field);
            });
        };
        join(lr, &Piece::lr);
        join(mn, &Piece::mn);
        join(left_positions, &Piece::left_positions);
        join(right_or_collision_sizes, &Piece::right_or_collision_sizes);
        join(none_sizes, &Piece::none_sizes);
        join(none_positions, &Piece::none_positions);
        for (const Piece& o : part) {
            universe += o.universe;
            x.n_maximal += o.n_maximal;
        }
    };
    std::vector<uint64_t> free_table;
    if (!x.minimizer_order.free_slots.decode(free_table, err)) return fail(LPHQ_ERR_FORMAT, err);
    if (!external) {
        TripletVec by_order(D);  // 0xFF below by all threads: not filled yet
        parallel_for((D + 65535) / 65536, threads, [&](size_t t) { memset((void*)(by_order.data() + t * 65536), 0xFF, (std::min<size_t>(D, (t + 1) * 65536) - t * 65536) * sizeof(Triplet)); });
        std::atomic<bool> bad{false};
        parallel_for((D + 65535) / 65536, threads, [&](size_t t) {
            for (size_t i = t * 65536, e = std::min<size_t>(D, i + 65536); i < e; ++i) {
                uint64_t o = phf_eval(x.minimizer_order, murmur64a_u64(uniq_mm[i], pthash_seed), &free_table);
                if (o >= D) {
                    bad.store(true);
                    continue;
                }
                by_order[o] = uniq[i];
            }
        });
        if (bad.load()) return fail(LPHQ_ERR_FORMAT, "internal: minimizer MPHF out of range");
        std::atomic<bool> hole{false};
        parallel_for((D + 65535) / 65536, threads, [&](size_t t) {
            for (size_t i = t * 65536, e = std::min<size_t>(D, i + 65536); i < e; ++i)
                if (by_order[i].size == 0xFF) hole.store(true);
        });
        if (hole.load()) return fail(LPHQ_ERR_FORMAT, "internal: minimizer MPHF is not a bijection");
        TripletVec().swap(uniq);
        WordVec().swap(uniq_mm);
        lap("part 3: order by minimizer MPHF");
        index_stretch(0, by_order.data(), D);
    } else {
        Spill<OrderRec, OrderVec, OrderKey> order_spill;
        order_spill.dir = spill.dir;
        order_spill.kind = "order";
        const uint32_t order_bits = msb64(D | 1) + 1;
        {
            if (!uniq_file.rewind_for_reading()) return fail(LPHQ_ERR_IO, "cannot reopen the temporary file " + uniq_file.path);
            OrderVec piece;
            piece.reserve(std::min<uint64_t>(ext_cap, D));
            std::atomic<bool> bad{false};
            bool more = true;
            while (more) {
                UniqRec u;
                while (piece.size() < ext_cap && (more = uniq_file.next(u))) piece.push_back(OrderRec{u.mm, u.p1, u.size});
                parallel_for((piece.size() + 4095) / 4096, threads, [&](size_t t) {
                    for (size_t i = t * 4096, e = std::min(piece.size(), i + 4096); i < e; ++i) {
                        piece[i].order = phf_eval(x.minimizer_order, murmur64a_u64(piece[i].order, pthash_seed), &free_table);
                        if (piece[i].order >= D) bad.store(true);
                    }
                });
                if (bad.load()) return fail(LPHQ_ERR_FORMAT, "internal: minimizer MPHF out of range");
                if (!order_spill.flush(piece, threads, order_bits, ext_merge_buf, err)) return fail(LPHQ_ERR_IO, err);
            }
            if (uniq_file.failed) return fail(LPHQ_ERR_IO, "read error on the temporary file " + uniq_file.path);
        }
        lap("part 3: order by minimizer MPHF (sorted pieces)");
        const uint64_t stretch_cap = std::max<uint64_t>(1u << 16, (uint64_t)(ext_cap & ~(size_t)65535));
        TripletVec stretch;
        stretch.reserve(std::min<uint64_t>(stretch_cap, D));
        uint64_t first = 0, seen = 0;
        Spill<OrderRec, OrderVec, OrderKey>::Merger mg;
        if (!mg.open(order_spill.files, ext_merge_buf)) return fail(LPHQ_ERR_IO, mg.err);
        OrderRec r;
        while (mg.next(r)) {
            if (r.order != seen) return fail(LPHQ_ERR_FORMAT, "internal: minimizer MPHF is not a bijection");
            ++seen;
            stretch.push_back(Triplet{r.p1, r.size});
            if (stretch.size() == stretch_cap) {
                index_stretch(first, stretch.data(), stretch.size());
                first += stretch.size();
                stretch.clear();
            }
        }
        if (mg.failed) return fail(LPHQ_ERR_IO, mg.err);
        if (seen != D) return fail(LPHQ_ERR_FORMAT, "internal: minimizer MPHF is not a bijection");
        index_stretch(first, stretch.data(), stretch.size());
    }
    std::vector<uint64_t>().swap(free_table);
    lap("  inverted index: classify by type");

    // Elias-Fano over the prefix sums of n bytes, value(i) each (append_iterator + cumulative_iterator, src/partitioned_mphf.cpp:246-262): the
    // sums are never stored -- per-chunk totals, a scan over the chunks, then every chunk on its own.
    auto ef_encode_prefix_sums = [&](EliasFano& e, uint64_t n, uint64_t total, const std::function<uint8_t(uint64_t)>& value) {
        const uint64_t chunk = 1u << 16;
        const size_t nch = (size_t)((n + chunk - 1) / chunk);
        std::vector<uint64_t> csum(nch + 1, 0);
        parallel_for(nch, threads, [&](size_t c) {
            uint64_t t = 0;
            for (uint64_t i = c * chunk, e2 = std::min(n, i + chunk); i < e2; ++i) t += value(i);
            csum[c + 1] = t;
        });
        for (size_t c = 0; c < nch; ++c) csum[c + 1] += csum[c];
        ef_encode_chunked(e, n, total, true, chunk, threads, [&](size_t c, const std::function<void(uint64_t, uint64_t)>& emit) {
            uint64_t t = csum[c];
            for (uint64_t i = c * chunk, e2 = std::min(n, i + chunk); i < e2; ++i) emit(i, t += value(i));
        });
    };
    if (kind == LPHQ_PARTITIONED) {
        auto pack_bits = [&](BitVector& bv, const std::vector<uint8_t>& bytes) {  // one 64-bit word per step: words are independent
            bv_init(bv, bytes.size());
            const size_t nw = bv.bits.size(), wchunk = 1 << 12;
            parallel_for((nw + wchunk - 1) / wchunk, threads, [&](size_t c) {
                for (size_t w = c * wchunk, e = std::min(nw, w + wchunk); w < e; ++w) {
                    uint64_t word = 0;
                    for (size_t i = w * 64, ie = std::min(bytes.size(), i + 64); i < ie; ++i) word |= (uint64_t)(bytes[i] & 1) << (i & 63);
                    bv.bits[w] = word;
                }
            });
        };
        pack_bits(x.left_right.bv, lr);
        pack_bits(x.max_none.bv, mn);
        std::vector<uint8_t>().swap(lr);
        std::vector<uint8_t>().swap(mn);
        rs_build(x.root, threads);
        rs_build(x.left_right, threads);
        rs_build(x.max_none, threads);
        lap("  inverted index: bit vectors + rank");
        x.right_coll_sizes_start = left_positions.size();
        x.none_sizes_start = x.right_coll_sizes_start + right_or_collision_sizes.size();
        x.none_pos_start = x.none_sizes_start + none_sizes.size();
        const std::vector<uint8_t>* arrays[4] = {&left_positions, &right_or_collision_sizes, &none_sizes, &none_positions};
        const uint64_t starts[5] = {0, x.right_coll_sizes_start, x.none_sizes_start, x.none_pos_start, x.none_pos_start + none_positions.size()};
        ef_encode_prefix_sums(x.sizes_and_positions, starts[4], universe, [&](uint64_t i) -> uint8_t {
            const int a = i < starts[2] ? (i < starts[1] ? 0 : 1) : (i < starts[3] ? 2 : 3);
            return (*arrays[a])[i - starts[a]];
        });
        lap("  inverted index: Elias-Fano over the cumulative sums");
    } else {
        uint64_t pos_sum = 0, size_sum = 0;
        {
            const uint64_t chunk = 1u << 16;
            const size_t nch = (size_t)((D + chunk - 1) / chunk);
            std::vector<uint64_t> ps(nch, 0), ss(nch, 0);
            parallel_for(nch, threads, [&](size_t c) {
                for (uint64_t i = c * chunk, e = std::min(D, i + chunk); i < e; ++i) {
                    ps[c] += u_pos[i];
                    ss[c] += u_size[i];
                }
            });
            for (size_t c = 0; c < nch; ++c) {
                pos_sum += ps[c];
                size_sum += ss[c];
            }
        }
        ef_encode_prefix_sums(x.positions, D, pos_sum, [&](uint64_t i) { return u_pos[i]; });
        ef_encode_prefix_sums(x.sizes, D, size_sum, [&](uint64_t i) { return u_size[i]; });
        x.num_kmers_in_main_index = size_sum;
    }
    lap("part 3: inverted index (wavelet tree, EF)");

    // ---- Part 4: fallback PTHash over the k-mers of repeated minimizers, in input order ----------------
    {
        // second pass over the input (partitioned_mphf.cpp:110-130); the super-k-mers of repeated minimizers by position: sorted in memory,
        // or the merge of the sorted pieces
        std::vector<uint64_t> fh;
        PhfExternalBuilder pb;
        Spill<CollRec, CollVec, CollKey>::Merger mg;
        size_t ci = 0;
        CollRec cw{0, 0, 0};
        bool have = false;
        if (!external) {
            std::sort(colliding.begin(), colliding.end());
            fh.reserve(n_colliding_kmers);
        } else {
            if (!coll_spill.flush(coll_buf, threads, coord_bits, ext_merge_buf, err)) return fail(LPHQ_ERR_IO, err);
            CollVec().swap(coll_buf);
            if (!mg.open(coll_spill.files, ext_merge_buf)) return fail(LPHQ_ERR_IO, mg.err);
            if (!pb.begin(x.fallback_kmer_order, n_colliding_kmers, pthash_seed, c, alpha, spill.dir, max_memory_bytes, threads, device, err)) return fail(LPHQ_ERR_FORMAT, err);
        }
        auto advance = [&]() {  // the next super-k-mer into cw
            if (!external) {
                have = ci < colliding.size();
                if (have) cw = CollRec{colliding[ci].first, colliding[ci].second, 0};
                ++ci;
            } else {
                have = mg.next(cw);
            }
        };
        advance();
        const u128 km_mask = (((u128)1) << (2 * k)) - 1;
        uint64_t at = 0;  // global coordinate of the current record's first base
        const char* const* records = nullptr;
        const size_t* lengths = nullptr;
        size_t nb = 0;
        int rc = 0;
        while (have && (rc = src.next(1, records, lengths, nb)) > 0) {
            for (size_t r = 0; r < nb && have; ++r) {
                const uint64_t end = at + lengths[r];
                for (; have && cw.first < end; advance()) {
                    if (cw.first < at || cw.first + cw.size + k - 1 > end) return fail(LPHQ_ERR_IO, "the second pass over the input does not match the first");
                    const char* s = records[r] + (cw.first - at);
                    u128 km = 0;
                    for (uint32_t j = 0; j < k - 1; ++j) km = (km << 2) | (u128)(nt4((uint8_t)s[j]) & 3);
                    for (uint32_t w = 0; w < cw.size; ++w) {
                        km = ((km << 2) | (u128)(nt4((uint8_t)s[w + k - 1]) & 3)) & km_mask;
                        // fallback_hasher, include/constants.hpp:56-70
                        uint64_t h;
                        if (kmer_bits == 64) h = murmur64a_u64((uint64_t)km, pthash_seed);
                        else h = murmur64a_u64((uint64_t)km, pthash_seed) ^ murmur64a_u64((uint64_t)(km >> 64), ~pthash_seed);
                        if (!external) fh.push_back(h);
                        else if (!pb.push(h, err)) return fail(LPHQ_ERR_IO, err);
                    }
                }
                at = end;
            }
        }
        if (rc < 0) return fail(LPHQ_ERR_IO, "the record source failed during the second pass");
        if (external && mg.failed) return fail(LPHQ_ERR_IO, mg.err);
        if (have) return fail(LPHQ_ERR_IO, "the second pass over the input is shorter than the first");
        if (!external) {
            if (!phf_build(x.fallback_kmer_order, fh, pthash_seed, c, alpha, threads, device, err)) return fail(LPHQ_ERR_FORMAT, err);
        } else {
            if (!pb.finish(err)) return fail(err.find("temporary file") != std::string::npos ? LPHQ_ERR_IO : LPHQ_ERR_FORMAT, err);
            coll_spill.drop();
        }
    }

    lap("part 4: fallback MPHF");
    // ---- essentials::save ----------------------------------------------------------------------------
    Writer w;
    if (!w.open(out_path)) return fail(LPHQ_ERR_IO, std::string("cannot write ") + out_path);
    w.raw(&x.k, 1);
    w.raw(&x.m, 1);
    w.u64(x.mm_seed);
    w.u64(x.nkmers);
    w.u64(x.distinct_minimizers);
    if (kind == LPHQ_PARTITIONED) {
        w.u64(x.n_maximal);
        w.u64(x.right_coll_sizes_start);
        w.u64(x.none_sizes_start);
        w.u64(x.none_pos_start);
        w.put(x.minimizer_order);
        w.put(x.root);
        w.put(x.left_right);
        w.put(x.max_none);
        w.put(x.sizes_and_positions);
        w.put(x.fallback_kmer_order);
    } else {
        w.u64(x.num_kmers_in_main_index);
        w.put(x.minimizer_order);
        w.put(x.positions);
        w.put(x.sizes);
        w.put(x.fallback_kmer_order);
    }
    if (!w.close()) {
        ::remove(out_path);  // (not a file a loader should ever meet)
        return fail(LPHQ_ERR_IO, std::string("short write on ") + out_path);
    }
    lap("save");
    if (stats) {
        stats->num_kmers = nkmers;
        stats->num_records = n_records;
        stats->num_super_kmers = n_super_kmers;
        stats->distinct_minimizers = D;
        stats->colliding_super_kmers = n_colliding;
        stats->fallback_kmers = x.fallback_kmer_order.num_keys;
        stats->file_bytes = w.bytes;
    }
    return LPHQ_OK;
}

// MurmurHash2_64 of the 16 bytes of a __uint128_t (low word first)
static uint64_t murmur64a_u128(u128 v, uint64_t seed) {
    const uint64_t M = 0xc6a4a7935bd1e995ULL;
    uint64_t h = seed ^ (16ull * M);
    for (int i = 0; i < 2; ++i) {
        uint64_t k = (uint64_t)(v >> (64 * i)) * M;
        k ^= k >> 47;
        k *= M;
        h ^= k;
        h *= M;
    }
    h ^= h >> 47;
    h *= M;
    h ^= h >> 47;
    return h;
}

// Competitor harness, PTHash half: ptbb_build -i <fasta> -k k -p <out>  (src/ptbb/ptbb_build.cpp:83-99; keys as ptbb_file_itr yields
// them, src/ptbb/ptbb.hpp:108-131: every window of k valid bases, forward strand; seed 1, alpha 0.94, minimal output)
extern "C" int lphq_build_pthash(const char* const* records, const size_t* lengths, size_t n_records, int k_, int kmer_bits, double c, int threads,
                                 int device, const char* out_path, lphq_build_stats* stats) {
    auto fail = [&](int code, const std::string& msg) {
        set_last_error(msg);
        return code;
    };
    if ((n_records && (!records || !lengths)) || !out_path) return fail(LPHQ_ERR_ARG, "null argument");
    if (kmer_bits != 64 && kmer_bits != 128) return fail(LPHQ_ERR_ARG, "kmer_bits must be 64 or 128");
    if (k_ < 1 || k_ > kmer_bits / 2 - 1) return fail(LPHQ_ERR_ARG, "k cannot be larger than " + std::to_string(kmer_bits / 2 - 1));
    if (c < 3.0 || c > 10.0) return fail(LPHQ_ERR_ARG, "3.0 <= c <= 10.0");
    if (threads <= 0) threads = usable_cpus();
    const uint32_t k = (uint32_t)k_;
    const uint64_t seed = 1;  // constants::default_pthash_seed
    const u128 km_mask = (((u128)1) << (2 * k)) - 1;
    // pass 1: k-mers per record; pass 2: their hash128, in input order
    std::vector<uint64_t> first_of(n_records + 1, 0);
    parallel_for(n_records, threads, [&](size_t r) {
        uint64_t since = 0, n = 0;
        for (size_t i = 0; i < lengths[r]; ++i) {
            if (nt4((uint8_t)records[r][i]) < 4) n += ++since >= k;
            else since = 0;
        }
        first_of[r + 1] = n;
    });
    for (size_t r = 0; r < n_records; ++r) first_of[r + 1] += first_of[r];
    const uint64_t n = first_of[n_records];
    std::vector<uint64_t> firsts(n), seconds(n);
    parallel_for(n_records, threads, [&](size_t r) {
        uint64_t since = 0, at = first_of[r];
        u128 km = 0;
        for (size_t i = 0; i < lengths[r]; ++i) {
            const uint8_t cb = nt4((uint8_t)records[r][i]);
            if (cb < 4) {
                km = ((km << 2) | (u128)cb) & km_mask;
                if (++since >= k) {  // ptbb::PTHasher::hash, src/ptbb/ptbb.hpp:16-23
                    if (kmer_bits == 64) {
                        firsts[at] = murmur64a_u64((uint64_t)km, seed);
                        seconds[at] = murmur64a_u64((uint64_t)km, ~seed);
                    } else {
                        firsts[at] = murmur64a_u128(km, seed);
                        seconds[at] = murmur64a_u128(km, ~seed);
                    }
                    ++at;
                }
            } else {
                since = 0;
            }
        }
    });
    SinglePhf f;
    std::string err;
    if (!phf_build(f, seconds, seed, c, 0.94, threads, device, err, &firsts)) return fail(LPHQ_ERR_FORMAT, err);
    Writer w;
    if (!w.open(out_path)) return fail(LPHQ_ERR_IO, std::string("cannot write ") + out_path);
    w.put(f);
    if (!w.close()) {
        ::remove(out_path);  // (not a file a loader should ever meet)
        return fail(LPHQ_ERR_IO, std::string("short write on ") + out_path);
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->num_kmers = n;
        stats->num_records = n_records;
        stats->file_bytes = w.bytes;
    }
    return LPHQ_OK;
}

static int build_memory(const char* const* records, const size_t* lengths, size_t n_records, int k, int m, uint64_t mm_seed, double c, int kind,
                        int kmer_bits, int threads, int device, const char* out_path, lphq_build_stats* stats) {
    if (n_records && (!records || !lengths)) {
        set_last_error("null argument");
        return LPHQ_ERR_ARG;
    }
    MemorySource src(records, lengths, n_records);
    return build_impl(src, k, m, mm_seed, c, kind, kmer_bits, threads, device, nullptr, 0, out_path, stats);
}

extern "C" int lphq_build(const char* const* records, const size_t* lengths, size_t n_records, int k, int m, uint64_t mm_seed, double c, int kind,
                          int kmer_bits, int threads, const char* out_path, lphq_build_stats* stats) {
    return build_memory(records, lengths, n_records, k, m, mm_seed, c, kind, kmer_bits, threads, -1, out_path, stats);
}

extern "C" int lphq_build_stream(lphq_record_source next, void* ctx, int k, int m, uint64_t mm_seed, double c, int kind, int kmer_bits, int threads, int device,
                                 const char* tmp_dir, uint64_t max_memory_bytes, const char* out_path, lphq_build_stats* stats) {
    if (!next) {
        set_last_error("null argument");
        return LPHQ_ERR_ARG;
    }
    CallbackSource src(next, ctx);
    return build_impl(src, k, m, mm_seed, c, kind, kmer_bits, threads, device < 0 ? -1 : device, tmp_dir, max_memory_bytes, out_path, stats);
}

extern "C" int lphq_build_device(const char* const* records, const size_t* lengths, size_t n_records, int k, int m, uint64_t mm_seed, double c, int kind,
                                 int kmer_bits, int threads, int device, const char* out_path, lphq_build_stats* stats) {
    return build_memory(records, lengths, n_records, k, m, mm_seed, c, kind, kmer_bits, threads, device < 0 ? -1 : device, out_path, stats);
}
