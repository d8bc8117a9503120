// bbhash.cpp -- see bbhash.hpp.  Host side only (no HIP): parsing, image building and construction of BBHash functions.
#include "bbhash.hpp"
#include "host_cpus.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <numeric>
#include <thread>

namespace lphq {
namespace {

typedef unsigned __int128 u128;

struct Rd {
    const uint8_t* p;
    size_t len, off = 0;
    bool bad = false;
    void raw(void* dst, size_t n) {
        if (n == 0) return;
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
};

// the level sizes of boomphf::mphf::setup() / load()  (BooPHF.hpp:1027-1043, 979-989)
void level_domains(double gamma, uint64_t nelem, int nb_levels, std::vector<uint64_t>& dom) {
    if (nelem == 0) {  // no key (the formula below divides by zero): every level is the minimum, one 64-bit word -- what the builder writes
        dom.assign((size_t)nb_levels, 64);
        return;
    }
    const double pc = 1.0 - pow(((gamma * (double)nelem - 1) / (gamma * (double)nelem)), (double)(nelem - 1));
    const uint64_t hash_domain = (size_t)(ceil(double(nelem) * gamma));
    dom.resize((size_t)nb_levels);
    for (int ii = 0; ii < nb_levels; ++ii) {
        uint64_t d = (((uint64_t)(hash_domain * pow(pc, ii)) + 63) / 64) * 64;
        if (d == 0) d = 64;
        dom[(size_t)ii] = d;
    }
}

uint64_t murmur64a(const uint64_t* words, int nwords, uint64_t seed) {  // MurmurHash64A over 8 * nwords bytes
    const uint64_t m = 0xc6a4a7935bd1e995ULL;
    uint64_t h = seed ^ ((uint64_t)(8 * nwords) * m);
    for (int i = 0; i < nwords; ++i) {
        uint64_t k = words[i] * m;
        k ^= k >> 47;
        k *= m;
        h ^= k;
        h *= m;
    }
    h ^= h >> 47;
    h *= m;
    h ^= h >> 47;
    return h;
}
// ptbb::BBHasher (src/ptbb/ptbb.hpp:25-32): hash128{Murmur(key, seed), Murmur(key, ~seed)}.mix(), mix = first ^ second (PTHash v1)
inline uint64_t bb_hash(const uint64_t* key, int nwords, uint64_t seed) { return murmur64a(key, nwords, seed) ^ murmur64a(key, nwords, ~seed); }
inline uint64_t fastrange64(uint64_t word, uint64_t p) { return (uint64_t)(((u128)word * (u128)p) >> 64); }

struct Wr {
    std::vector<uint8_t>& b;
    void raw(const void* p, size_t n) {
        const uint8_t* c = (const uint8_t*)p;
        b.insert(b.end(), c, c + n);
    }
    void u64(uint64_t v) { raw(&v, 8); }
};

inline uint8_t nt4(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}

}  // namespace

bool parse_bbhash(const uint8_t* buf, size_t len, int k, int kmer_bits, BbhFile& f, std::string& err) {
    if (kmer_bits != 64 && kmer_bits != 128) {
        err = "bbhash: kmer_bits must be 64 or 128";
        return false;
    }
    if (k < 1 || k > kmer_bits / 2 - 1) {
        err = "bbhash: k does not fit the k-mer type";
        return false;
    }
    Rd r{buf, len};
    f = BbhFile();
    f.k = k;
    f.kmer_bits = kmer_bits;
    r.raw(&f.gamma, 8);
    r.raw(&f.nb_levels, 4);
    f.last_bitset_rank = r.u64();
    f.nelem = r.u64();
    if (r.bad || f.nb_levels < 2 || f.nb_levels > 1024 || !(f.gamma >= 1.0) || f.gamma > 1e6) {
        err = "bbhash: implausible header (not a boomphf::mphf::save file?)";
        return false;
    }
    f.levels.resize((size_t)f.nb_levels);
    std::vector<uint64_t> dom;
    level_domains(f.gamma, f.nelem, f.nb_levels, dom);
    uint64_t ones = 0;
    for (int ii = 0; ii < f.nb_levels; ++ii) {
        BbhFile::Level& lv = f.levels[(size_t)ii];
        lv.size = r.u64();
        const uint64_t nchar = r.u64();
        if (r.bad || nchar != 1 + lv.size / 64 || nchar > (len - r.off) / 8) {
            err = "bbhash: level " + std::to_string(ii) + ": bit array header does not fit the file";
            return false;
        }
        lv.words.resize(nchar);
        r.raw(lv.words.data(), nchar * 8);
        const uint64_t sizer = r.u64();
        if (r.bad || sizer != (nchar + 7) / 8 || sizer > (len - r.off) / 8) {
            err = "bbhash: level " + std::to_string(ii) + ": rank samples do not match the bit array";
            return false;
        }
        lv.ranks.resize(sizer);
        r.raw(lv.ranks.data(), sizer * 8);
        lv.hash_domain = dom[(size_t)ii];
        if (lv.size != lv.hash_domain) {  // the redundancy load() relies on: sizes recomputed from gamma and nelem must be the stored ones
            err = "bbhash: level " + std::to_string(ii) + ": stored size differs from the size implied by gamma and the number of keys";
            return false;
        }
        for (uint64_t w = 0; w < nchar; ++w) {  // the rank samples are redundant too
            if (w % 8 == 0 && lv.ranks[w / 8] != ones) {
                err = "bbhash: level " + std::to_string(ii) + ": rank sample disagrees with the bits";
                return false;
            }
            ones += (uint64_t)__builtin_popcountll(lv.words[w]);
        }
    }
    if (ones != f.last_bitset_rank) {
        err = "bbhash: total of the bit arrays differs from the stored last rank";
        return false;
    }
    const uint64_t nfinal = r.u64();
    const int nw = kmer_bits / 64;
    if (r.bad || nfinal > (len - r.off) / (size_t)(8 * nw + 8) || ones + nfinal != f.nelem) {
        err = "bbhash: final map does not complete the key count";
        return false;
    }
    std::vector<std::pair<std::pair<uint64_t, uint64_t>, uint64_t>> fin(nfinal);
    for (uint64_t i = 0; i < nfinal; ++i) {
        uint64_t kw[2] = {0, 0};
        r.raw(kw, (size_t)(8 * nw));
        fin[i] = {{kw[1], kw[0]}, r.u64()};
    }
    if (r.bad || r.off != len) {
        err = "bbhash: truncated file or trailing bytes";
        return false;
    }
    std::sort(fin.begin(), fin.end());
    for (auto& e : fin) {
        f.final_keys.push_back(e.first.second);
        if (nw == 2) f.final_keys.push_back(e.first.first);
        f.final_vals.push_back(e.second);
    }
    f.bytes_read = r.off;
    return true;
}

bool build_bbhash_image(const BbhFile& f, std::vector<uint8_t>& img, std::string& err) {
    (void)err;
    img.clear();
    auto reserve = [&](size_t bytes) {
        size_t off = (img.size() + 255) & ~(size_t)255;
        img.resize(off + bytes + 64, 0);
        return (uint64_t)off;
    };
    reserve(sizeof(lphq_image_header));
    lphq_image_header h;
    memset(&h, 0, sizeof h);
    h.magic = LPHQ_IMAGE_MAGIC;
    h.kind = 3;
    h.kmer_bits = (uint32_t)f.kmer_bits;
    h.k = h.m = (uint32_t)f.k;
    h.W = 1;
    h.P = 4;
    h.nkmers = f.nelem;
    h.n_maximal = (uint64_t)f.nb_levels;
    h.collision_base = f.last_bitset_rank;
    h.fallback_kmers = f.final_vals.size();
    h.off_uni = reserve((size_t)f.nb_levels * sizeof(lphq_bbh_level));
    for (int ii = 0; ii < f.nb_levels; ++ii) {
        const BbhFile::Level& lv = f.levels[(size_t)ii];
        const uint64_t ob = reserve(lv.words.size() * 8), orr = reserve(lv.ranks.size() * 8);
        memcpy(img.data() + ob, lv.words.data(), lv.words.size() * 8);
        memcpy(img.data() + orr, lv.ranks.data(), lv.ranks.size() * 8);
        lphq_bbh_level d;
        d.hash_domain = lv.hash_domain;
        d.off_bits = ob;
        d.off_ranks = orr;
        d.nwords = lv.words.size();
        memcpy(img.data() + h.off_uni + (size_t)ii * sizeof d, &d, sizeof d);
    }
    // exact map of the last level as an open-addressing table (linear probing, load <= 1/2): capacity 2^reserved0 slots, the keys
    // (kmer_bits / 64 words each) followed by the values, ~0 = empty slot.  (The reference keeps a std::unordered_map.)
    const size_t nfin = f.final_vals.size(), nw = (size_t)f.kmer_bits / 64;
    uint32_t lg = 1;
    while (((size_t)1 << lg) < 2 * nfin) ++lg;
    const size_t cap = (size_t)1 << lg;
    h.reserved0 = lg;
    h.off_upair = reserve(cap * nw * 8 + cap * 8);
    {
        uint64_t* keys = (uint64_t*)(img.data() + h.off_upair);
        uint64_t* vals = keys + cap * nw;
        for (size_t i = 0; i < cap; ++i) vals[i] = ~0ull;
        for (size_t i = 0; i < nfin; ++i) {
            const uint64_t lo = f.final_keys[i * nw], hi = nw == 2 ? f.final_keys[i * 2 + 1] : 0;
            size_t slot = (size_t)(bbh_slot_hash(lo, hi) >> (64 - lg));
            while (vals[slot] != ~0ull) slot = (slot + 1) & (cap - 1);
            keys[slot * nw] = lo;
            if (nw == 2) keys[slot * 2 + 1] = hi;
            vals[slot] = f.final_vals[i];
        }
    }
    img.resize((img.size() + 255) & ~(size_t)255, 0);
    h.total_bytes = img.size();
    memcpy(img.data(), &h, sizeof h);
    return true;
}

bool build_bbhash_file(const char* const* records, const size_t* lengths, size_t n_records, int k_, int kmer_bits, double gamma, int threads,
                       std::vector<uint8_t>& out, uint64_t& nkeys, std::string& err) {
    if (kmer_bits != 64 && kmer_bits != 128) {
        err = "kmer_bits must be 64 or 128";
        return false;
    }
    if (k_ < 1 || k_ > kmer_bits / 2 - 1) {
        err = "k cannot be larger than " + std::to_string(kmer_bits / 2 - 1);
        return false;
    }
    if (!(gamma >= 1.0)) {
        err = "BBHash gamma factor < 1";  // ptbb_build.cpp:149
        return false;
    }
    if (threads <= 0) threads = usable_cpus();
    const uint32_t k = (uint32_t)k_;
    const int nw = kmer_bits / 64;
    const u128 mask = (((u128)1) << (2 * k)) - 1;
    // keys: every window of k valid bases, input order
    std::vector<uint64_t> first_of(n_records + 1, 0);
    auto parallel = [&](size_t n, const std::function<void(size_t)>& fn) {
        const int T = (int)std::min<size_t>((size_t)threads, std::max<size_t>(1, n));
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t)
            th.emplace_back([&, t] {
                for (size_t i = (size_t)t; i < n; i += (size_t)T) fn(i);
            });
        for (size_t i = 0; i < n; i += (size_t)T) fn(i);
        for (auto& x : th) x.join();
    };
    parallel(n_records, [&](size_t r) {
        uint64_t since = 0, n = 0;
        for (size_t i = 0; i < lengths[r]; ++i) {
            if (nt4((uint8_t)records[r][i]) < 4) n += ++since >= k;
            else since = 0;
        }
        first_of[r + 1] = n;
    });
    for (size_t r = 0; r < n_records; ++r) first_of[r + 1] += first_of[r];
    const uint64_t n = first_of[n_records];
    nkeys = n;
    std::vector<uint64_t> keys(n * (size_t)nw);
    parallel(n_records, [&](size_t r) {
        uint64_t since = 0, at = first_of[r];
        u128 km = 0;
        for (size_t i = 0; i < lengths[r]; ++i) {
            const uint8_t c = nt4((uint8_t)records[r][i]);
            if (c < 4) {
                km = ((km << 2) | (u128)c) & mask;
                if (++since >= k) {
                    keys[at * (size_t)nw] = (uint64_t)km;
                    if (nw == 2) keys[at * 2 + 1] = (uint64_t)(km >> 64);
                    ++at;
                }
            } else
                since = 0;
        }
    });
    // boomphf::mphf constructor (BooPHF.hpp:728-784): 25 levels; at level i every key that no earlier level holds sets bit
    // fastrange64(hash_i, domain_i); bits hit twice are cleared again, their keys go on; what reaches the last level is mapped exactly
    const int nb_levels = 25;
    std::vector<uint64_t> dom;
    level_domains(gamma, n, nb_levels, dom);
    struct St { uint64_t s0, s1; };
    std::vector<St> st(n);           // xorshift state of every key still looking for a level
    std::vector<uint64_t> live(n);   // indices of those keys
    std::iota(live.begin(), live.end(), 0);
    Wr w{out};
    out.clear();
    w.raw(&gamma, 8);
    const int32_t nl = nb_levels;
    w.raw(&nl, 4);
    const size_t pos_last_rank = out.size();
    w.u64(0);
    w.u64(n);
    uint64_t offset = 0;
    std::vector<uint64_t> hv;
    for (int ii = 0; ii < nb_levels; ++ii) {
        const uint64_t d = dom[(size_t)ii], nchar = 1 + d / 64;
        std::vector<uint64_t> bits(nchar, 0), coll(nchar, 0);
        if (ii < nb_levels - 1) {
            hv.resize(live.size());
            parallel((live.size() + 65535) / 65536, [&](size_t c) {
                for (size_t j = c * 65536, e = std::min(live.size(), j + 65536); j < e; ++j) {
                    const uint64_t key = live[j];
                    St& s = st[key];
                    uint64_t hsh;
                    if (ii == 0) hsh = s.s0 = bb_hash(&keys[key * (size_t)nw], nw, 0xAAAAAAAA55555555ULL);
                    else if (ii == 1) hsh = s.s1 = bb_hash(&keys[key * (size_t)nw], nw, 0x33333333CCCCCCCCULL);
                    else {  // XorshiftHashFunctors::next (BooPHF.hpp:417-423)
                        uint64_t s1 = s.s0;
                        const uint64_t s0 = s.s1;
                        s.s0 = s0;
                        s1 ^= s1 << 23;
                        s.s1 = s1 ^ s0 ^ (s1 >> 17) ^ (s0 >> 26);
                        hsh = s.s1 + s0;
                    }
                    const uint64_t pos = fastrange64(hsh, d);
                    hv[j] = pos;
                    const uint64_t bit = 1ULL << (pos & 63);
                    if (__sync_fetch_and_or(&bits[pos >> 6], bit) & bit) __sync_fetch_and_or(&coll[pos >> 6], bit);
                }
            });
            for (uint64_t x = 0; x < nchar; ++x) bits[x] &= ~coll[x];
            size_t keep = 0;
            for (size_t j = 0; j < live.size(); ++j)
                if (!((bits[hv[j] >> 6] >> (hv[j] & 63)) & 1)) live[keep++] = live[j];
            live.resize(keep);
        }
        std::vector<uint64_t> ranks;
        for (uint64_t x = 0; x < nchar; ++x) {
            if (x % 8 == 0) ranks.push_back(offset);
            offset += (uint64_t)__builtin_popcountll(bits[x]);
        }
        w.u64(d);
        w.u64(nchar);
        w.raw(bits.data(), nchar * 8);
        w.u64(ranks.size());
        w.raw(ranks.data(), ranks.size() * 8);
    }
    memcpy(out.data() + pos_last_rank, &offset, 8);
    w.u64(live.size());
    for (size_t j = 0; j < live.size(); ++j) {  // exact map of the last level: value = index in input order (the reference: arrival order)
        w.raw(&keys[live[j] * (size_t)nw], (size_t)(8 * nw));
        w.u64((uint64_t)j);
    }
    return true;
}

}  // namespace lphq
