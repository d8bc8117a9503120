// image.cpp -- re-pack a parsed `.lph` (lph_format.hpp) into the flat device image (image_layout.h).
#include "image.hpp"

#include <cstdio>
#include <cstring>

#include "../../include/lphq.h"

namespace lphq {
namespace {

struct ImageWriter {
    std::vector<uint8_t>& img;
    explicit ImageWriter(std::vector<uint8_t>& v) : img(v) {}
    uint64_t reserve(size_t bytes) {  // 256-B aligned section, zero filled, 64 B of slack after it
        size_t off = (img.size() + 255) & ~(size_t)255;
        img.resize(off + bytes + 64, 0);
        return off;
    }
    template <typename T>
    T* at(uint64_t off) {
        return reinterpret_cast<T*>(img.data() + off);
    }
};

lphq_divisor make_divisor(uint64_t d) {
    lphq_divisor r;
    memset(&r, 0, sizeof r);
    r.d64 = d;
    if (d != 0 && d <= 0xFFFFFFFFull) {
        r.d = (uint32_t)d;
        r.shift = (uint32_t)__builtin_clz((uint32_t)d);
        r.dn = (uint32_t)d << r.shift;
        r.inv = (uint32_t)(~(uint64_t)0 / r.dn - (1ull << 32));
    }
    return r;
}

bool build_phf(const SinglePhf& f, ImageWriter& w, lphq_phf_desc& d, const char* what, std::string& err, std::vector<uint64_t>* free_slots_out = nullptr) {
    memset(&d, 0, sizeof d);
    d.seed = f.seed;
    d.num_keys = f.num_keys;
    d.table_size = f.table_size;
    d.num_dense = f.num_dense;
    d.num_sparse = f.num_sparse;
    d.front_size = f.front.ranks.size;
    d.div_table = make_divisor(f.table_size);
    d.div_dense = make_divisor(f.num_dense);
    d.div_sparse = make_divisor(f.num_sparse);
    if (f.num_keys == 0) return true;
    if (f.num_keys > 0xFFFFFFFFull || f.table_size > 0xFFFFFFFFull) {
        err = std::string(what) + ": more than 2^32-1 keys is beyond the device layout's u32 free-slot table";
        return false;
    }
    if (f.num_dense > 0xFFFFFFFFull || f.num_sparse > 0xFFFFFFFFull || f.num_dense + f.num_sparse > 0xFFFFFFFFull) {
        err = std::string(what) + ": more than 2^32-1 buckets is beyond the device layout's 32-bit bucket arithmetic";
        return false;
    }
    if (f.front.ranks.width > 32 || f.back.ranks.width > 32) {
        err = std::string(what) + ": pilot codewords wider than 32 bits";
        return false;
    }
    d.front_width = (uint32_t)f.front.ranks.width;
    d.back_width = (uint32_t)f.back.ranks.width;
    d.n_front_dict = (uint32_t)f.front.dict.size;
    d.n_back_dict = (uint32_t)f.back.dict.size;
    const Dictionary* dicts[2] = {&f.front, &f.back};
    uint64_t* off_ranks[2] = {&d.off_front_ranks, &d.off_back_ranks};
    uint64_t* off_hp[2] = {&d.off_front_hp, &d.off_back_hp};
    for (int s = 0; s < 2; ++s) {
        const Dictionary& dc = *dicts[s];
        size_t nw = dc.ranks.bits.size() + 1;  // one more padding word: the kernel reads two dwords
        *off_ranks[s] = w.reserve(nw * 8);
        memcpy(w.at<uint64_t>(*off_ranks[s]), dc.ranks.bits.data(), dc.ranks.bits.size() * 8);
        *off_hp[s] = w.reserve((dc.dict.size ? dc.dict.size : 1) * 8);
        uint64_t* hp = w.at<uint64_t>(*off_hp[s]);
        // single_phf::position: hashed_pilot = default_hash64(pilot, m_seed)
        for (uint64_t i = 0; i < dc.dict.size; ++i) hp[i] = murmur64a_u64(dc.dict.access(i), f.seed);
    }
    std::vector<uint64_t> fs;
    if (!f.free_slots.decode(fs, err)) {
        err = std::string(what) + " free slots: " + err;
        return false;
    }
    d.off_free = w.reserve((fs.size() ? fs.size() : 1) * 4);
    uint32_t* ft = w.at<uint32_t>(d.off_free);
    for (size_t i = 0; i < fs.size(); ++i) {
        if (fs[i] >= f.num_keys) {
            err = std::string(what) + ": free slot >= num_keys";
            return false;
        }
        ft[i] = (uint32_t)fs[i];
    }
    if (free_slots_out) free_slots_out->swap(fs);
    return true;
}

// (base, mode) of a lookup as the kernels' LookupCtx holds it, packed for the tail table (image_layout.h)
bool pack_ctx(uint64_t base, uint32_t mode, uint64_t& out, std::string& err) {
    if (base > LPHQ_TAIL_BASE_MASK) {
        err = "hash codes beyond 2^62 are beyond the device layout's tail entries";
        return false;
    }
    out = ((uint64_t)mode << LPHQ_TAIL_MODE_SHIFT) | base;
    return true;
}
// tail[t] = ctx of minimizer order fs[t]: orders are visited in increasing order by the callers, fs is non-decreasing (prefix sums of an Elias-Fano code)
struct TailFiller {
    const std::vector<uint64_t>& fs;
    uint64_t* tail;
    size_t t = 0;
    bool put(uint64_t order, uint64_t base, uint32_t mode, std::string& err) {
        while (t < fs.size() && fs[t] == order) {
            if (!pack_ctx(base, mode, tail[t], err)) return false;
            ++t;
        }
        return true;
    }
};

// pair blocks: size planes w[0..P), position planes w[P..2P), 32 entries
bool fill_pair_blocks(ImageWriter& w, uint64_t off, const uint64_t* size_prefix, const uint64_t* pos_prefix, uint64_t n, uint64_t base, uint32_t P,
                      std::string& err) {
    const uint32_t E = LPHQ_PAIR_PER_BLOCK;
    uint64_t nblocks = (n + E - 1) / E;
    for (uint64_t b = 0; b < nblocks; ++b) {
        lphq_plane_block* blk = w.at<lphq_plane_block>(off) + b;
        blk->abs = base + size_prefix[b * E];
        for (uint32_t j = 0; j < E && b * E + j < n; ++j) {
            uint64_t sz = size_prefix[b * E + j + 1] - size_prefix[b * E + j];
            uint64_t ps = pos_prefix[b * E + j + 1] - pos_prefix[b * E + j];
            if ((sz >> P) || (ps >> P)) {
                err = "stored size/position does not fit the plane width";
                return false;
            }
            for (uint32_t p = 0; p < P; ++p) {
                if ((sz >> p) & 1) blk->w[p] |= 1u << j;
                if ((ps >> p) & 1) blk->w[P + p] |= 1u << j;
            }
        }
    }
    return true;
}

}  // namespace

bool build_image(const LphFile& x, std::vector<uint8_t>& img, std::string& err) {
    img.clear();
    ImageWriter w(img);
    w.reserve(sizeof(lphq_image_header));
    lphq_image_header h;
    memset(&h, 0, sizeof h);
    h.magic = LPHQ_IMAGE_MAGIC;
    h.kind = (uint32_t)x.kind;
    h.kmer_bits = (uint32_t)x.kmer_bits;
    h.k = x.k;
    h.m = x.m;
    h.W = (uint32_t)x.k - x.m + 1;
    h.P = h.W <= 15 ? 4 : 6;  // stored sizes/positions are <= W <= 63
    h.mm_seed = x.mm_seed;
    h.nkmers = x.nkmers;
    h.distinct_minimizers = x.distinct_minimizers;
    h.fallback_kmers = x.fallback_kmer_order.num_keys;
    h.n_maximal = x.n_maximal;
    const uint64_t D = x.distinct_minimizers;
    if (D > 0xFFFFFFFFull) {
        err = "more than 2^32-1 distinct minimizers is beyond the device layout's u32 rank counters";
        return false;
    }
    std::vector<uint64_t> mm_free;  // free_slots of the minimizer function, decoded
    if (!build_phf(x.minimizer_order, w, h.mm, "minimizer_order", err, &mm_free)) return false;
    for (size_t i = 1; i < mm_free.size(); ++i)
        if (mm_free[i] < mm_free[i - 1]) {
            err = "minimizer_order free slots are not monotone";
            return false;
        }
    if (!build_phf(x.fallback_kmer_order, w, h.fb, "fallback_kmer_order", err)) return false;

    if (x.kind == 2) {
        // bare PTHash over k-mers (competitor harness): the function is all there is
    } else if (x.kind == LPHQ_PARTITIONED) {
        const uint64_t base = (uint64_t)h.W * x.n_maximal;  // (k - m + 1) * n_maximal, partitioned_mphf.cpp:300
        std::vector<uint64_t> vals;                          // EF[0..n]: prefix sums with the leading 0
        if (!x.sizes_and_positions.decode(vals, err)) return false;
        if (vals.empty()) vals.push_back(0);
        const uint64_t rcs = x.right_coll_sizes_start, nss = x.none_sizes_start, nps = x.none_pos_start;
        const uint64_t n_left = rcs, n_right = nss - rcs, n_none = nps - nss;
        if (vals.size() != nps + n_none + 1) {
            err = "sizes_and_positions: entry count does not match the region starts";
            return false;
        }
        h.collision_base = vals[nps] + base;  // sizes_and_positions.access(none_pos_start) + W*n_maximal, partitioned_mphf.cpp:309-311

        // fused blocks, in minimizer order (src/partitioned_mphf.cpp:183-267 builds the four arrays in this order)
        const uint32_t T = LPHQ_UNI_T(h.P), E = LPHQ_UNI_PER_BLOCK(h.P), P = h.P;
        const uint64_t nblk = (D + E - 1) / E;
        h.off_uni = w.reserve((nblk + 1) * sizeof(lphq_uni_block));
        h.off_tail = w.reserve((mm_free.size() ? mm_free.size() : 1) * 8);
        TailFiller tailf{mm_free, w.at<uint64_t>(h.off_tail)};
        uint64_t n_max_seen = 0, i_l = 0, i_r = 0, i_n = 0, i_lr = 0, i_mn = 0;
        for (uint64_t i = 0; i < D; ++i) {
            lphq_uni_block* blk = w.at<lphq_uni_block>(h.off_uni) + i / E;
            const uint32_t j = (uint32_t)(i % E);
            if (j == 0) {
                blk->base[0] = (uint64_t)h.W * n_max_seen;
                blk->base[1] = base + vals[i_l];
                blk->base[2] = base + vals[rcs + i_r];
                blk->base[3] = base + vals[nss + i_n];
            }
            const bool msb = x.root.bv.get(i);
            const bool lsb = msb ? x.max_none.bv.get(i_mn++) : x.left_right.bv.get(i_lr++);
            uint64_t size = 0, pos = 0;
            uint64_t cbase;  // what the kernels' ctx_partitioned() makes of this minimizer (src/partitioned_mphf.cpp:297-336), for the tail table
            uint32_t cmode = 0;
            if (!msb && !lsb) {  // LEFT
                size = vals[i_l + 1] - vals[i_l];
                cbase = base + vals[i_l];
                ++i_l;
            } else if (!msb && lsb) {  // RIGHT_OR_COLLISION
                size = vals[rcs + i_r + 1] - vals[rcs + i_r];
                if (size == 0) {
                    cbase = h.collision_base;
                    cmode = 2;
                } else {
                    cbase = base + vals[rcs + i_r] + (uint64_t)(x.k - x.m);
                    cmode = 1;
                }
                ++i_r;
            } else if (msb && !lsb) {  // MAXIMAL
                cbase = (uint64_t)h.W * n_max_seen;
                ++n_max_seen;
            } else {  // NONE
                size = vals[nss + i_n + 1] - vals[nss + i_n];
                pos = vals[nps + i_n + 1] - vals[nps + i_n];
                cbase = base + vals[nss + i_n] + pos;
                cmode = 1;
                ++i_n;
            }
            if (!tailf.put(i, cbase, cmode, err)) return false;
            if ((size >> P) || (pos >> P)) {
                err = "stored size/position does not fit the plane width (value > k-m+1?)";
                return false;
            }
            const uint32_t wi = j >> 5, bit = 1u << (j & 31);
            if (msb) blk->w[wi] |= bit;
            if (lsb) blk->w[T + wi] |= bit;
            for (uint32_t q = 0; q < P; ++q) {
                if ((size >> q) & 1) blk->w[2 * T + q * T + wi] |= bit;
                if ((pos >> q) & 1) blk->w[2 * T + P * T + q * T + wi] |= bit;
            }
        }
        if (tailf.t != mm_free.size()) {
            err = "minimizer_order free slots point beyond the minimizers";
            return false;
        }
        if (i_l != n_left || i_r != n_right || i_n != n_none || n_max_seen != x.n_maximal) {
            err = "wavelet tree census does not match the region starts";
            return false;
        }
        h.kmers_by_type[0] = vals[rcs];
        h.kmers_by_type[1] = vals[nss] - vals[rcs];
        h.kmers_by_type[2] = base;
        h.kmers_by_type[3] = vals[nps] - vals[nss];
        // consistency: every k-mer is counted exactly once (maximal + EF sizes + fallback)
        if (base + vals[nps] + x.fallback_kmer_order.num_keys != x.nkmers) {
            err = "k-mer census mismatch: W*n_maximal + sum(sizes) + fallback != nkmers";
            return false;
        }
    } else {
        std::vector<uint64_t> sz, ps;
        if (!x.sizes.decode(sz, err) || !x.positions.decode(ps, err)) return false;
        if (sz.empty()) sz.push_back(0);
        if (ps.empty()) ps.push_back(0);
        if (sz.size() != D + 1 || ps.size() != D + 1) {
            err = "positions/sizes: entry count != distinct_minimizers + 1";
            return false;
        }
        if (sz[D] != x.num_kmers_in_main_index || sz[D] + x.fallback_kmer_order.num_keys != x.nkmers) {
            err = "k-mer census mismatch: sum(sizes) != num_kmers_in_main_index or + fallback != nkmers";
            return false;
        }
        h.collision_base = x.num_kmers_in_main_index;  // unpartitioned_mphf.cpp:198
        h.off_upair = w.reserve(((D + LPHQ_PAIR_PER_BLOCK - 1) / LPHQ_PAIR_PER_BLOCK + 1) * sizeof(lphq_plane_block));
        if (!fill_pair_blocks(w, h.off_upair, sz.data(), ps.data(), D, 0, h.P, err)) return false;
        // tail entries: what ctx_unpartitioned() makes of a minimizer (src/unpartitioned_mphf.cpp:194-204)
        h.off_tail = w.reserve((mm_free.size() ? mm_free.size() : 1) * 8);
        TailFiller tailf{mm_free, w.at<uint64_t>(h.off_tail)};
        for (uint64_t i = 0; i < D; ++i) {
            const uint64_t size = sz[i + 1] - sz[i], pos = ps[i + 1] - ps[i];
            if (!(size == 0 ? tailf.put(i, h.collision_base, 2, err) : tailf.put(i, sz[i] + pos, 1, err))) return false;
        }
        if (tailf.t != mm_free.size()) {
            err = "minimizer_order free slots point beyond the minimizers";
            return false;
        }
    }
    img.resize((img.size() + 255) & ~(size_t)255, 0);
    h.total_bytes = img.size();
    memcpy(img.data(), &h, sizeof h);
    return true;
}

}  // namespace lphq
