// lph_format.hpp -- host-side model of the serialized LP-MPHF (`.lph`) and its parser.
//
// The byte layout is the reference's essentials::save/load walk over visit():
//   mphf       include/partitioned_mphf.hpp:204-219     mphf_alt  include/unpartitioned_mphf.hpp:199-211
//   quartet_wtree include/quartet_wtree.hpp:43-48       rs_bit_vector include/rs_bit_vector.hpp:91-96
//   ef_sequence include/ef_sequence.hpp:107-112         PTHash single_phf & encoders: SURVEY.md Appendix A
// (paths relative to the reference tree).  POD = raw bytes, std::vector<T> = u64 n + n*sizeof(T).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace lphq {

typedef unsigned __int128 u128;

struct BitVector {  // pthash::bit_vector
    uint64_t size = 0;
    std::vector<uint64_t> bits;
    bool get(uint64_t i) const { return (bits[i >> 6] >> (i & 63)) & 1; }
};
struct CompactVector {  // pthash::compact_vector
    uint64_t size = 0, width = 0, mask = 0;
    std::vector<uint64_t> bits;
    uint64_t access(uint64_t i) const {
        if (!width) return 0;
        uint64_t pos = i * width, blk = pos >> 6, sh = pos & 63;
        uint64_t v = bits[blk] >> sh;
        if (sh + width > 64) v |= bits[blk + 1] << (64 - sh);
        return v & mask;
    }
};
struct DArray1 {  // pthash::darray1 (only carried through; the device image does not use it)
    uint64_t positions = 0;
    std::vector<int64_t> block_inventory;
    std::vector<uint16_t> subblock_inventory;
    std::vector<uint64_t> overflow_positions;
};
struct EliasFano {  // pthash::ef_sequence<false> and lphash::ef_sequence share the serialized shape
    BitVector high;
    DArray1 d1;
    CompactVector low;
    uint64_t size() const { return low.size; }
    // Decode all stored values by one sequential pass over the high bits.
    bool decode(std::vector<uint64_t>& out, std::string& err) const;
};
struct RsBitVector {  // lphash::rs_bit_vector
    BitVector bv;
    std::vector<uint64_t> block_rank_pairs, select_hints;
};
struct Dictionary { CompactVector ranks, dict; };
struct SinglePhf {  // pthash::single_phf<Hasher, dictionary_dictionary, true>
    uint64_t seed = 0, num_keys = 0, table_size = 0;
    u128 M = 0;
    uint64_t num_dense = 0, num_sparse = 0;
    u128 M_dense = 0, M_sparse = 0;
    Dictionary front, back;
    EliasFano free_slots;
};

struct LphFile {
    int kind = 0;  // LPHQ_PARTITIONED / LPHQ_UNPARTITIONED / LPHQ_PTHASH
    int kmer_bits = 64;
    uint8_t k = 0, m = 0;
    uint64_t mm_seed = 0, nkmers = 0, distinct_minimizers = 0;
    // partitioned
    uint64_t n_maximal = 0, right_coll_sizes_start = 0, none_sizes_start = 0, none_pos_start = 0;
    RsBitVector root, left_right, max_none;
    EliasFano sizes_and_positions;
    // unpartitioned
    uint64_t num_kmers_in_main_index = 0;
    EliasFano positions, sizes;
    SinglePhf minimizer_order, fallback_kmer_order;
    uint64_t bytes_read = 0;
    // serialized size of the parts print_statistics() reports (src/partitioned_mphf.cpp:341-380): minimizer MPHF, wavelet tree,
    // Elias-Fano arrays (with their four region scalars; both sequences for the unpartitioned function), fallback MPHF
    uint64_t part_bytes[4] = {0, 0, 0, 0};
};

// A bare single_phf file (the reference's ptbb harness, src/ptbb/ptbb_build.cpp:99): kind = 2, k from the caller, m = k; the
// function goes to minimizer_order, nkmers = distinct_minimizers = its number of keys.
bool parse_pthash(const uint8_t* buf, size_t len, int k, int kmer_bits, LphFile& out, std::string& err);

// Parse + cross-check redundant fields.  Returns false and fills err on failure.
bool parse_lph(const uint8_t* buf, size_t len, int kind, int kmer_bits, LphFile& out, std::string& err);

u128 compute_M_u64(uint64_t d);
uint64_t murmur64a_u64(uint64_t v, uint64_t seed);  // MurmurHash2_64 of the 8 bytes of v

}  // namespace lphq
