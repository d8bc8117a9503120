// bbhash.hpp -- the BBHash half of the reference's competitor harness (src/ptbb/ptbb_build.cpp:140-199, src/ptbb/ptbb_query.cpp:56-72):
// reader of the file boomphf::mphf<kmer_t, ptbb::BBHasher>::save writes (external/BooPHF.hpp:948-965, bitVector::save :628-635), its flat
// device image (image_layout.h: kind 3), and a builder that writes a file the reference's load() accepts.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "image_layout.h"

namespace lphq {

struct BbhFile {
    int k = 0, kmer_bits = 64;
    double gamma = 0;
    int32_t nb_levels = 0;
    uint64_t last_bitset_rank = 0, nelem = 0;
    struct Level {
        uint64_t size = 0;             // bits (= hash_domain of the level)
        std::vector<uint64_t> words;   // 1 + size / 64 words
        std::vector<uint64_t> ranks;   // ones before every 512-bit block (offset by the levels before)
        uint64_t hash_domain = 0;      // recomputed as load() does (BooPHF.hpp:979-989)
    };
    std::vector<Level> levels;
    std::vector<uint64_t> final_keys;  // kmer_bits / 64 words per key (low word first), sorted ascending by (high, low)
    std::vector<uint64_t> final_vals;
    uint64_t bytes_read = 0;
};

bool parse_bbhash(const uint8_t* buf, size_t len, int k, int kmer_bits, BbhFile& out, std::string& err);
bool build_bbhash_image(const BbhFile& f, std::vector<uint8_t>& image, std::string& err);
// BBHash over every window of k valid bases of the records (ptbb_file_itr, src/ptbb/ptbb.hpp:108-131), gamma >= 1, 25 levels
bool build_bbhash_file(const char* const* records, const size_t* lengths, size_t n_records, int k, int kmer_bits, double gamma, int threads,
                       std::vector<uint8_t>& file_bytes, uint64_t& nkeys, std::string& err);

}  // namespace lphq
