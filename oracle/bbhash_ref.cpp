/* bbhash_ref.cpp -- TEST INFRASTRUCTURE ONLY.  The BBHash half of the reference's competitor harness (src/ptbb/ptbb_build.cpp:140-199,
 * src/ptbb/ptbb_query.cpp:56-72) over the REAL BooPHF: this driver includes the reference's own external/BooPHF.hpp and external/kseq.h from
 * /root/reference, where they lie -- nothing of them is copied into this repo -- and is compiled by oracle/Makefile into oracle/_ref/bbhash_ref
 * only where /root/reference exists.  What is restated here is the 8 lines around the library: ptbb::BBHasher (src/ptbb/ptbb.hpp:25-32 --
 * MurmurHash2_64 of the k-mer's bytes with seed and ~seed, combined by pthash::hash128::mix(), which is first ^ second in PTHash v1; PTHash is
 * absent from the reference tree) and the k-mer iteration of ptbb_file_itr (src/ptbb/ptbb.hpp:108-131).
 *
 *   bbhash_ref build <fasta[.gz]> <k> <kmer_bits> <gamma> <out.bbh>                 build + save, exactly as ptbb_build does (writeEach = true, 1 thread)
 *   bbhash_ref query <in.bbh> <k> <kmer_bits> <fasta[.gz]> <codes.u64>              load + lookup of every window of k valid bases -> raw u64 codes
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <zlib.h>

#include <fstream>
#include <string>
#include <vector>

extern "C" {
#include "kseq.h"
}
KSEQ_INIT(gzFile, gzread)
#include "BooPHF.hpp"

static uint64_t murmur64a(const void* key, size_t len, uint64_t seed) {  // public algorithm (MurmurHash64A)
    const uint64_t m = 0xc6a4a7935bd1e995ULL;
    const int r = 47;
    uint64_t h = seed ^ (len * m);
    const unsigned char* d = (const unsigned char*)key;
    for (size_t i = 0; i + 8 <= len; i += 8) {
        uint64_t k;
        memcpy(&k, d + i, 8);
        k *= m; k ^= k >> r; k *= m;
        h ^= k; h *= m;
    }
    h ^= h >> r; h *= m; h ^= h >> r;  // (len is 8 or 16 here: no tail bytes)
    return h;
}
template <typename kmer_t>
struct BBHasher {  // src/ptbb/ptbb.hpp:25-32
    uint64_t operator()(kmer_t val, uint64_t seed = 1234567890) const { return murmur64a(&val, sizeof val, seed) ^ murmur64a(&val, sizeof val, ~seed); }
};
static int nt4(unsigned char c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': case 'U': case 'u': return 3; default: return 4; }
}
template <typename kmer_t>
static bool kmers_of(const char* path, unsigned k, std::vector<kmer_t>& out) {  // ptbb_file_itr: every window of k valid bases, forward strand
    gzFile fp = gzopen(path, "r");
    if (!fp) return false;
    kseq_t* seq = kseq_init(fp);
    const kmer_t mask = (((kmer_t)1) << (2 * k)) - 1;
    while (kseq_read(seq) >= 0) {
        kmer_t km = 0;
        uint64_t since = 0;
        for (size_t i = 0; i < seq->seq.l; ++i) {
            const int c = nt4((unsigned char)seq->seq.s[i]);
            if (c < 4) {
                km = ((km << 2) | (kmer_t)c) & mask;
                if (++since >= k) out.push_back(km);
            } else
                since = 0;
        }
    }
    kseq_destroy(seq);
    gzclose(fp);
    return true;
}
template <typename kmer_t>
static int run(int argc, char** argv) {
    typedef boomphf::mphf<kmer_t, BBHasher<kmer_t>> mphf_t;
    const std::string cmd = argv[1];
    if (cmd == "build") {
        const unsigned k = (unsigned)atoi(argv[3]);
        const double gamma = atof(argv[5]);
        std::vector<kmer_t> keys;
        if (!kmers_of<kmer_t>(argv[2], k, keys)) return 2;
        auto it = boomphf::range(keys.begin(), keys.end());
        mphf_t f(keys.size(), it, 1, gamma, true, false, 0);  // ptbb_build.cpp:158-159
        std::ofstream os(argv[6], std::ios::binary);
        f.save(os);
        printf("%zu\n", keys.size());
        return 0;
    }
    const unsigned k = (unsigned)atoi(argv[3]);
    mphf_t f;
    std::ifstream is(argv[2], std::ios::binary);
    if (!is) return 2;
    f.load(is);
    std::vector<kmer_t> keys;
    if (!kmers_of<kmer_t>(argv[5], k, keys)) return 2;
    std::vector<uint64_t> codes(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) codes[i] = f.lookup(keys[i]);  // ptbb_query.cpp:64
    FILE* fo = fopen(argv[6], "wb");
    if (!fo) return 2;
    fwrite(codes.data(), 8, codes.size(), fo);
    fclose(fo);
    printf("%zu\n", codes.size());
    return 0;
}
int main(int argc, char** argv) {
    if (argc != 7) {
        fprintf(stderr, "usage: %s build <fasta> <k> <kmer_bits> <gamma> <out.bbh> | query <in.bbh> <k> <kmer_bits> <fasta> <codes.u64>\n", argv[0]);
        return 1;
    }
    return atoi(argv[4]) == 64 ? run<uint64_t>(argc, argv) : run<__uint128_t>(argc, argv);
}
