// dump_codes.cpp -- writes the hash codes the REFERENCE computes, so that they can be compared here (tests/golden/ref/README.md).
// Compile INSIDE a recursive clone of jermp/lphash (it includes the reference's own headers; nothing of the reference is copied here):
//   g++ -std=c++17 -O3 -pthread -I. <this file> src/constants.cpp src/quartet_wtree.cpp src/minimizer.cpp src/partitioned_mphf.cpp
//       src/unpartitioned_mphf.cpp src/mphf_utils.cpp -lz -o dump_codes
// usage: dump_codes p|u index.lph queries.fa[.gz] codes_streaming.u64 codes_random.u64
// The loop is the reference driver's (src/query.cpp:48-79): one hf(seq, len, streaming) call per record, streaming pass then random pass;
// instead of discarding the vectors (query.cpp:54) their values are written as little-endian uint64, records concatenated.
extern "C" {
#include "external/kseq.h"
}
#include <zlib.h>

#include <cstdio>
#include <string>
#include <vector>

#include "external/pthash/external/essentials/include/essentials.hpp"
#include "include/partitioned_mphf.hpp"
#include "include/unpartitioned_mphf.hpp"

KSEQ_INIT(gzFile, gzread)

template <typename MPHF>
static int run(const char* index, const char* queries, const char* out_streaming, const char* out_random) {
    MPHF hf;
    essentials::load(hf, index);
    const char* outs[2] = {out_streaming, out_random};
    for (int pass = 0; pass < 2; ++pass) {
        gzFile fp = gzopen(queries, "r");
        if (!fp) return 2;
        FILE* f = fopen(outs[pass], "wb");
        if (!f) return 2;
        kseq_t* seq = kseq_init(fp);
        while (kseq_read(seq) >= 0) {
            std::vector<uint64_t> codes = hf(seq->seq.s, seq->seq.l, pass == 0);
            if (!codes.empty()) fwrite(codes.data(), sizeof(uint64_t), codes.size(), f);
        }
        kseq_destroy(seq);
        gzclose(fp);
        fclose(f);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 6) {
        fprintf(stderr, "usage: %s p|u index.lph queries.fa[.gz] codes_streaming.u64 codes_random.u64\n", argv[0]);
        return 1;
    }
    return std::string(argv[1]) == "p" ? run<lphash::mphf>(argv[2], argv[3], argv[4], argv[5])
                                       : run<lphash::mphf_alt>(argv[2], argv[3], argv[4], argv[5]);
}
