// gpu_mphf.hpp -- the reference-side binding of INTEGRATION.md section 2, as compiled code: a drop-in for lphash::mphf /
// lphash::mphf_alt in the reference's src/query.cpp and src/build.cpp that forwards to the C ABI (include/lphq.h).
//
// Inside the reference tree a maintainer adds this header, links -llphq (CMakeLists.txt:56) and changes two lines of src/lphash.cpp:28-31:
//     return query<gpu_mphf<LPHQ_PARTITIONED>>(argc - 1, argv + 1);      // was query<mphf>
//     return query<gpu_mphf<LPHQ_UNPARTITIONED>>(argc - 1, argv + 1);    // was query<mphf_alt>
// and, in query<MPHF>() (src/query.cpp:37), essentials::load(hf, name) -> hf.load(name).  examples/query_main.cpp is that driver,
// restated, so that the binding is built and run by this repository's tests (tests/test_zz_gpu_next.py::test_integration_stub_*).
//
// Only lphq.h and the standard library are needed: plain pointers and sizes cross the boundary, no exception does.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "lphq.h"

namespace lphash {

template <int KIND /* LPHQ_PARTITIONED | LPHQ_UNPARTITIONED */, typename KmerT /* the reference's kmer_t, include/compile_constants.tpd:1-2 */>
class gpu_mphf {
public:
    gpu_mphf() = default;
    gpu_mphf(const gpu_mphf&) = delete;
    gpu_mphf& operator=(const gpu_mphf&) = delete;
    ~gpu_mphf() { lphq_close(h_); }

    // replaces essentials::load(hf, path)                                   (src/query.cpp:37)
    void load(const char* path, int device = 0) {
        lphq_close(h_);
        h_ = nullptr;
        if (lphq_open(path, KIND, (int)sizeof(KmerT) * 8, &h_) != LPHQ_OK || lphq_attach(h_, device) != LPHQ_OK) throw std::runtime_error(lphq_last_error());
        // several GPUs in one process: lphq_attach_devices(h_, devs, n); a calling thread then selects its GPU with hipSetDevice
    }

    // replaces MPHF::operator()(contig, length, streaming)                   (src/query.cpp:52,72; include/partitioned_mphf.hpp:21-23)
    std::vector<uint64_t> operator()(const char* contig, std::size_t length, bool streaming = true) const {
        std::vector<uint64_t> res(length ? length : 1);  // >= length - k + 1
        std::size_t n = 0;
        if (lphq_query_contig(h_, contig, length, streaming ? 1 : 0, res.data(), res.size(), &n) != LPHQ_OK) throw std::runtime_error(lphq_last_error());
        res.resize(n);
        return res;
    }

    uint64_t get_kmer_count() const { return info().num_kmers; }              // src/partitioned_mphf.cpp:272
    uint64_t get_minimizer_L0() const { return info().distinct_minimizers; }  // src/partitioned_mphf.cpp:270

    // replaces MPHF::build(config, res) + essentials::save(hf, output_filename)   (src/build.cpp:24-28): `next` hands the library batches of whole
    // records, twice (the reference reads its input twice, src/partitioned_mphf.cpp:66-78 and :118-130)
    static void build(lphq_record_source next, void* ctx, int k, int m, uint64_t mm_seed, double c, int threads, int device, const char* tmp_dir,
                      uint64_t max_memory_bytes, const char* out_path, lphq_build_stats* stats = nullptr) {
        if (lphq_build_stream(next, ctx, k, m, mm_seed, c, KIND, (int)sizeof(KmerT) * 8, threads, device, tmp_dir, max_memory_bytes, out_path, stats) != LPHQ_OK)
            throw std::runtime_error(lphq_last_error());
    }

private:
    lphq_info info() const {
        lphq_info i;
        if (lphq_get_info(h_, &i) != LPHQ_OK) throw std::runtime_error(lphq_last_error());
        return i;
    }
    lphq_index* h_ = nullptr;
};

}  // namespace lphash
