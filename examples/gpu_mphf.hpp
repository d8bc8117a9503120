// gpu_mphf.hpp -- the reference-side binding of INTEGRATION.md section 2, as compiled code: a drop-in for lphash::mphf /
// lphash::mphf_alt in the reference's src/query.cpp and src/build.cpp that forwards to the C ABI (include/lphq.h).
//
// Inside the reference tree a maintainer adds this header, links -llphq (CMakeLists.txt:56) and changes two lines of src/lphash.cpp:28-31:
//     return query<gpu_mphf<LPHQ_PARTITIONED, kmer_t>>(argc - 1, argv + 1);      // was query<mphf>
//     return query<gpu_mphf<LPHQ_UNPARTITIONED, kmer_t>>(argc - 1, argv + 1);    // was query<mphf_alt>
// and, in query<MPHF>() (src/query.cpp:37), essentials::load(hf, name) -> hf.load(name).  examples/query_main.cpp is that driver,
// restated, so that the binding is built and run by this repository's tests (tests/test_zz_gpu_next.py::test_integration_stub_*).
//
// Two ways to ask: operator()(contig, length, streaming) is the reference's call, one record at a time (copies in and out and a wait per record:
// faithful, and ~200x below what the device does); submit() / collect() take MANY records per call and keep batches in flight
// (lphq_session_*): that is the one a driver should loop over -- examples/query_main.cpp does, and keeps the literal loop as --per-record.
//
// Only lphq.h and the standard library are needed: plain pointers and sizes cross the boundary, no exception does.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "lphq.h"

namespace lphash {

template <int KIND /* LPHQ_PARTITIONED | LPHQ_UNPARTITIONED */, typename KmerT /* the reference's kmer_t, include/compile_constants.tpd:1-2 */>
class gpu_mphf {
public:
    gpu_mphf() = default;
    gpu_mphf(const gpu_mphf&) = delete;
    gpu_mphf& operator=(const gpu_mphf&) = delete;
    ~gpu_mphf() {
        lphq_session_close(s_);
        lphq_close(h_);
    }

    // replaces essentials::load(hf, path)                                   (src/query.cpp:37)
    void load(const char* path, int device = 0) { load(path, std::vector<int>{device}); }
    // ... with a replica of the function on every GPU named: submit() / collect() then spread their batches over them in turn (one immutable `MPHF hf`
    // shared by all callers, include/partitioned_mphf.hpp:21-23); operator() below keeps to the first
    void load(const char* path, const std::vector<int>& devices) {
        lphq_session_close(s_);
        s_ = nullptr;
        lphq_close(h_);
        h_ = nullptr;
        devices_ = devices.empty() ? std::vector<int>{0} : devices;
        if (lphq_open(path, KIND, (int)sizeof(KmerT) * 8, &h_) != LPHQ_OK || lphq_attach_devices(h_, devices_.data(), (int)devices_.size()) != LPHQ_OK)
            throw std::runtime_error(lphq_last_error());
    }

    // replaces MPHF::operator()(contig, length, streaming)                   (src/query.cpp:52,72; include/partitioned_mphf.hpp:21-23)
    std::vector<uint64_t> operator()(const char* contig, std::size_t length, bool streaming = true) const {
        std::vector<uint64_t> res(length ? length : 1);  // >= length - k + 1
        std::size_t n = 0;
        if (lphq_query_contig(h_, contig, length, streaming ? 1 : 0, res.data(), res.size(), &n) != LPHQ_OK) throw std::runtime_error(lphq_last_error());
        res.resize(n);
        return res;
    }

    // ---- many records per call ----
    // The codes of one batch: record r's codes are [begin(r), begin(r) + size(r)), in the session's pinned memory, valid until the next submit().
    class batch {
    public:
        std::size_t records() const { return n_; }
        const uint64_t* begin(std::size_t r) const { return codes_ + offs_[r]; }
        std::size_t size(std::size_t r) const { return (std::size_t)(offs_[r + 1] - offs_[r]); }
        std::size_t total() const { return n_ ? (std::size_t)offs_[n_] : 0; }
        std::vector<uint64_t> operator[](std::size_t r) const { return std::vector<uint64_t>(begin(r), begin(r) + size(r)); }  // (a copy: what operator() returns)

    private:
        friend class gpu_mphf;
        const uint64_t *codes_ = nullptr, *offs_ = nullptr;
        std::size_t n_ = 0;
    };
    // The streaming result of one batch as the reference COMPUTES it (include/partitioned_mphf.hpp:131-145): one lphq_run per stretch of k-mers whose codes
    // go up or down by one -- what res.push_back(mm_ctx.hval) spells out k-mer by k-mer.  windows() is the sum of the vector sizes a loop over operator() would
    // have seen (src/query.cpp:53); a caller that indexes per-k-mer data reads table[code .. code + length) per run.  Valid until the next submit().
    class run_batch {
    public:
        std::size_t records() const { return n_; }
        std::size_t runs() const { return n_runs_; }
        const lphq_run& operator[](std::size_t i) const { return runs_[i]; }
        uint64_t windows() const { return windows_; }
        const uint64_t* explicit_codes() const { return codes_; }
        uint64_t record_begin(std::size_t r) const { return offs_[r]; }  // where record r's windows begin in the space LPHQ_RUN_FIRST counts in
        // code of window i (< LPHQ_RUN_LENGTH) of run r
        uint64_t code(const lphq_run& r, uint32_t i) const {
            const uint32_t kind = LPHQ_RUN_KIND(r);
            return kind == LPHQ_RUN_UP ? r.code + i : kind == LPHQ_RUN_DOWN ? r.code - i : codes_[r.code + i];
        }

    private:
        friend class gpu_mphf;
        const lphq_run* runs_ = nullptr;
        const uint64_t *codes_ = nullptr, *offs_ = nullptr;
        std::size_t n_runs_ = 0, n_codes_ = 0, n_ = 0;
        uint64_t windows_ = 0;
    };
    // Set-up before a timer starts: the session and its buffers for batches of up to total_bases bases in n_records records (optional).
    void reserve(std::size_t total_bases, std::size_t n_records) {
        open_session();
        if (lphq_session_reserve(s_, total_bases, n_records) != LPHQ_OK) throw std::runtime_error(lphq_last_error());
    }
    // Queue one batch (the records' memory must stay valid until the batch is collected; depth() batches may be in flight per GPU).
    void submit(const char* const* records, const std::size_t* lengths, std::size_t n, bool streaming = true) {
        open_session();
        if (lphq_session_submit(s_, records, lengths, n, streaming ? 1 : 0) != LPHQ_OK) throw std::runtime_error(lphq_last_error());
    }
    // ... to be collected as runs (streaming pass only)
    void submit_runs(const char* const* records, const std::size_t* lengths, std::size_t n) {
        open_session();
        if (lphq_session_submit_runs(s_, records, lengths, n) != LPHQ_OK) throw std::runtime_error(lphq_last_error());
    }
    run_batch collect_runs() {
        run_batch b;
        if (lphq_session_collect_runs(s_, &b.runs_, &b.n_runs_, &b.codes_, &b.n_codes_, &b.windows_, &b.offs_, &b.n_) != LPHQ_OK) throw std::runtime_error(lphq_last_error());
        return b;
    }
    int in_flight_capacity() const { return kDepth * (int)devices_.size(); }
    void submit(const std::vector<std::string_view>& records, bool streaming = true) {
        ptrs_.resize(records.size());
        lens_.resize(records.size());
        for (std::size_t i = 0; i < records.size(); ++i) {
            ptrs_[i] = records[i].data();
            lens_[i] = records[i].size();
        }
        submit(ptrs_.data(), lens_.data(), records.size(), streaming);
    }
    // The oldest batch in flight.
    batch collect() {
        batch b;
        if (lphq_session_collect(s_, &b.codes_, &b.offs_, &b.n_) != LPHQ_OK) throw std::runtime_error(lphq_last_error());
        return b;
    }
    static constexpr int depth() { return kDepth; }
    // MPHF::operator() over a whole batch: one vector per record, like a loop over operator() (and one copy more than submit / collect)
    std::vector<std::vector<uint64_t>> operator()(const std::vector<std::string_view>& records, bool streaming = true) {
        submit(records, streaming);
        const batch b = collect();
        std::vector<std::vector<uint64_t>> res(b.records());
        for (std::size_t r = 0; r < b.records(); ++r) res[r] = b[r];
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
    void open_session() {
        if (s_) return;
        const int rc = devices_.size() > 1 ? lphq_session_open_devices(h_, kDepth, devices_.data(), (int)devices_.size(), &s_) : lphq_session_open(h_, kDepth, &s_);
        if (rc != LPHQ_OK) throw std::runtime_error(lphq_last_error());
    }
    static constexpr int kDepth = 2;
    std::vector<int> devices_{0};
    lphq_index* h_ = nullptr;
    lphq_session* s_ = nullptr;
    std::vector<const char*> ptrs_;
    std::vector<std::size_t> lens_;
};

}  // namespace lphash
