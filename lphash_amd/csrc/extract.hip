// extract.hip -- index construction, parts 1-2 on the GPU ("minimizer extraction + radix sort of
// mm_record_t on GPU"): the super-k-mer records of a batch of records (k_extract_runs, extract_device.cuh), sorted by minimizer
// (rocPRIM radix sort: a plain library sort of (key, value) pairs; the key is the 2m-bit minimizer) and, when the whole input is one batch, classified
// there as well (src/minimizer.cpp:5-50: a minimizer seen once keeps its offset and size, one seen more often sends its super-k-mers to the fallback),
// so that only the distinct minimizers travel back.
// Reference: minimizer::from_string (include/minimizer.hpp:11-170) and the sorted external_memory_vector<mm_record_t> it fills
// (src/partitioned_mphf.cpp:62-65).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "extract.hpp"
#include "extract_device.cuh"

namespace lphq {
namespace {

constexpr size_t kStage = 64u << 20;  // bytes per pinned staging buffer (two of them)
constexpr int kFront = 64;            // spare bytes before the first base and after the last one

// bytes [from, from + n) of the concatenated records into dst, by `threads` threads
void gather_bytes(const char* const* records, const size_t* lengths, const std::vector<uint64_t>& off, uint64_t from, size_t n, char* dst, int threads) {
    if (!n) return;
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n / (1u << 20) + 1));
    auto piece = [&](int t) {
        const uint64_t a = from + (uint64_t)n * t / T, b = from + (uint64_t)n * (t + 1) / T;
        size_t r = (size_t)(std::upper_bound(off.begin(), off.end(), a) - off.begin()) - 1;  // record of byte a
        uint64_t at = a;
        while (at < b) {
            while (off[r + 1] <= at) ++r;
            const uint64_t e = std::min<uint64_t>(b, off[r + 1]);
            memcpy(dst + (at - from), records[r] + (at - off[r]), (size_t)(e - at));
            at = e;
        }
    };
    if (T == 1) {
        piece(0);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(piece, t);
    for (auto& x : th) x.join();
}

// n bytes from a pinned buffer into pageable memory, by `threads` threads
void scatter_bytes(const char* src, char* dst, size_t n, int threads) {
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n / (1u << 20) + 1));
    if (T == 1) {
        memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] { memcpy(dst + n * t / T, src + n * t / T, n * (t + 1) / T - n * t / T); });
    for (auto& x : th) x.join();
}

// Everything the device path owns, released in one place.  step() = a HIP call checked; after a failure ok is false and err says what.
struct Extractor {
    std::string& err;
    bool verbose;
    int threads;
    bool ok = true;
    int prev = -1;
    uint8_t* d_seq = nullptr;
    uint64_t *d_off = nullptr, *d_mm[2] = {nullptr, nullptr};
    RunVal* d_val[2] = {nullptr, nullptr};
    unsigned long long* d_cursor = nullptr;
    void* d_tmp = nullptr;
    char* h_stage[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    unsigned long long produced = 0;  // super-k-mers of the batch; sorted by minimizer in d_mm[1] / d_val[1] after extract_and_sort()
    unsigned long long lost = 0;      // valid windows in no super-k-mer (the reference leaves them out, builder.cpp: lost_window); part of the k-mer count
    std::chrono::steady_clock::time_point tprev = std::chrono::steady_clock::now();

    Extractor(std::string& e, bool v, int t) : err(e), verbose(v), threads(t) { (void)hipGetDevice(&prev); }
    ~Extractor() {
        if (stream) (void)hipStreamSynchronize(stream);
        for (int i = 0; i < 2; ++i) {
            if (d_mm[i]) (void)hipFree(d_mm[i]);
            if (d_val[i]) (void)hipFree(d_val[i]);
            if (h_stage[i]) (void)hipHostFree(h_stage[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
        if (d_tmp) (void)hipFree(d_tmp);
        if (d_seq) (void)hipFree(d_seq);
        if (d_off) (void)hipFree(d_off);
        if (d_cursor) (void)hipFree(d_cursor);
        if (stream) (void)hipStreamDestroy(stream);
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    void lap(const char* what) {
        auto now = std::chrono::steady_clock::now();
        if (verbose) fprintf(stderr, "[lphq_build]   gpu extraction: %-30s %8.3f s\n", what, std::chrono::duration<double>(now - tprev).count());
        tprev = now;
    }
    bool step(hipError_t e, const char* what) {
        if (ok && e != hipSuccess) {
            err = std::string(what) + ": " + hipGetErrorString(e);
            ok = false;
        }
        return ok;
    }
#define STEP(call)                \
    do {                          \
        if (!step((call), #call)) return false; \
    } while (0)

    void release(void*& p) {
        if (p) (void)hipFree(p);
        p = nullptr;
    }

    // bases up, k_extract_runs (a second launch with the exact room if the first buffer was too small), radix sort by minimizer.
    // true with produced == 0: the batch has no window.
    bool extract_and_sort(int device, const char* const* records, const size_t* lengths, size_t n_records, uint64_t first_base, uint32_t k, uint32_t m, uint64_t mm_seed) {
        std::vector<uint64_t> off(n_records + 1, 0);
        for (size_t r = 0; r < n_records; ++r) off[r + 1] = off[r] + lengths[r];
        const uint64_t total = off[n_records];
        const uint32_t W = k - m + 1u;
        if (n_records == 0 || total < k) return true;
        if (n_records > 0xFFFFFFFFull) {
            err = "gpu extraction: more than 2^32 - 1 records in one batch";
            return ok = false;
        }
        uint64_t windows = 0;
        for (size_t r = 0; r < n_records; ++r)
            if (lengths[r] >= k) windows += lengths[r] - k + 1;
        if (windows == 0) return true;
        STEP(hipSetDevice(device));
        STEP(hipStreamCreate(&stream));
        STEP(hipMalloc((void**)&d_seq, (size_t)total + 2 * kFront + 16));
        STEP(hipMalloc((void**)&d_off, (n_records + 1) * sizeof(uint64_t)));
        STEP(hipMalloc((void**)&d_cursor, 3 * sizeof(unsigned long long)));  // [0] runs produced, [1] k-mers of the runs (classification), [2] lost windows (extract_device.cuh)
        STEP(hipMemsetAsync(d_seq, 0, kFront, stream));
        STEP(hipMemsetAsync(d_seq + kFront + total, 0, kFront + 16, stream));
        STEP(hipMemcpyAsync(d_off, off.data(), (n_records + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        // the bases: the concatenated records through two pinned staging buffers (records may be many and small, or few and huge)
        for (int i = 0; i < 2; ++i) {
            STEP(hipHostMalloc((void**)&h_stage[i], kStage, hipHostMallocDefault));
            STEP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        lap("device and pinned buffers");
        {
            int which = 0;
            for (uint64_t at = 0; at < total; at += kStage, which ^= 1) {
                const size_t n = (size_t)std::min<uint64_t>(kStage, total - at);
                STEP(hipEventSynchronize(ev[which]));  // the copy that last used this buffer (a fresh event is complete)
                gather_bytes(records, lengths, off, at, n, h_stage[which], threads);
                STEP(hipMemcpyAsync(d_seq + kFront + at, h_stage[which], n, hipMemcpyHostToDevice, stream));
                STEP(hipEventRecord(ev[which], stream));
            }
        }
        lap("bases to the device");
        // expected number of runs on random sequence: 2 / (W + 1) per window; room for 1.5x that (repetitive input has FEWER runs), exact on a second go
        uint64_t cap = (uint64_t)((double)windows * 3.0 / (double)(W + 1)) + (1u << 16);
        cap = std::min(cap, windows);
        if (const char* e = getenv("LPHQ_EXTRACT_INITIAL_CAP")) {  // (tests: a too-small first buffer exercises the second launch with the exact count)
            const uint64_t v = strtoull(e, nullptr, 10);
            if (v >= 1) cap = std::min(cap, v);
        }
        const uint64_t shift = 16;  // d_seq + kFront is 16-byte aligned (hipMalloc: 256), the kernel reads from d_seq + kFront - 16 on
        const uint8_t* seq_al = d_seq + kFront - shift;
        const uint32_t S = extract_tile_stride(W, 4 * kStreamBlockDefault);
        const uint64_t tiles = (shift + total - k) / S + 1;  // the last window start has shifted coordinate shift + total - k >= 1
        for (int attempt = 0; attempt < 2; ++attempt) {
            for (int i = 0; i < 2; ++i) {
                release((void*&)d_mm[i]);
                release((void*&)d_val[i]);
            }
            for (int i = 0; i < 2; ++i) {
                STEP(hipMalloc((void**)&d_mm[i], (size_t)cap * sizeof(uint64_t)));
                STEP(hipMalloc((void**)&d_val[i], (size_t)cap * sizeof(RunVal)));
            }
            STEP(hipMemsetAsync(d_cursor, 0, 3 * sizeof(unsigned long long), stream));
            const uint64_t kMaxTiles = (1u << 24) - 1;
            for (uint64_t tb = 0; tb < tiles; tb += kMaxTiles) {
                const unsigned grid = (unsigned)std::min<uint64_t>(kMaxTiles, tiles - tb);
                if (W >= kBlockMinW)
                    k_extract_runs<kStreamBlockDefault, true><<<grid, kStreamBlockDefault, 0, stream>>>(seq_al, shift, total, d_off, (uint32_t)n_records, k, m, mm_seed,
                                                                                                         first_base, d_mm[0], d_val[0], d_cursor, cap, tb, d_cursor + 2);
                else
                    k_extract_runs<kStreamBlockDefault, false><<<grid, kStreamBlockDefault, 0, stream>>>(seq_al, shift, total, d_off, (uint32_t)n_records, k, m, mm_seed,
                                                                                                          first_base, d_mm[0], d_val[0], d_cursor, cap, tb, d_cursor + 2);
                STEP(hipGetLastError());
            }
            STEP(hipMemcpyAsync(&produced, d_cursor, sizeof produced, hipMemcpyDeviceToHost, stream));
            STEP(hipMemcpyAsync(&lost, d_cursor + 2, sizeof lost, hipMemcpyDeviceToHost, stream));
            STEP(hipStreamSynchronize(stream));
            if (produced <= cap) break;
            if (attempt == 1) {
                err = "gpu extraction: the run count changed between two launches";
                return ok = false;
            }
            cap = produced;
        }
        lap("k_extract_runs");
        release((void*&)d_seq);
        if (produced) {
            size_t tmp_bytes = 0;
            STEP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_mm[0], d_mm[1], d_val[0], d_val[1], (size_t)produced, 0u, 2u * m, stream));
            STEP(hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)));
            STEP(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_mm[0], d_mm[1], d_val[0], d_val[1], (size_t)produced, 0u, 2u * m, stream));
            STEP(hipStreamSynchronize(stream));
            release((void*&)d_mm[0]);  // the unsorted buffers and the sort's scratch have done their job
            release((void*&)d_val[0]);
            release(d_tmp);
            lap("radix sort by minimizer");
        }
        return true;
    }

    // bytes from device memory into pageable host memory through the two pinned buffers: the copy of one piece overlaps with the host-side move
    // (all threads: also the first touch of the destination's pages) of the piece before
    bool download(void* dst, const void* d_src, size_t bytes) {
        size_t pending_at = 0, pending_n = 0;
        int which = 0, pending_which = -1;
        for (size_t at = 0; at < bytes || pending_which >= 0;) {
            size_t n = 0;
            if (at < bytes) {
                n = std::min(kStage, bytes - at);
                STEP(hipMemcpyAsync(h_stage[which], (const char*)d_src + at, n, hipMemcpyDeviceToHost, stream));
                STEP(hipEventRecord(ev[which], stream));
            }
            if (pending_which >= 0) {
                STEP(hipEventSynchronize(ev[pending_which]));
                scatter_bytes(h_stage[pending_which], (char*)dst + pending_at, pending_n, threads);
                pending_which = -1;
            }
            if (n) {
                pending_which = which;
                pending_at = at;
                pending_n = n;
                at += n;
                which ^= 1;
            }
        }
        return true;
    }
};

}  // namespace

bool gpu_extract_sorted_runs(int device, const char* const* records, const size_t* lengths, size_t n_records, uint64_t first_base, uint32_t k, uint32_t m,
                             uint64_t mm_seed, int threads, bool verbose, SuperKmerRuns& out, uint64_t& lost_windows, std::string& err) {
    const size_t out_before = out.size();
    lost_windows = 0;
    Extractor x(err, verbose, threads);
    auto body = [&]() -> bool {
        if (!x.extract_and_sort(device, records, lengths, n_records, first_base, k, m, mm_seed)) return false;
        lost_windows = x.lost;
        if (!x.produced) return true;
        SuperKmerRun* d_runs = nullptr;
        if (!x.step(hipMalloc((void**)&d_runs, (size_t)x.produced * sizeof(SuperKmerRun)), "hipMalloc(records)")) return false;
        x.d_tmp = d_runs;  // (released with the rest)
        k_assemble_runs<<<(unsigned)((x.produced + 255) / 256), 256, 0, x.stream>>>(x.d_mm[1], x.d_val[1], (size_t)x.produced, d_runs);
        if (!x.step(hipGetLastError(), "k_assemble_runs") || !x.step(hipStreamSynchronize(x.stream), "hipStreamSynchronize")) return false;
        out.resize(out_before + (size_t)x.produced);
        if (!x.download(out.data() + out_before, d_runs, (size_t)x.produced * sizeof(SuperKmerRun))) return false;
        x.lap("records to the host");
        return true;
    };
    const bool ok = body() && x.ok;
    if (!ok) out.resize(out_before);
    return ok;
}

bool gpu_extract_classified(int device, const char* const* records, const size_t* lengths, size_t n_records, uint32_t k, uint32_t m, uint64_t mm_seed, int threads,
                            bool verbose, ClassifiedMinimizers& out, std::string& err) {
    out = ClassifiedMinimizers();
    Extractor x(err, verbose, threads);
    auto body = [&]() -> bool {
        if (!x.extract_and_sort(device, records, lengths, n_records, 0, k, m, mm_seed)) return false;
        const size_t n = (size_t)x.produced;
        out.num_super_kmers = n;
        out.num_kmers = x.lost;  // (+ the k-mers of the super-k-mers, below)
        if (!n) return true;
        // flags: low word 1 = first record of its minimizer, high word 1 = record of a minimizer that has several; an exclusive scan of the
        // packed pair gives every record its slot among the distinct minimizers and among the colliding records
        uint64_t *d_flags = nullptr, *d_slots = nullptr, *d_umm = nullptr, *d_cfirst = nullptr;
        uint16_t* d_ups = nullptr;
        uint32_t* d_csize = nullptr;
        void* d_scan_tmp = nullptr;
        bool good = true;
        auto S = [&](hipError_t e, const char* what) { return good = good && x.step(e, what); };
        unsigned long long totals[2] = {0, 0};
        uint64_t last_flag = 0, last_slot = 0;
        size_t tmp_bytes = 0;
        const unsigned grid = (unsigned)((n + 255) / 256);
        do {
            if (!S(hipMalloc((void**)&d_flags, n * 8), "hipMalloc(flags)") || !S(hipMalloc((void**)&d_slots, n * 8), "hipMalloc(slots)")) break;
            k_group_flags<<<grid, 256, 0, x.stream>>>(x.d_mm[1], x.d_val[1], n, d_flags, x.d_cursor + 1);  // (d_cursor[1]: zeroed before the extraction)
            if (!S(hipGetLastError(), "k_group_flags")) break;
            if (!S(rocprim::exclusive_scan(nullptr, tmp_bytes, d_flags, d_slots, (uint64_t)0, n, rocprim::plus<uint64_t>(), x.stream), "exclusive_scan (size)")) break;
            if (!S(hipMalloc(&d_scan_tmp, std::max<size_t>(tmp_bytes, 16)), "hipMalloc(scan)")) break;
            if (!S(rocprim::exclusive_scan(d_scan_tmp, tmp_bytes, d_flags, d_slots, (uint64_t)0, n, rocprim::plus<uint64_t>(), x.stream), "exclusive_scan")) break;
            if (!S(hipMemcpyAsync(&last_flag, d_flags + (n - 1), 8, hipMemcpyDeviceToHost, x.stream), "copy") ||
                !S(hipMemcpyAsync(&last_slot, d_slots + (n - 1), 8, hipMemcpyDeviceToHost, x.stream), "copy") ||
                !S(hipMemcpyAsync(totals, x.d_cursor, sizeof totals, hipMemcpyDeviceToHost, x.stream), "copy") || !S(hipStreamSynchronize(x.stream), "sync"))
                break;
            const uint64_t sum = last_flag + last_slot;
            const size_t n_uniq = (size_t)(sum & 0xFFFFFFFFull), n_coll = (size_t)(sum >> 32);
            out.num_kmers = totals[1] + x.lost;  // the reference's count: every window of k valid bases, also the ones it leaves out of every super-k-mer
            if (!S(hipMalloc((void**)&d_umm, std::max<size_t>(n_uniq, 1) * 8), "hipMalloc") || !S(hipMalloc((void**)&d_ups, std::max<size_t>(n_uniq, 1) * 2), "hipMalloc") ||
                !S(hipMalloc((void**)&d_cfirst, std::max<size_t>(n_coll, 1) * 8), "hipMalloc") || !S(hipMalloc((void**)&d_csize, std::max<size_t>(n_coll, 1) * 4), "hipMalloc"))
                break;
            k_group_scatter<<<grid, 256, 0, x.stream>>>(x.d_mm[1], x.d_val[1], d_slots, n, d_umm, d_ups, d_cfirst, d_csize);
            if (!S(hipGetLastError(), "k_group_scatter") || !S(hipStreamSynchronize(x.stream), "sync")) break;
            x.lap("classification");
            out.minimizer.resize(n_uniq);
            out.offset_size.resize(n_uniq);
            out.colliding_first.resize(n_coll);
            out.colliding_size.resize(n_coll);
            good = good && x.download(out.minimizer.data(), d_umm, n_uniq * 8) && x.download(out.offset_size.data(), d_ups, n_uniq * 2) &&
                   x.download(out.colliding_first.data(), d_cfirst, n_coll * 8) && x.download(out.colliding_size.data(), d_csize, n_coll * 4);
            x.lap("minimizers to the host");
        } while (false);
        for (void* p : {(void*)d_flags, (void*)d_slots, (void*)d_umm, (void*)d_cfirst, (void*)d_ups, (void*)d_csize, d_scan_tmp})
            if (p) (void)hipFree(p);
        return good;
    };
    const bool ok = body() && x.ok;
    if (!ok) out = ClassifiedMinimizers();
    return ok;
}

}  // namespace lphq
