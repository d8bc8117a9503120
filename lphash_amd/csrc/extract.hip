// extract.hip -- index construction, parts 1-2 on the GPU ("minimizer extraction + radix sort of
// mm_record_t on GPU"): the super-k-mer records of a batch of records (k_extract_runs, extract_device.cuh), sorted by minimizer
// (rocPRIM radix sort: a plain library sort of (key, value) pairs; the key is the 2m-bit minimizer).
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

#include "extract.hpp"
#include "extract_device.cuh"

namespace lphq {
namespace {

#define HIP_OK(call)                                                 \
    do {                                                             \
        hipError_t e_ = (call);                                      \
        if (e_ != hipSuccess) {                                      \
            err = std::string(#call) + ": " + hipGetErrorString(e_); \
            goto done;                                               \
        }                                                            \
    } while (0)

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

}  // namespace

bool gpu_extract_sorted_runs(int device, const char* const* records, const size_t* lengths, size_t n_records, uint64_t first_base, uint32_t k, uint32_t m,
                             uint64_t mm_seed, int threads, bool verbose, SuperKmerRuns& out, std::string& err) {
    bool ok = false;
    const size_t out_before = out.size();
    SuperKmerRun* d_runs = nullptr;
    auto tprev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        auto now = std::chrono::steady_clock::now();
        if (verbose) fprintf(stderr, "[lphq_build]   gpu extraction: %-30s %8.3f s\n", what, std::chrono::duration<double>(now - tprev).count());
        tprev = now;
    };
    int prev = -1;
    uint8_t* d_seq = nullptr;
    uint64_t *d_off = nullptr, *d_mm[2] = {nullptr, nullptr};
    RunVal* d_val[2] = {nullptr, nullptr};
    unsigned long long* d_cursor = nullptr;
    void* d_tmp = nullptr;
    char* h_stage[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    std::vector<uint64_t> off(n_records + 1, 0);
    for (size_t r = 0; r < n_records; ++r) off[r + 1] = off[r] + lengths[r];
    const uint64_t total = off[n_records];
    const uint32_t W = k - m + 1u;
    if (n_records == 0 || total < k) return true;  // no window at all
    if (n_records > 0xFFFFFFFFull) {
        err = "gpu extraction: more than 2^32 - 1 records in one batch";
        return false;
    }
    uint64_t windows = 0;
    for (size_t r = 0; r < n_records; ++r)
        if (lengths[r] >= k) windows += lengths[r] - k + 1;
    if (windows == 0) return true;
    (void)hipGetDevice(&prev);
    {
        HIP_OK(hipSetDevice(device));
        HIP_OK(hipStreamCreate(&stream));
        HIP_OK(hipMalloc((void**)&d_seq, (size_t)total + 2 * kFront + 16));
        HIP_OK(hipMalloc((void**)&d_off, (n_records + 1) * sizeof(uint64_t)));
        HIP_OK(hipMalloc((void**)&d_cursor, sizeof(unsigned long long)));
        HIP_OK(hipMemsetAsync(d_seq, 0, kFront, stream));
        HIP_OK(hipMemsetAsync(d_seq + kFront + total, 0, kFront + 16, stream));
        HIP_OK(hipMemcpyAsync(d_off, off.data(), (n_records + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        // the bases: the concatenated records through two pinned staging buffers (records may be many and small, or few and huge)
        for (int i = 0; i < 2; ++i) {
            HIP_OK(hipHostMalloc((void**)&h_stage[i], kStage, hipHostMallocDefault));
            HIP_OK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        lap("device and pinned buffers");
        {
            int which = 0;
            for (uint64_t at = 0; at < total; at += kStage, which ^= 1) {
                const size_t n = (size_t)std::min<uint64_t>(kStage, total - at);
                HIP_OK(hipEventSynchronize(ev[which]));  // the copy that last used this buffer (a fresh event is complete)
                gather_bytes(records, lengths, off, at, n, h_stage[which], threads);
                HIP_OK(hipMemcpyAsync(d_seq + kFront + at, h_stage[which], n, hipMemcpyHostToDevice, stream));
                HIP_OK(hipEventRecord(ev[which], stream));
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
        unsigned long long produced = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            for (int i = 0; i < 2; ++i) {
                if (d_mm[i]) (void)hipFree(d_mm[i]);
                if (d_val[i]) (void)hipFree(d_val[i]);
                d_mm[i] = nullptr;
                d_val[i] = nullptr;
            }
            for (int i = 0; i < 2; ++i) {
                HIP_OK(hipMalloc((void**)&d_mm[i], (size_t)cap * sizeof(uint64_t)));
                HIP_OK(hipMalloc((void**)&d_val[i], (size_t)cap * sizeof(RunVal)));
            }
            HIP_OK(hipMemsetAsync(d_cursor, 0, sizeof(unsigned long long), stream));
            const uint64_t kMaxTiles = (1u << 24) - 1;
            for (uint64_t tb = 0; tb < tiles; tb += kMaxTiles) {
                const unsigned grid = (unsigned)std::min<uint64_t>(kMaxTiles, tiles - tb);
                if (W >= kBlockMinW)
                    k_extract_runs<kStreamBlockDefault, true><<<grid, kStreamBlockDefault, 0, stream>>>(seq_al, shift, total, d_off, (uint32_t)n_records, k, m, mm_seed,
                                                                                                         first_base, d_mm[0], d_val[0], d_cursor, cap, tb);
                else
                    k_extract_runs<kStreamBlockDefault, false><<<grid, kStreamBlockDefault, 0, stream>>>(seq_al, shift, total, d_off, (uint32_t)n_records, k, m, mm_seed,
                                                                                                          first_base, d_mm[0], d_val[0], d_cursor, cap, tb);
                HIP_OK(hipGetLastError());
            }
            HIP_OK(hipMemcpyAsync(&produced, d_cursor, sizeof produced, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
            if (produced <= cap) break;
            if (attempt == 1) {
                err = "gpu extraction: the run count changed between two launches";
                goto done;
            }
            cap = produced;
        }
        lap("k_extract_runs");
        if (produced) {
            size_t tmp_bytes = 0;
            HIP_OK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_mm[0], d_mm[1], d_val[0], d_val[1], (size_t)produced, 0u, 2u * m, stream));
            HIP_OK(hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)));
            HIP_OK(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_mm[0], d_mm[1], d_val[0], d_val[1], (size_t)produced, 0u, 2u * m, stream));
            // the unsorted buffers have done their job: the builder's records go where they were (32 B each = the 8 + 16 of a pair and room to spare
            // only if both are reused; a fresh buffer keeps this simple)
            (void)hipStreamSynchronize(stream);
            (void)hipFree(d_mm[0]);
            (void)hipFree(d_val[0]);
            d_mm[0] = nullptr;
            d_val[0] = nullptr;
            HIP_OK(hipMalloc((void**)&d_runs, (size_t)produced * sizeof(SuperKmerRun)));
            k_assemble_runs<<<(unsigned)((produced + 255) / 256), 256, 0, stream>>>(d_mm[1], d_val[1], (size_t)produced, d_runs);
            HIP_OK(hipGetLastError());
            HIP_OK(hipStreamSynchronize(stream));
            lap("radix sort by minimizer");
            // out through the two pinned buffers: the copy of one piece overlaps with the host-side move of the piece before
            out.resize(out_before + (size_t)produced);
            const size_t per = kStage / sizeof(SuperKmerRun);
            size_t pending_at = 0, pending_n = 0;
            int which = 0, pending_which = -1;
            for (size_t at = 0; at < (size_t)produced || pending_which >= 0;) {
                size_t n = 0;
                if (at < (size_t)produced) {
                    n = std::min(per, (size_t)produced - at);
                    HIP_OK(hipMemcpyAsync(h_stage[which], d_runs + at, n * sizeof(SuperKmerRun), hipMemcpyDeviceToHost, stream));
                    HIP_OK(hipEventRecord(ev[which], stream));
                }
                if (pending_which >= 0) {
                    HIP_OK(hipEventSynchronize(ev[pending_which]));
                    scatter_bytes(h_stage[pending_which], (char*)(out.data() + out_before + pending_at), pending_n * sizeof(SuperKmerRun), threads);
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
            lap("records to the host");
        }
        ok = true;
    }
done:
    if (stream) (void)hipStreamSynchronize(stream);
    for (int i = 0; i < 2; ++i) {
        if (d_mm[i]) (void)hipFree(d_mm[i]);
        if (d_val[i]) (void)hipFree(d_val[i]);
        if (h_stage[i]) (void)hipHostFree(h_stage[i]);
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    }
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_runs) (void)hipFree(d_runs);
    if (d_seq) (void)hipFree(d_seq);
    if (d_off) (void)hipFree(d_off);
    if (d_cursor) (void)hipFree(d_cursor);
    if (stream) (void)hipStreamDestroy(stream);
    if (prev >= 0) (void)hipSetDevice(prev);
    if (!ok) out.resize(out_before);
    return ok;
}

}  // namespace lphq
