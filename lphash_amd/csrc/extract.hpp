// extract.hpp -- super-k-mer extraction + sort on the GPU (extract.hip), parts 1-2 of index construction.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace lphq {
struct RunVal {      // what travels with the minimizer through the sort
    uint64_t first;  // global coordinate (base index over the whole input) of the run's first window
    uint32_t p1;     // offset of the minimizer inside that window (mm_record_t::p1)
    uint32_t size;   // number of windows (mm_record_t::size)
};
struct SuperKmerRun {  // one super-k-mer: windows [first, first + size) share the minimizer at q (the builder's record; spilled to disk as is)
    uint64_t mm;        // minimizer value (2-bit packed m-mer)
    uint64_t first;     // global coordinate of the first window
    uint64_t q;         // global coordinate of the minimizer
    uint32_t size;
    uint32_t pad;       // always 0: no indeterminate bytes
};
// std::allocator whose construct() default-initialises: resize() of a vector of records does not touch (and page in) the memory the next step
// overwrites anyway -- with GBs of records that first touch, on one thread, costs more than the GPU work it follows.
template <typename T>
struct DefaultInitAllocator : std::allocator<T> {
    template <typename U>
    struct rebind {
        typedef DefaultInitAllocator<U> other;
    };
    DefaultInitAllocator() = default;
    template <typename U>
    DefaultInitAllocator(const DefaultInitAllocator<U>&) {}
    template <typename U>
    void construct(U* p) {
        ::new ((void*)p) U;
    }
    template <typename U, typename... A>
    void construct(U* p, A&&... a) {
        ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
typedef std::vector<SuperKmerRun, DefaultInitAllocator<SuperKmerRun>> SuperKmerRuns;

// minimizer::classify (src/minimizer.cpp:5-50) done on the device: the distinct minimizers ascending, each with (offset | size << 8) of its one
// super-k-mer or 0 when it has several; the (first window, size) of every super-k-mer of those, in no particular order; totals.
struct ClassifiedMinimizers {
    std::vector<uint64_t, DefaultInitAllocator<uint64_t>> minimizer;
    std::vector<uint16_t, DefaultInitAllocator<uint16_t>> offset_size;
    std::vector<uint64_t, DefaultInitAllocator<uint64_t>> colliding_first;
    std::vector<uint32_t, DefaultInitAllocator<uint32_t>> colliding_size;
    uint64_t num_super_kmers = 0, num_kmers = 0;
};
// The whole input as ONE batch (global coordinates from 0): extraction, sort and classification on the device.  false + err when the device path is
// unavailable or fails.
bool gpu_extract_classified(int device, const char* const* records, const size_t* lengths, size_t n_records, uint32_t k, uint32_t m, uint64_t mm_seed, int threads,
                            bool verbose, ClassifiedMinimizers& out, std::string& err);

// Every super-k-mer of the batch (whole records; first_base = global coordinate of the batch's first base) APPENDED to `out`, ascending by
// minimizer (ties in no particular order).  false + err when the device path is unavailable or fails (out is then as it was) -- the caller
// runs the host extraction instead, loudly.  lost_windows: valid windows that belong to no super-k-mer because the reference leaves them out (the one window
// of a stretch of exactly k valid bases closed by a non-ACGT base inside its record, builder.cpp: lost_window) -- they count as k-mers.
bool gpu_extract_sorted_runs(int device, const char* const* records, const size_t* lengths, size_t n_records, uint64_t first_base, uint32_t k, uint32_t m,
                             uint64_t mm_seed, int threads, bool verbose, SuperKmerRuns& out, uint64_t& lost_windows, std::string& err);
}  // namespace lphq
