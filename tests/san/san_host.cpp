// TEST INFRASTRUCTURE ONLY: the product's host-side sources that need no GPU (loader lph_format.cpp, image builder image.cpp,
// index builder builder.cpp with the CPU pilot search, the FASTA readers) linked into one executable and compiled with
// -fsanitize=address,undefined (tests/test_host_cpu.py::test_host_sources_under_sanitizers).  The symbols that live in the HIP
// part of the library are stubbed: the GPU pilot search and the GPU super-k-mer extraction report "unavailable", errors go to a string.
//   san_host build <fasta> <k> <m> <kind> <kmer_bits> <threads> <out.lph>     build with the product builder (CPU search)
//   san_host build_stream <fasta> <k> <m> <kind> <kmer_bits> <threads> <out.lph> <max_memory_bytes> <tmp_dir>   lphq_build_stream, small batches
//   san_host inflate <file.gz>                                                 the .gz through GzFastSource (decoder + ring + CRC thread)
//   san_host inflate_fuzz <file.gz> <n> <seed>                                  n corrupted / truncated copies through DeflateDecoder
//   san_host inflate_par <file.gz> <threads> <span>                            the .gz through ParallelGzReader (every piece into a heap block of its exact size)
//   san_host inflate_par_fuzz <file.gz> <n> <seed> <threads> <span> <tmp>      n corrupted / truncated copies (written to <tmp>) through ParallelGzReader
//   san_host load  <file.lph> <kind> <kmer_bits>                             parse + cross-check + flat image
//   san_host fuzz  <file.lph> <kind> <kmer_bits> <n> <seed>                  n corrupted copies through the loader
//   san_host limits <file.lph> <kind> <kmer_bits>                            the parsed file with each count of the device layout's 32-bit limits forged past it
//   san_host runs <n_windows> <seed> <threads>                               random run-encoded results (runs_host.hpp) expanded by the pool == a serial expansion; plan_runs
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/lphq.h"
#include "fastx_batch.hpp"
#include "gz_source.hpp"
#include "parallel_gz.hpp"
#include "image.hpp"
#include "lph_format.hpp"
#include "extract.hpp"
#include "search.hpp"
#include "runs_host.hpp"

namespace lphq {
static std::string g_err;
void set_last_error(const std::string& e) { g_err = e; }
bool gpu_pilot_search(int, const std::vector<uint64_t>&, const std::vector<uint64_t>&, const std::vector<uint32_t>&, uint64_t, uint64_t, uint64_t,
                      std::vector<uint64_t>&, std::vector<uint64_t>&, std::string& err) {
    err = "no GPU in the sanitizer harness";
    return false;
}
bool gpu_extract_sorted_runs(int, const char* const*, const size_t*, size_t, uint64_t, uint32_t, uint32_t, uint64_t, int, bool, SuperKmerRuns&, uint64_t&,
                             std::string& err) {
    err = "no GPU in the sanitizer harness";
    return false;
}
bool gpu_extract_classified(int, const char* const*, const size_t*, size_t, uint32_t, uint32_t, uint64_t, int, bool, ClassifiedMinimizers&, std::string& err) {
    err = "no GPU in the sanitizer harness";
    return false;
}
}  // namespace lphq

static std::vector<uint8_t> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static int load(const std::vector<uint8_t>& buf, int kind, int bits, bool verbose) {
    lphq::LphFile x;
    std::string err;
    if (!lphq::parse_lph(buf.data(), buf.size(), kind, bits, x, err)) {
        if (verbose) fprintf(stderr, "rejected: %s\n", err.c_str());
        return 1;
    }
    std::vector<uint8_t> image;
    if (!lphq::build_image(x, image, err)) {
        if (verbose) fprintf(stderr, "image: %s\n", err.c_str());
        return 2;
    }
    if (verbose) printf("ok %zu image bytes\n", image.size());
    return 0;
}

// The device layout is narrower than the format: < 2^32 keys / table positions, buckets, distinct minimizers.  A file that large cannot be
// forged byte by byte (its vectors would have to be that long to pass the loader's cross-checks), so the PARSED file is forged: each limit must be refused by
// build_image with the text lphq_open maps to LPHQ_ERR_UNSUPPORTED (capi.cpp: "beyond the device layout"), never hashed with truncated 32-bit arithmetic.
static int limits(const std::vector<uint8_t>& buf, int kind, int bits) {
    lphq::LphFile good;
    std::string err;
    if (!lphq::parse_lph(buf.data(), buf.size(), kind, bits, good, err)) return 1;
    const uint64_t big = 1ull << 32;
    struct Case {
        const char* name;
        void (*forge)(lphq::LphFile&, uint64_t);
    } cases[] = {
        {"num_keys", [](lphq::LphFile& x, uint64_t b) { x.minimizer_order.num_keys = b; }},
        {"table_size", [](lphq::LphFile& x, uint64_t b) { x.minimizer_order.table_size = b + 7; }},
        {"num_dense", [](lphq::LphFile& x, uint64_t b) { x.minimizer_order.num_dense = b; }},
        {"num_sparse", [](lphq::LphFile& x, uint64_t b) { x.minimizer_order.num_sparse = b; }},
        {"buckets", [](lphq::LphFile& x, uint64_t b) { x.minimizer_order.num_dense = b / 2, x.minimizer_order.num_sparse = b / 2; }},
        {"distinct_minimizers", [](lphq::LphFile& x, uint64_t b) { x.distinct_minimizers = b; }},
        {"fallback_keys", [](lphq::LphFile& x, uint64_t b) { x.fallback_kmer_order.num_keys = b; }},
    };
    int bad = 0;
    for (const Case& c : cases) {
        lphq::LphFile x = good;
        c.forge(x, big);
        std::vector<uint8_t> image;
        err.clear();
        const bool ok = lphq::build_image(x, image, err);
        const bool refused = !ok && err.find("beyond the device layout") != std::string::npos;
        printf("%s: %s (%s)\n", c.name, refused ? "refused" : "NOT REFUSED", err.c_str());
        bad += !refused;
    }
    std::vector<uint8_t> image;
    if (!lphq::build_image(good, image, err)) return 3;  // (the unforged file still builds)
    return bad ? 2 : 0;
}

struct Source {  // lphq_record_source over a file: batches of ~20,000 bases, the file reopened for the second pass
    const char* path;
    int threads, pass = -1;
    lphq::FastxBatchReader* rd = nullptr;
    lphq::FastxBatch b;
    std::vector<const char*> ptrs;
    std::vector<size_t> lens;
    ~Source() { delete rd; }
    static int next(void* ctx, int pass, const char* const** records, const size_t** lengths, size_t* n) {
        Source* s = (Source*)ctx;
        if (s->pass != pass) {
            delete s->rd;
            s->rd = new lphq::FastxBatchReader(s->path, s->threads, nullptr, nullptr, 4096);
            s->pass = pass;
        }
        if (!s->rd->ok()) return -1;
        if (!s->rd->next(s->b, 20000)) return 0;
        s->b.compact();
        s->ptrs.resize(s->b.lens.size());
        s->lens.resize(s->b.lens.size());
        size_t so = 0;
        for (size_t i = 0; i < s->lens.size(); ++i) {
            s->lens[i] = (size_t)s->b.lens[i];
            s->ptrs[i] = s->b.segs.empty() ? "" : s->b.segs[0].p + so;
            so += s->lens[i];
        }
        *records = s->ptrs.data();
        *lengths = s->lens.data();
        *n = s->lens.size();
        return 1;
    }
};

int main(int argc, char** argv) {
    if (argc < 2) return 64;
    const std::string cmd = argv[1];
    if (cmd == "limits" && argc == 5) return limits(slurp(argv[2]), atoi(argv[3]), atoi(argv[4]));
    if (cmd == "build_stream" && argc == 11) {
        Source src;
        src.path = argv[2];
        src.threads = atoi(argv[7]);
        lphq_build_stats st;
        const int rc = lphq_build_stream(&Source::next, &src, atoi(argv[3]), atoi(argv[4]), 42, 3.0, atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), -1, argv[10],
                                         strtoull(argv[9], nullptr, 10), argv[8], &st);
        if (rc != LPHQ_OK) fprintf(stderr, "build failed (%d): %s\n", rc, lphq::g_err.c_str());
        return rc == LPHQ_OK ? 0 : 4;
    }
    if (cmd == "build" && argc == 9) {
        lphq::FastxBatchReader rd(argv[2], atoi(argv[7]), nullptr, nullptr, 4096);
        if (!rd.ok()) return 3;
        lphq::FastxBatch b;
        (void)rd.next(b, ~(size_t)0);
        b.compact();
        std::vector<const char*> ptrs(b.lens.size());
        std::vector<size_t> lens(b.lens.size());
        size_t so = 0;
        for (size_t i = 0; i < lens.size(); ++i) {
            lens[i] = (size_t)b.lens[i];
            ptrs[i] = b.segs.empty() ? "" : b.segs[0].p + so;
            so += lens[i];
        }
        lphq_build_stats st;
        const int rc = lphq_build(ptrs.data(), lens.data(), lens.size(), atoi(argv[3]), atoi(argv[4]), 42, 3.0, atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), argv[8], &st);
        if (rc != LPHQ_OK) fprintf(stderr, "build failed (%d): %s\n", rc, lphq::g_err.c_str());
        return rc == LPHQ_OK ? 0 : 4;
    }
    if (cmd == "inflate" && argc == 3) {
        lphq::GzFastSource* src = lphq::GzFastSource::open(argv[2]);
        if (!src) return 3;
        std::vector<char> buf(1 << 20);
        uint64_t total = 0, h = 1469598103934665603ull;
        size_t n;
        while ((n = src->read(buf.data(), 1 + (total * 7919) % buf.size())) > 0) {  // odd read sizes: chunk borders inside and between reads
            for (size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)buf[i]) * 1099511628211ull;
            total += n;
        }
        printf("%llu,%016llx,%s\n", (unsigned long long)total, (unsigned long long)h, src->error() ? "error" : "ok");
        delete src;
        return 0;
    }
    if ((cmd == "inflate_par" && argc == 5) || (cmd == "inflate_par_fuzz" && argc == 8)) {
        // one file through the parallel decoder; pieces go into exact-size heap blocks (ASan sees a write past a piece), several pieces in flight
        auto run = [&](const char* path, int threads, size_t span, uint64_t& total, uint64_t& h, bool& err) -> bool {
            lphq::ParallelGzReader* src = lphq::ParallelGzReader::open(path, threads, span);
            if (!src) return false;
            src->prealloc();
            total = 0;
            h = 1469598103934665603ull;
            std::vector<std::vector<uint8_t>> blocks;
            auto drain = [&] {
                src->flush();
                for (auto& b : blocks)
                    for (uint8_t c : b) h = (h ^ c) * 1099511628211ull;
                blocks.clear();
            };
            size_t n;
            while ((n = src->next_len()) > 0) {
                blocks.emplace_back(n);
                src->emit(blocks.back().data());
                total += n;
                if (blocks.size() >= 5) drain();
            }
            drain();
            err = src->error();
            delete src;
            return true;
        };
        uint64_t total = 0, h = 0;
        bool err = false;
        if (cmd == "inflate_par") {
            if (!run(argv[2], atoi(argv[3]), (size_t)atol(argv[4]), total, h, err)) return 3;
            printf("%llu,%016llx,%s\n", (unsigned long long)total, (unsigned long long)h, err ? "error" : "ok");
            return 0;
        }
        const std::vector<uint8_t> good = slurp(argv[2]);
        uint64_t s = strtoull(argv[4], nullptr, 10) | 1;
        auto rnd = [&] {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            return s;
        };
        int fine = 0, errors = 0, declined = 0;
        for (int t = 0; t < atoi(argv[3]); ++t) {
            std::vector<uint8_t> bad = good;
            switch (rnd() % 3) {
                case 0: bad.resize(18 + rnd() % (bad.size() - 17)); break;
                case 1: for (int j = 0; j < 1 + (int)(rnd() % 6); ++j) bad[10 + rnd() % (bad.size() - 10)] ^= (uint8_t)(1u << (rnd() % 8)); break;
                default: for (size_t j = 10 + rnd() % (bad.size() - 10), e = std::min(bad.size(), j + 1 + rnd() % 64); j < e; ++j) bad[j] = (uint8_t)rnd(); break;
            }
            {
                std::ofstream f(argv[7], std::ios::binary | std::ios::trunc);
                f.write((const char*)bad.data(), (std::streamsize)bad.size());
            }
            if (!run(argv[7], atoi(argv[5]), (size_t)atol(argv[6]), total, h, err)) ++declined;
            else (err ? errors : fine)++;
        }
        printf("parallel inflate fuzz: %d fine, %d errors, %d declined\n", fine, errors, declined);
        return 0;
    }
    if (cmd == "inflate_fuzz" && argc == 5) {
        const std::vector<uint8_t> good = slurp(argv[2]);
        uint64_t s = strtoull(argv[4], nullptr, 10) | 1;
        auto rnd = [&] {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            return s;
        };
        static lphq::DeflateDecoder dec;
        int ended = 0, errors = 0, full = 0;
        for (int t = 0; t < atoi(argv[3]); ++t) {
            std::vector<uint8_t> bad = good;
            switch (rnd() % 3) {
                case 0: bad.resize(18 + rnd() % (bad.size() - 17)); break;
                case 1: for (int j = 0; j < 1 + (int)(rnd() % 6); ++j) bad[10 + rnd() % (bad.size() - 10)] ^= (uint8_t)(1u << (rnd() % 8)); break;
                default: for (size_t j = 10 + rnd() % (bad.size() - 10), e = std::min(bad.size(), j + 1 + rnd() % 64); j < e; ++j) bad[j] = (uint8_t)rnd(); break;
            }
            const size_t hdr = lphq::gzip_header_bytes(bad.data(), bad.size());
            if (!hdr) continue;
            // exact-size heap blocks: ASan sees any read past the input and any write past limit + kSlack
            std::vector<uint8_t> in(bad.begin() + (long)hdr, bad.end());
            const size_t limit = 64 + rnd() % (6 * good.size());
            std::vector<uint8_t> out(limit + 1 + lphq::DeflateDecoder::kSlack);
            dec.reset(in.data(), in.data() + in.size());
            size_t pos = 0;
            const int r = dec.decode(out.data(), pos, limit);
            if (pos > out.size()) return 9;
            (r == 1 ? ended : r == 0 ? full : errors)++;
        }
        printf("inflate fuzz: %d ended, %d out of room, %d errors\n", ended, full, errors);
        return 0;
    }
    if (cmd == "runs" && argc == 5) {
        // a batch of n windows cut into random runs (some windows in no run: they hold a non-ACGT base), kinds UP / DOWN / EXPLICIT, in shuffled order as the
        // device leaves them: the pool's expansion must equal a serial one, touch nothing outside the runs and nothing beyond the buffer (exact-size heap blocks)
        const size_t n = (size_t)atoll(argv[2]);
        uint64_t x = strtoull(argv[3], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
        auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        std::vector<lphq_run> runs;
        std::vector<uint64_t> xc, want(n, 0xABABABABABABABABull);
        for (size_t w = 0; w < n;) {
            if (rnd() % 11 == 0) { ++w; continue; }
            const uint32_t len = (uint32_t)std::min<uint64_t>(n - w, 1 + rnd() % (rnd() % 7 == 0 ? 512 : 24)), kind = (uint32_t)(rnd() % 3);
            lphq_run r;
            r.where = (uint64_t)w | ((uint64_t)len << 40) | ((uint64_t)kind << 62);
            if (kind == LPHQ_RUN_EXPLICIT) {
                r.code = xc.size();
                for (uint32_t i = 0; i < len; ++i) { xc.push_back(rnd()); want[w + i] = xc.back(); }
            } else {
                r.code = rnd() % 3 == 0 ? rnd() % 5 : rnd();  // (wrapping below zero / above 2^64 - 1 like the reference's arithmetic)
                for (uint32_t i = 0; i < len; ++i) want[w + i] = kind == LPHQ_RUN_UP ? r.code + i : r.code - i;
            }
            if (LPHQ_RUN_FIRST(r) != w || LPHQ_RUN_LENGTH(r) != len || LPHQ_RUN_KIND(r) != kind) return 3;
            runs.push_back(r);
            w += len;
        }
        for (size_t i = runs.size(); i > 1; --i) std::swap(runs[i - 1], runs[rnd() % i]);
        std::vector<uint64_t> got(n, 0xABABABABABABABABull);
        lphq::WorkPool pool(atoi(argv[4]));
        lphq::expand_runs_host(runs.data(), runs.size(), xc.data(), got.data(), &pool);
        if (got != want) return 4;
        std::fill(got.begin(), got.end(), 0xABABABABABABABABull);
        lphq::expand_runs_host(runs.data(), runs.size(), xc.data(), got.data(), nullptr);
        if (got != want) return 5;
        // the sizing of a session's arenas: whole arenas, room for the expectation on random sequence, and the worst case holds anything
        lphq_image_header H{};
        for (uint32_t W : {1u, 8u, 12u, 44u, 63u}) {
            H.W = W;
            const lphq::RunPlan p = lphq::plan_runs(H, n, n + 30, 7, 0, false), wc = lphq::plan_runs(H, n, n + 30, 7, 0, true);
            if (p.per_arena * LPHQ_RUN_ARENAS < 2 * n / (W + 1) || p.copy_runs > p.per_arena * LPHQ_RUN_ARENAS || wc.per_arena < std::min<size_t>(n, 1024)) return 6;
            if (wc.per_arena * LPHQ_RUN_ARENAS < std::min<size_t>(n, wc.per_arena * LPHQ_RUN_ARENAS)) return 7;
        }
        printf("runs ok: %zu runs, %zu explicit codes over %zu windows\n", runs.size(), xc.size(), n);
        return 0;
    }
    if (cmd == "load" && argc == 5) return load(slurp(argv[2]), atoi(argv[3]), atoi(argv[4]), true);
    if (cmd == "fuzz" && argc == 7) {
        const std::vector<uint8_t> good = slurp(argv[2]);
        uint64_t s = strtoull(argv[6], nullptr, 10) | 1;
        auto rnd = [&] {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            return s;
        };
        int ok = 0, rejected = 0;
        for (int t = 0; t < atoi(argv[5]); ++t) {
            std::vector<uint8_t> bad = good;
            switch (rnd() % 4) {
                case 0: bad.resize(rnd() % (bad.size() + 1)); break;                                        // truncation
                case 1: for (int j = 0; j < 1 + (int)(rnd() % 8); ++j) bad[rnd() % bad.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;  // bit flips
                case 2: { const size_t at = rnd() % bad.size() & ~(size_t)7; const uint64_t v = rnd() % 3 ? rnd() : ~0ull >> (rnd() % 64);
                          if (at + 8 <= bad.size()) memcpy(&bad[at], &v, 8); break; }                         // a wild 64-bit field
                default: bad.insert(bad.end(), (size_t)(rnd() % 64), (uint8_t)rnd()); break;                 // trailing bytes
            }
            if (bad.empty()) continue;
            (load(bad, atoi(argv[3]), atoi(argv[4]), false) == 0 ? ok : rejected)++;
        }
        printf("fuzz: %d accepted, %d rejected\n", ok, rejected);
        return 0;
    }
    return 64;
}
