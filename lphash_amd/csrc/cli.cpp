// cli.cpp -- `lphash` command line, drop-in for the reference driver (src/lphash.cpp:21-35):
//   lphash query-p|query-u -i <mphf> -q <fasta|fastq[.gz]>      src/query.cpp:14-88
//   lphash build-p|build-u -i <fasta[.gz]> -k K -m M [-s seed] [-t threads] [-o out] [-c c] [--check] [--verbose]
//                                                               src/build.cpp:10-70, src/parser_build.cpp:7-92
// Same flags, same CSV on stdout, same exit codes (1 usage/parse error, 2 cannot open the query file).
// Extra flags (all optional, defaults keep the reference behaviour): --kmer-bits 64|128 (the reference's
// compile-time kmer_t, default 128 like include/compile_constants.tpd:1; 64 reproduces a `uint64_t` build),
// --device N or --gpus N (replicas on devices 0..N-1, every batch sharded over them), --batch-mb N, --threads N (host threads of the
// query tools for .gz / BGZF decoding and for FASTA / FASTQ parsing; default: 16 for parsing, every usable CPU for decoding), --pageable (parse into
// ordinary instead of pinned host memory).  The work itself runs on the GPU through the C ABI (include/lphq.h).
//   lphash records -i <fasta|fastq[.gz]> [-t N] [--serial] [--batch-bytes N] [--min-piece N]
// is a diagnostic: it prints the record count, base count and a checksum of what the reader yields (tests compare the
// multi-threaded reader with the serial one through it).
#include <hip/hip_runtime.h>

#include <sys/stat.h>

#include <chrono>
#include <cstring>
#include <future>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lphq.h"
#include "fastx.hpp"
#include "fastx_batch.hpp"
#include "host_cpus.hpp"
#include "parallel_gz.hpp"

namespace {

int help(const char* arg0) {
    std::cerr << "LP-Hash: (L)ocality (P)reserving Minimal Perfect (Hash)ing of k-mers\n\n";
    std::cerr << "Usage: " << arg0 << " <tool> ...\n\n";
    std::cerr << "Available tools:\n";
    std::cerr << "  build-p      build a partitioned LP-MPHF\n";
    std::cerr << "  build-u      build an unpartitioned LP-MPHF\n";
    std::cerr << "  query-p      query a partitioned LP-MPHF\n";
    std::cerr << "  query-u      query an unpartitioned LP-MPHF\n";
    std::cerr << "  ptbb-build   build a plain PTHash MPHF over the k-mers (the reference's ptbb_build, PTHash half)\n";
    std::cerr << "  ptbb-query   query it (the reference's ptbb_query, PTHash half)\n";
    return 1;
}

struct Args {
    std::vector<std::pair<std::string, std::string>> kv;
    std::vector<std::string> flags;
    bool help = false, bad = false;
    const std::string* get(const std::string& k) const {
        for (auto& p : kv)
            if (p.first == k) return &p.second;
        return nullptr;
    }
    bool has(const std::string& f) const {
        for (auto& x : flags)
            if (x == f) return true;
        return false;
    }
};
Args parse(int argc, char** argv, const std::vector<std::string>& valued, const std::vector<std::string>& boolean) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string s = argv[i];
        if (s == "-h" || s == "--help") {
            a.help = true;
            continue;
        }
        bool known = false;
        for (auto& v : valued)
            if (s == v) {
                known = true;
                if (i + 1 >= argc) a.bad = true;
                else a.kv.emplace_back(s, argv[++i]);
            }
        for (auto& b : boolean)
            if (s == b) {
                known = true;
                a.flags.push_back(s);
            }
        if (!known) a.bad = true;
    }
    return a;
}

// Device side of one GPU of the query tools: its own stream and two slots of device buffers, used alternately, so that the
// copy-in of batch i+1 overlaps the kernels of batch i.  One host thread drives every GPU: all copies come from pinned
// memory and every call here is asynchronous, so the devices run concurrently without a thread each.
struct GpuWorker {
    struct Slot {
        char* d_seq = nullptr;
        uint64_t* d_offs = nullptr;  // seq offsets then out offsets
        uint64_t* d_out = nullptr;
        size_t seq_cap = 0, rec_cap = 0, out_cap = 0;
        uint64_t* h_offs = nullptr;  // pinned staging of the offsets
        size_t h_offs_cap = 0;
        uint64_t* h_codes = nullptr;  // pinned, only with --output
        size_t h_codes_cap = 0;
        uint64_t* d_count = nullptr;  // number of valid codes of the batch, counted on the device ...
        uint64_t* h_count = nullptr;  // ... and copied here (pinned)
        hipEvent_t copied = nullptr;  // the batch's bases have left the parse arena
        bool busy = false;
        size_t n_out = 0;
    };
    int device = 0;
    hipStream_t stream = nullptr;
    Slot slot[2];

    bool init(int dev) {
        device = dev;
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return false;
        for (Slot& s : slot)
            if (hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess || hipMalloc((void**)&s.d_count, 8) != hipSuccess ||
                hipHostMalloc((void**)&s.h_count, 8, hipHostMallocDefault) != hipSuccess)
                return false;
        return true;
    }
    void destroy() {
        (void)hipSetDevice(device);
        for (Slot& s : slot) {
            if (s.d_seq) (void)hipFree(s.d_seq);
            if (s.d_offs) (void)hipFree(s.d_offs);
            if (s.d_out) (void)hipFree(s.d_out);
            if (s.h_offs) (void)hipHostFree(s.h_offs);
            if (s.h_codes) (void)hipHostFree(s.h_codes);
            if (s.d_count) (void)hipFree(s.d_count);
            if (s.h_count) (void)hipHostFree(s.h_count);
            if (s.copied) (void)hipEventDestroy(s.copied);
        }
        if (stream) (void)hipStreamDestroy(stream);
    }
    static bool ensure(Slot& s, size_t bases, size_t recs, size_t outs, bool want_codes) {
        if (bases + 64 > s.seq_cap) {
            if (s.d_seq) (void)hipFree(s.d_seq);
            s.seq_cap = (bases + 64) * 5 / 4;
            if (hipMalloc((void**)&s.d_seq, s.seq_cap) != hipSuccess) return false;
        }
        if (2 * (recs + 1) > s.rec_cap) {
            if (s.d_offs) (void)hipFree(s.d_offs);
            s.rec_cap = 2 * (recs + 1) * 5 / 4;
            if (hipMalloc((void**)&s.d_offs, s.rec_cap * 8) != hipSuccess) return false;
        }
        if (2 * (recs + 1) > s.h_offs_cap) {
            if (s.h_offs) (void)hipHostFree(s.h_offs);
            s.h_offs_cap = 2 * (recs + 1) * 5 / 4;
            if (hipHostMalloc((void**)&s.h_offs, s.h_offs_cap * 8, hipHostMallocDefault) != hipSuccess) return false;
        }
        if (outs + 1 > s.out_cap) {
            if (s.d_out) (void)hipFree(s.d_out);
            s.out_cap = (outs + 1) * 5 / 4;
            if (hipMalloc((void**)&s.d_out, s.out_cap * 8) != hipSuccess) return false;
        }
        if (want_codes && outs + 1 > s.h_codes_cap) {
            if (s.h_codes) (void)hipHostFree(s.h_codes);
            s.h_codes_cap = (outs + 1) * 5 / 4;
            if (hipHostMalloc((void**)&s.h_codes, s.h_codes_cap * 8, hipHostMallocDefault) != hipSuccess) return false;
        }
        return true;
    }
};

// Parse arenas are kept between the two passes over the query file (streaming, then random): memory that has been touched
// once is several times cheaper to fill than fresh pages.  Blocks are handed back to the system when the process ends.
struct ArenaCache {
    struct Block {
        void* p;
        size_t cap;
        bool in_use;
    };
    std::vector<Block> blocks;
    bool pinned = false;
    void* get(size_t n) {
        Block* best = nullptr;
        for (Block& b : blocks)
            if (!b.in_use && b.cap >= n && (!best || b.cap < best->cap)) best = &b;
        if (best) {
            best->in_use = true;
            return best->p;
        }
        void* p = nullptr;
        if (pinned) {
            // (portable: one parse arena feeds the copies to EVERY device of --gpus N, whichever device was current when it was allocated)
            if (hipHostMalloc(&p, n, hipHostMallocPortable) != hipSuccess) p = nullptr;
        } else
            p = malloc(n);
        if (p) blocks.push_back({p, n, true});
        return p;
    }
    void put(void* p) {
        for (Block& b : blocks)
            if (b.p == p) b.in_use = false;
    }
} g_arenas;
void* arena_alloc(size_t n) { return g_arenas.get(n); }
void arena_free(void* p) { g_arenas.put(p); }
int default_parse_threads() { return std::min(16, lphq::usable_cpus()); }

// One pass of src/query.cpp:48-56 / :68-76 over the whole file, batched: returns false if the file cannot be opened.
// Every batch is split into contiguous runs of records of about equal base count, one per GPU (the index is replicated,
// include/lphq.h: lphq_attach_devices); per GPU: copy in (async, from the pinned parse arena), kernel, count, all on that GPU's
// stream.  The host only waits when it needs a slot or an arena back, so parsing of batch i+1, the copies of batch i and the
// kernels of batch i-1 overlap.
bool query_pass(lphq_index* idx, const lphq_info& info, const std::string& path, int streaming, size_t batch_bases, int threads, int gz_threads, bool pinned,
                std::vector<GpuWorker>& gpus, uint64_t& total_kmers, uint64_t& elapsed_us, std::string& err, FILE* sink) {
    g_arenas.pinned = pinned;
    lphq::FastxBatchReader rd(path.c_str(), threads, &arena_alloc, &arena_free, 1 << 16, gz_threads);
    if (!rd.ok()) return false;
    rd.prealloc(batch_bases);  // set-up, like the reference's gzopen + kseq_init before t.start() (src/query.cpp:43-48)
    if (!rd.ok()) {
        err = "out of host memory";
        return true;
    }
    // ... and the device buffers of both slots of every GPU for a whole batch (its share of it): allocation is set-up too
    for (GpuWorker& g : gpus) {
        const size_t share = batch_bases / gpus.size() + batch_bases / 8 + (1 << 16);
        if (hipSetDevice(g.device) != hipSuccess) continue;
        for (GpuWorker::Slot& s : g.slot)
            if (!s.busy) (void)GpuWorker::ensure(s, share, 1024, share, sink != nullptr);
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    lphq::FastxBatch bufs[2];
    total_kmers = 0;
    const size_t ng = gpus.size();
    const bool trace = getenv("LPHQ_CLI_TRACE") != nullptr;  // per-batch timeline on stderr (ms since the pass's timer started)
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count(); };
    // wait for slot `si` of every GPU: its count is in, its buffers are free; with --output its codes go to the sink (in GPU
    // order = record order)
    auto retire = [&](int si) -> bool {
        for (GpuWorker& g : gpus) {
            GpuWorker::Slot& s = g.slot[si];
            if (!s.busy) continue;
            (void)hipSetDevice(g.device);
            if (hipStreamSynchronize(g.stream) != hipSuccess) {
                err = "device work failed";
                return false;
            }
            s.busy = false;
            total_kmers += *s.h_count;
            if (sink && s.n_out) {  // result sink: the reference discards the codes (src/query.cpp:54); here they can be kept (raw little-endian u64)
                size_t nv = 0;
                for (size_t i = 0; i < s.n_out; ++i)
                    if (s.h_codes[i] != LPHQ_INVALID) s.h_codes[nv++] = s.h_codes[i];
                if (fwrite(s.h_codes, 8, nv, sink) != nv) {
                    err = "cannot write the output file";
                    return false;
                }
            }
        }
        return true;
    };
    // the bases of slot `si` have left the parse arena on every GPU
    auto arena_released = [&](int si) {
        for (GpuWorker& g : gpus)
            if (g.slot[si].busy) {
                (void)hipSetDevice(g.device);
                (void)hipEventSynchronize(g.slot[si].copied);
            }
    };
    auto flush = [&](const lphq::FastxBatch& b, int si) -> bool {
        const size_t n = b.lens.size();
        if (n == 0) return true;
        // record runs of about equal base count, one per GPU
        std::vector<size_t> cut(ng + 1, n);
        cut[0] = 0;
        {
            size_t r = 0;
            uint64_t acc = 0;
            for (size_t g = 1; g < ng; ++g) {
                const uint64_t want = (uint64_t)b.bases * g / ng;
                while (r < n && acc + b.lens[r] <= want) acc += b.lens[r++];
                cut[g] = r;
            }
        }
        size_t seg_i = 0, seg_o = 0;  // walk the segments in step with the records
        for (size_t gi = 0; gi < ng; ++gi) {
            GpuWorker& g = gpus[gi];
            GpuWorker::Slot& s = g.slot[si];
            const size_t r0 = cut[gi], r1 = cut[gi + 1], nr = r1 - r0;
            uint64_t so = 0, oo = 0;
            for (size_t i = r0; i < r1; ++i) {
                so += b.lens[i];
                oo += b.lens[i] >= info.k ? b.lens[i] - info.k + 1 : 0;
            }
            s.n_out = 0;
            if (nr == 0 || oo == 0) {  // nothing to look up: only move past this run's bases
                uint64_t left = so;
                while (left) {
                    const size_t take = (size_t)std::min<uint64_t>(left, b.segs[seg_i].n - seg_o);
                    seg_o += take;
                    left -= take;
                    if (seg_o == b.segs[seg_i].n) {
                        ++seg_i;
                        seg_o = 0;
                    }
                }
                continue;
            }
            if (hipSetDevice(g.device) != hipSuccess || !GpuWorker::ensure(s, so, nr, oo, sink != nullptr)) {
                err = "device allocation failed";
                return false;
            }
            uint64_t a = 0, o = 0;
            for (size_t i = 0; i < nr; ++i) {
                s.h_offs[i] = a;
                s.h_offs[nr + 1 + i] = o;
                a += b.lens[r0 + i];
                o += b.lens[r0 + i] >= info.k ? b.lens[r0 + i] - info.k + 1 : 0;
            }
            s.h_offs[nr] = a;
            s.h_offs[2 * nr + 1] = o;
            uint64_t left = so;
            size_t at = 0;
            if (trace) fprintf(stderr, "[cli]   flush: buffers ready at %.2f ms\n", ms());
            while (left) {  // one copy per parsed piece, compacting on the way
                const size_t take = (size_t)std::min<uint64_t>(left, b.segs[seg_i].n - seg_o);
                if (take && hipMemcpyAsync(s.d_seq + at, b.segs[seg_i].p + seg_o, take, hipMemcpyHostToDevice, g.stream) != hipSuccess) {
                    err = "host to device copy failed";
                    return false;
                }
                at += take;
                seg_o += take;
                left -= take;
                if (seg_o == b.segs[seg_i].n) {
                    ++seg_i;
                    seg_o = 0;
                }
            }
            if (hipEventRecord(s.copied, g.stream) != hipSuccess ||
                hipMemcpyAsync(s.d_offs, s.h_offs, 2 * (nr + 1) * 8, hipMemcpyHostToDevice, g.stream) != hipSuccess) {
                err = "host to device copy failed";
                return false;
            }
            s.busy = true;
            if (trace) fprintf(stderr, "[cli]   flush: copies queued at %.2f ms\n", ms());
            if (lphq_query_records(idx, s.d_seq, so, s.d_offs, nr, s.d_out, s.d_offs + nr + 1, streaming, g.stream) != LPHQ_OK) {
                err = lphq_last_error();
                return false;
            }
            if (trace) fprintf(stderr, "[cli]   flush: kernel queued at %.2f ms\n", ms());
            // hashes.size() of the reference: counted on the device, read back when the slot is retired
            s.n_out = oo;
            if (lphq_count_valid_async(idx, s.d_out, oo, s.d_count, g.stream) != LPHQ_OK) {
                err = lphq_last_error();
                return false;
            }
            if (hipMemcpyAsync(s.h_count, s.d_count, 8, hipMemcpyDeviceToHost, g.stream) != hipSuccess ||
                (sink && hipMemcpyAsync(s.h_codes, s.d_out, oo * 8, hipMemcpyDeviceToHost, g.stream) != hipSuccess)) {
                err = "device to host copy failed";
                return false;
            }
        }
        return true;
    };
    // batch i+1 is parsed (by `threads` threads) while batch i is copied to the devices and looked up; the reader keeps
    // both batches' bases alive (fastx_batch.hpp), and bufs[i] may be parsed into again once slot i's copies are done
    auto parse = [&](int i) {
        return std::async(std::launch::async, [&rd, &bufs, i, batch_bases, trace, &ms] {
            const double a = trace ? ms() : 0;
            const bool r = rd.next(bufs[i], batch_bases);
            if (trace) fprintf(stderr, "[cli] next(): %.2f -> %.2f ms, %zu bases in %zu records, %zu pieces\n", a, ms(), bufs[i].bases, bufs[i].lens.size(), bufs[i].segs.size());
            return r;
        });
    };
    std::future<bool> pending = parse(0);
    bool ok = true;
    int last = 1;  // slot of the most recent batch
    for (int i = 0; ok; i ^= 1) {
        if (!pending.get()) break;
        arena_released(i ^ 1);
        pending = parse(i ^ 1);
        const double a = trace ? ms() : 0;
        ok = retire(i);
        const double b = trace ? ms() : 0;
        ok = ok && flush(bufs[i], i);
        if (trace) fprintf(stderr, "[cli] batch of %zu bases: retire %.2f -> %.2f, queued at %.2f ms\n", bufs[i].bases, a, b, ms());
        last = i;
        if (!ok) pending.wait();
    }
    ok = ok && retire(last ^ 1) && retire(last);  // the older batch first: the sink keeps record order
    if (trace) fprintf(stderr, "[cli] all retired at %.2f ms\n", ms());
    if (ok && !rd.ok()) err = "out of host memory while reading the query file";
    auto t1 = std::chrono::high_resolution_clock::now();
    elapsed_us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();
    return true;
}

// diagnostic: the decompressed bytes of a .gz as the serial reader gets them (gz_source.hpp; --zlib: through gzread instead), to a file or nowhere
int inflate_main(int argc, char** argv) {
    Args a = parse(argc, argv, {"-i", "-o", "--parallel", "--span"}, {"--zlib"});
    if (a.help || a.bad || !a.get("-i")) {
        std::cerr << "Usage: " << argv[0] << " -i file.gz [-o output] [--zlib | --parallel threads [--span bytes]]\n";
        return 1;
    }
    FILE* out = a.get("-o") ? fopen(a.get("-o")->c_str(), "wb") : nullptr;
    if (a.get("-o") && !out) return 2;
    std::vector<char> buf(4u << 20);
    uint64_t total = 0;
    bool error = false;
    const auto t0 = std::chrono::high_resolution_clock::now();
    if (a.has("--zlib")) {
        gzFile fp = gzopen(a.get("-i")->c_str(), "r");
        if (!fp) return 2;
        gzbuffer(fp, 1 << 20);
        int n;
        while ((n = gzread(fp, buf.data(), (unsigned)buf.size())) > 0) {
            total += (uint64_t)n;
            if (out) fwrite(buf.data(), 1, (size_t)n, out);
        }
        error = n < 0;
        gzclose(fp);
    } else if (a.get("--parallel")) {
        // parallel_gz.hpp: every piece into one buffer that is written out whenever it holds 64 MiB; prints pieces taken from the pool / decoded serially
        lphq::ParallelGzReader* src = lphq::ParallelGzReader::open(a.get("-i")->c_str(), std::atoi(a.get("--parallel")->c_str()),
                                                                   a.get("--span") ? (size_t)std::atol(a.get("--span")->c_str()) : 0);
        if (!src) {
            std::cerr << "not a regular gzip file, or too small for its spans\n";
            return 2;
        }
        src->prealloc();
        std::vector<uint8_t> big((size_t)256 << 20, 1);  // (allocated and touched before the timer, like the reader's arenas)
        const auto t1 = std::chrono::high_resolution_clock::now();
        size_t used = 0;
        auto drain = [&] {
            src->flush();
            if (out && used) fwrite(big.data(), 1, used, out);
            total += used;
            used = 0;
        };
        size_t n;
        while ((n = src->next_len()) > 0) {
            if (used + n > big.size()) {
                drain();
                if (n > big.size()) big.resize(n);
            }
            src->emit(big.data() + used);
            used += n;
        }
        drain();
        error = src->error();
        const double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t1).count();
        if (out) fclose(out);
        std::cout << total << "," << (error ? "error" : "ok") << "," << sec << "," << src->pieces_parallel() << "," << src->pieces_serial() << "," << src->span() << "\n";
        delete src;
        return error ? 3 : 0;
    } else {
        lphq::GzFastSource* src = lphq::GzFastSource::open(a.get("-i")->c_str());
        if (!src) {
            std::cerr << "not a regular gzip file\n";
            return 2;
        }
        size_t n;
        while ((n = src->read(buf.data(), buf.size())) > 0) {
            total += n;
            if (out) fwrite(buf.data(), 1, n, out);
        }
        error = src->error();
        delete src;
    }
    const double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    if (out) fclose(out);
    std::cout << total << "," << (error ? "error" : "ok") << "," << sec << "\n";
    return error ? 3 : 0;
}

// diagnostic: what the reader yields (see the file header)
int records_main(int argc, char** argv) {
    Args a = parse(argc, argv, {"-i", "-t", "--batch-bytes", "--min-piece", "--passes"}, {"--serial", "--no-hash"});
    if (a.help || a.bad || !a.get("-i")) {
        std::cerr << "Usage: " << argv[0] << " -i fastx [-t threads] [--serial] [--no-hash] [--batch-bytes N] [--min-piece N] [--passes N]\n";
        return 1;
    }
    const int threads = a.get("-t") ? std::atoi(a.get("-t")->c_str()) : default_parse_threads();
    const size_t batch = a.get("--batch-bytes") ? (size_t)std::strtoull(a.get("--batch-bytes")->c_str(), nullptr, 10) : (size_t)512 << 20;
    const size_t min_piece = a.get("--min-piece") ? (size_t)std::strtoull(a.get("--min-piece")->c_str(), nullptr, 10) : (size_t)1 << 16;
    const int passes = a.get("--passes") ? std::max(1, std::atoi(a.get("--passes")->c_str())) : 1;  // later passes reuse the (warm) parse arenas
    uint64_t n_rec = 0, n_bases = 0, h = 0, n_batches = 0, n_parallel = 0, window = 0, n_pgz = 0;
    std::chrono::high_resolution_clock::time_point t0;
    for (int pass = 0; pass < passes; ++pass) {
    n_rec = n_bases = n_batches = n_parallel = n_pgz = 0;
    h = 1469598103934665603ull;  // FNV-1a over (length, bases) of every record
    const bool no_hash = a.has("--no-hash");  // timing of the reader alone
    auto mix = [&](const void* p, size_t n) {
        if (no_hash) return;
        const unsigned char* c = (const unsigned char*)p;
        for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 1099511628211ull;
    };
    t0 = std::chrono::high_resolution_clock::now();
    if (a.has("--serial")) {
        lphq::FastxReader rd(a.get("-i")->c_str());
        if (!rd.ok()) return 2;
        std::string seq;
        while (rd.next(seq)) {
            const uint64_t len = seq.size();
            mix(&len, 8);
            mix(seq.data(), seq.size());
            ++n_rec;
            n_bases += len;
        }
    } else {
        lphq::FastxBatchReader rd(a.get("-i")->c_str(), threads, &arena_alloc, &arena_free, min_piece);
        if (!rd.ok()) return 2;
        lphq::FastxBatch b;
        while (rd.next(b, batch)) {
            ++n_batches;
            n_parallel += rd.parallel();
            n_pgz += rd.parallel_gz() && rd.parallel();  // ... of which: text that came from the span-parallel .gz decoder
            size_t si = 0, so = 0;  // walk the segments record by record
            for (uint64_t len : b.lens) {
                mix(&len, 8);
                uint64_t left = len;
                while (left) {
                    while (si < b.segs.size() && so == b.segs[si].n) {
                        ++si;
                        so = 0;
                    }
                    if (si == b.segs.size()) return 5;  // lengths and segments disagree
                    const size_t take = (size_t)std::min<uint64_t>(left, b.segs[si].n - so);
                    mix(b.segs[si].p + so, take);
                    so += take;
                    left -= take;
                }
                ++n_rec;
                n_bases += len;
            }
        }
        if (!rd.ok()) return 3;
        window = rd.window_capacity();  // BGZF and other .gz: bytes of inflated text held at once (0 for every other kind of input)
    }
    }
    auto t1 = std::chrono::high_resolution_clock::now();
    const double sec = std::chrono::duration<double>(t1 - t0).count();
    std::cout << n_rec << "," << n_bases << "," << h << "," << n_batches << "," << n_parallel << "," << sec << "," << window << "," << n_pgz << std::endl;
    return 0;
}

// --device D: that one device (default 0); --gpus N: devices 0..N-1; --devices a,b,..: the listed devices, one worker (stream, arenas, buffer
// slots) per entry -- an entry may repeat, which puts several workers on one GPU (the replica is shared): that is how the N-worker sharding
// is tested on a machine with one GPU.  Every parsed batch is sharded over the workers.
bool device_list(const Args& a, int device, std::vector<int>& devices) {
    const int given = (a.get("--gpus") ? 1 : 0) + (a.get("--device") ? 1 : 0) + (a.get("--devices") ? 1 : 0);
    if (given > 1) {
        std::cerr << "--device, --gpus and --devices exclude one another\n";
        return false;
    }
    if (a.get("--devices")) {
        const std::string& l = *a.get("--devices");
        size_t i = 0;
        while (i <= l.size()) {
            const size_t j = std::min(l.find(',', i), l.size());
            if (j == i || l.find_first_not_of("0123456789", i) < j) {
                std::cerr << "--devices wants a comma-separated list of device numbers\n";
                return false;
            }
            devices.push_back(std::atoi(l.substr(i, j - i).c_str()));
            i = j + 1;
        }
        return true;
    }
    const int ngpus = a.get("--gpus") ? std::atoi(a.get("--gpus")->c_str()) : 1;
    if (ngpus < 1) {
        std::cerr << "--gpus N needs N >= 1\n";
        return false;
    }
    for (int d = 0; d < ngpus; ++d) devices.push_back(a.get("--gpus") ? d : device);
    return true;
}

int query_main(int kind, int argc, char** argv) {
    Args a = parse(argc, argv, {"-i", "-q", "--kmer-bits", "--device", "--gpus", "--devices", "--batch-mb", "--output", "--threads"}, {"--pinned", "--pageable"});
    if (a.help || a.bad || !a.get("-i") || !a.get("-q")) {
        std::cerr << "Usage: " << argv[0]
                  << " [-h,--help] -i mphf -q query_filename [--kmer-bits 64|128] [--device N | --gpus N | --devices a,b,..] [--batch-mb N] [--threads N] [--pageable] [--output codes.u64]\n\n"
                  << " -i mphf\n\tREQUIRED: LP-Hash MPHF saved on disk.\n\n"
                  << " -q query_filename\n\tREQUIRED: Must be a FASTA file (.fa/fasta extension) compressed with gzip (.gz) or not.\n\n"
                  << " --kmer-bits\n\tk-mer word width the MPHF was built with (the reference's compile-time kmer_t): 128 (default) or 64.\n";
        return 1;  // ParseError -> 1, src/query.cpp:31-33
    }
    const std::string mphf_filename = *a.get("-i"), query_filename = *a.get("-q");
    const int kmer_bits = a.get("--kmer-bits") ? std::atoi(a.get("--kmer-bits")->c_str()) : 128;
    const int device = a.get("--device") ? std::atoi(a.get("--device")->c_str()) : 0;
    // 32 MB batches: the parse arenas stay small and warm, and parsing overlaps the device side (see query_pass)
    const size_t batch_bases = (size_t)(a.get("--batch-mb") ? std::max(1l, std::atol(a.get("--batch-mb")->c_str())) : 32) << 20;
    const int threads = a.get("--threads") ? std::max(1, std::atoi(a.get("--threads")->c_str())) : default_parse_threads();
    const int gz_threads = a.get("--threads") ? threads : lphq::usable_cpus();  // (the .gz decoder scales further than the parser: fastx_batch.hpp)
    const bool pinned = !a.has("--pageable");  // parse arenas in pinned host memory (default): copies to the devices are asynchronous
    // --gpus N: replicas on devices 0..N-1, every batch sharded over them; --device D: that one device
    std::vector<int> devices;
    if (!device_list(a, device, devices)) return 1;
    lphq_index* idx = nullptr;
    if (lphq_open(mphf_filename.c_str(), kind, kmer_bits, &idx) != LPHQ_OK || lphq_attach_devices(idx, devices.data(), (int)devices.size()) != LPHQ_OK) {
        std::cerr << "[Error] " << lphq_last_error() << "\n";  // the reference dies in essentials::load here
        if (idx) lphq_close(idx);
        return 2;
    }
    lphq_info info;
    lphq_get_info(idx, &info);
    std::vector<GpuWorker> gpus(devices.size());
    for (size_t d = 0; d < devices.size(); ++d)
        if (!gpus[d].init(devices[d])) {
            std::cerr << "[Error] cannot create a stream on device " << devices[d] << "\n";
            lphq_close(idx);
            return 2;
        }
    auto release_gpus = [&gpus] {
        for (GpuWorker& g : gpus) g.destroy();
    };
    // set-up outside the timers: the first launch of a kernel loads the code object (tens of ms); do it on a dummy record
    {
        const std::string dummy(info.k, 'A');
        std::vector<uint64_t> one(1);
        size_t n1 = 0;
        for (size_t d = 0; d < devices.size(); ++d) {
            (void)hipSetDevice(devices[d]);
            (void)lphq_query_contig(idx, dummy.data(), dummy.size(), 1, one.data(), 1, &n1);
            (void)lphq_query_contig(idx, dummy.data(), dummy.size(), 0, one.data(), 1, &n1);
        }
    }
    uint64_t total_on = 0, total_off = 0, t_on = 0, t_off = 0;
    std::string err;
    FILE* sink = nullptr;
    if (a.get("--output") && !(sink = fopen(a.get("--output")->c_str(), "wb"))) {
        std::cerr << "Unable to open the output file " << *a.get("--output") << "\n";
        lphq_close(idx);
        return 2;
    }
    for (int pass = 0; pass < 2; ++pass) {
        bool opened = query_pass(idx, info, query_filename, pass == 0, batch_bases, threads, gz_threads, pinned, gpus, pass == 0 ? total_on : total_off,
                                 pass == 0 ? t_on : t_off, err, pass == 0 ? sink : nullptr);
        if (!opened) {
            std::cerr << "Unable to open the input file " << query_filename << "\n";
            release_gpus();
            lphq_close(idx);
            return 2;  // src/query.cpp:43-46
        }
        if (!err.empty()) {
            std::cerr << "[Error] " << err << "\n";
            release_gpus();
            lphq_close(idx);
            return 3;
        }
    }
    if (sink) fclose(sink);
    // src/query.cpp:83-86
    std::cout << query_filename << "," << mphf_filename << "," << total_on << "," << static_cast<double>(t_on * 1000) / total_on << ","
              << static_cast<double>(t_off * 1000) / total_off << std::endl;
    release_gpus();
    lphq_close(idx);
    return 0;
}

// one pointer and length per record of a batch (the batch is slid together first: pieces of the parallel reader may split a record)
void batch_pointers(lphq::FastxBatch& b, std::vector<const char*>& ptrs, std::vector<size_t>& lens) {
    b.compact();
    const size_t n_recs = b.lens.size();
    ptrs.resize(n_recs);
    lens.resize(n_recs);
    size_t si = 0, so = 0;
    for (size_t i = 0; i < n_recs; ++i) {
        while (si < b.segs.size() && so == b.segs[si].n) {
            ++si;
            so = 0;
        }
        lens[i] = (size_t)b.lens[i];
        ptrs[i] = si < b.segs.size() ? b.segs[si].p + so : "";
        so += lens[i];
    }
}

// lphq_record_source over a FASTA / FASTQ file: batches of about batch_bases bases, the file reopened for every pass (the reference opens it
// twice as well, src/partitioned_mphf.cpp:66 and :118)
struct FileRecordSource {
    std::string path;
    int threads = 1;
    size_t batch_bases = ~(size_t)0;
    int pass = -1;
    std::unique_ptr<lphq::FastxBatchReader> rd;
    lphq::FastxBatch batch;
    std::vector<const char*> ptrs;
    std::vector<size_t> lens;
    // the next batch of `pass` into ptrs / lens: 1, 0 at the end of the file, -1 on error
    int advance(int want_pass) {
        if (pass != want_pass) {
            rd.reset(new lphq::FastxBatchReader(path.c_str(), std::max(1, threads)));
            pass = want_pass;
        }
        if (!rd->ok()) return -1;
        if (!rd->next(batch, batch_bases)) return rd->ok() ? 0 : -1;
        if (!rd->ok()) return -1;
        batch_pointers(batch, ptrs, lens);
        return 1;
    }
    static int next(void* ctx, int pass, const char* const** records, const size_t** lengths, size_t* n_records) {
        FileRecordSource* s = (FileRecordSource*)ctx;
        const int rc = s->advance(pass);
        if (rc <= 0) return rc;
        *records = s->ptrs.data();
        *lengths = s->lens.data();
        *n_records = s->ptrs.size();
        return 1;
    }
};

int build_main(int kind, int argc, char** argv) {
    Args a = parse(argc, argv, {"-i", "-k", "-m", "-s", "-t", "-o", "-d", "-c", "--max-memory", "--max-memory-mb", "--kmer-bits", "--device"},
                   {"--check", "--verbose", "--cpu"});
    if (a.help || a.bad || !a.get("-i") || !a.get("-k") || !a.get("-m")) {
        std::cerr << "Usage: " << argv[0]
                  << " [-h,--help] -i input_filename -k k -m m [-s seed] [-t threads] [-o output_filename] [-d tmp_dirname] [-c c] "
                     "[--max-memory GB | --max-memory-mb MB] [--check] [--verbose] [--kmer-bits 64|128] [--device N] [--cpu]\n";
        if (!a.help) std::cerr << "Unable to parse the arguments\n";
        return 2;  // ParseError -> 2, src/build.cpp:14-16
    }
    const std::string input = *a.get("-i");
    const int k = std::atoi(a.get("-k")->c_str()), m = std::atoi(a.get("-m")->c_str());
    const int kmer_bits = a.get("--kmer-bits") ? std::atoi(a.get("--kmer-bits")->c_str()) : 128;
    const uint64_t seed = a.get("-s") ? std::strtoull(a.get("-s")->c_str(), nullptr, 10) : 42;  // constants::default_seed
    const int threads = a.get("-t") ? std::atoi(a.get("-t")->c_str()) : 1;                      // constants::default_num_threads
    const double c = a.get("-c") ? std::atof(a.get("-c")->c_str()) : 3.0;                       // constants::c
    const bool verbose = a.has("--verbose"), check = a.has("--check");
    // OptionError -> 3, src/build.cpp:17-20, parser_build.cpp:64-88
    if (kmer_bits != 64 && kmer_bits != 128) {
        std::cerr << "--kmer-bits must be 64 or 128" << std::endl;
        return 3;
    }
    if (k > kmer_bits / 2 - 1) {
        std::cerr << "k cannot be larger than " << kmer_bits / 2 - 1 << std::endl;
        return 3;
    }
    if (m > k) {
        std::cerr << "m cannot be larger than k" << std::endl;
        return 3;
    }
    // src/parser_build.cpp:74-86: -d creates the directory, --max-memory is capped at 255 (GB).  With a limit the input is streamed in
    // batches and every vector of the construction lives in sorted pieces under -d (lphq_build_stream: the reference's external_memory_vector
    // and build_in_external_memory);
    // without one (the reference would assume 8 GB) the input is read whole and nothing touches the disk.  --max-memory-mb: the same in MB.
    const std::string tmp_dir = a.get("-d") ? *a.get("-d") : ".";  // constants::default_tmp_dirname
    if (a.get("-d")) (void)::mkdir(tmp_dir.c_str(), 0777);         // essentials::create_directory
    uint64_t max_memory = 0;
    if (a.get("--max-memory")) {
        const uint64_t gb = std::strtoull(a.get("--max-memory")->c_str(), nullptr, 10);
        if (gb > 255) {
            std::cerr << "The maximum allowed amount of ram is 255GB" << std::endl;
            return 3;
        }
        max_memory = gb << 30;
    }
    if (a.get("--max-memory-mb")) max_memory = std::strtoull(a.get("--max-memory-mb")->c_str(), nullptr, 10) << 20;
    const bool streamed = max_memory != 0;
    if (c > 10.0 || c < 3.0) {
        std::cerr << "3.0 <= c <= 10.0" << std::endl;
        return 3;
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    FileRecordSource file;
    file.path = input;
    file.threads = threads;
    // a streamed batch: an eighth of the limit (the reader keeps two arenas), within [1 MiB, 256 MiB]
    if (streamed) file.batch_bases = (size_t)std::min<uint64_t>(256u << 20, std::max<uint64_t>(1u << 20, max_memory / 8));
    std::vector<const char*> ptrs;  // in-memory construction: the whole input as one batch, valid while file.rd lives
    std::vector<size_t> lens;
    {
        lphq::FastxBatchReader probe(input.c_str(), 1);
        if (!probe.ok()) {
            std::cerr << "Unable to open the input file " << input << "\n";
            return 4;  // the reference throws std::runtime_error here (partitioned_mphf.cpp:66-67)
        }
    }
    if (!streamed) {
        if (file.advance(0) < 0) {
            std::cerr << "[Error] out of host memory while reading " << input << "\n";
            return 4;
        }
        ptrs = file.ptrs;
        lens = file.lens;
    }
    const size_t n_recs = ptrs.size();
    std::string out = a.get("-o") ? *a.get("-o") : "";
    const bool temp_out = out.empty();
    if (temp_out) out = "/tmp/lphash_build_" + std::to_string((long)getpid()) + ".lph";
    lphq_build_stats st;
    if (verbose) std::cerr << "Part 1-4: super-k-mers, minimizer MPHF, inverted index, fallback MPHF\n";
    // construction is not the query path: the pilot search runs on the GPU when one is usable, else on the CPU (same file)
    int build_device = a.has("--cpu") ? -1 : (a.get("--device") ? std::atoi(a.get("--device")->c_str()) : 0);
    if (build_device >= 0) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || build_device >= ndev) build_device = -1;
    }
    if (verbose) std::cerr << "PTHash pilot search on " << (build_device >= 0 ? "GPU " + std::to_string(build_device) : std::string("the CPU")) << "\n";
    if (verbose && streamed) std::cerr << "Memory limit " << (max_memory >> 20) << " MB: input in batches of " << (file.batch_bases >> 20) << " Mbases, temporary files in " << tmp_dir << "\n";
    const int brc = streamed ? lphq_build_stream(&FileRecordSource::next, &file, k, m, seed, c, kind, kmer_bits, threads, build_device, tmp_dir.c_str(), max_memory,
                                                 out.c_str(), &st)
                             : lphq_build_device(ptrs.data(), lens.data(), n_recs, k, m, seed, c, kind, kmer_bits, threads, build_device, out.c_str(), &st);
    if (brc != LPHQ_OK) {
        std::cerr << "[Error] " << lphq_last_error() << "\n";
        if (temp_out) std::remove(out.c_str());
        return 4;
    }
    // partitioned_mphf.cpp:138-144 (the last field is the serialized size: the reference's num_bits() adds
    // sizeof(pthash::build_configuration), which has no counterpart here)
    const uint64_t total_contigs = st.num_records > 0 ? st.num_records - 1 : 0;
    std::cout << input << "," << k << "," << m << "," << static_cast<double>(st.colliding_super_kmers) / st.distinct_minimizers << ","
              << 2.0 / ((k - m + 1) + 1) << "," << static_cast<double>(st.num_super_kmers) / st.num_kmers << ","
              << static_cast<double>(total_contigs) / st.num_kmers << "," << static_cast<double>(st.file_bytes * 8) / st.num_kmers << "\n";
    auto t1 = std::chrono::high_resolution_clock::now();
    std::cerr << "function built in " << std::chrono::duration_cast<std::chrono::seconds>(t1 - t0).count() << " [sec]" << std::endl;
    int rc = 0;
    if (check) {  // src/build.cpp:53-70 on the GPU path: collisions/range, streaming == random, perfection
        std::cerr << "Checking...\n";
        const int device = a.get("--device") ? std::atoi(a.get("--device")->c_str()) : 0;
        lphq_index* idx = nullptr;
        if (lphq_open(out.c_str(), kind, kmer_bits, &idx) != LPHQ_OK || lphq_attach(idx, device) != LPHQ_OK) {
            std::cerr << "[Error] " << lphq_last_error() << "\n";
            if (idx) lphq_close(idx);
            return 4;
        }
        lphq_info info;
        lphq_get_info(idx, &info);
        std::cerr << "[Info] Loaded " << info.file_bytes * 8 << " bits\n";
        std::vector<uint64_t> population((st.num_kmers + 63) / 64, 0);
        bool good = true;
        std::vector<uint64_t> a_codes, b_codes;
        auto check_records = [&](const std::vector<const char*>& ptrs, const std::vector<size_t>& lens) {
            for (size_t i = 0; i < ptrs.size() && good; ++i) {
                if (lens[i] < (size_t)k) continue;
                const size_t nw = lens[i] - k + 1;
                a_codes.resize(nw);
                b_codes.resize(nw);
                size_t na = 0, nb = 0;
                if (lphq_query_contig(idx, ptrs[i], lens[i], 0, a_codes.data(), nw, &na) != LPHQ_OK ||
                    lphq_query_contig(idx, ptrs[i], lens[i], 1, b_codes.data(), nw, &nb) != LPHQ_OK) {
                    std::cerr << "[Error] " << lphq_last_error() << "\n";
                    good = false;
                    break;
                }
                for (size_t j = 0; j < na && good; ++j) {
                    const uint64_t hsh = a_codes[j];
                    if (hsh >= st.num_kmers) {
                        std::cerr << "[Error] overflow : " << hsh << " > " << st.num_kmers << std::endl;
                        good = false;
                    } else if ((population[hsh >> 6] >> (hsh & 63)) & 1) {
                        std::cerr << "[Error] collision at position (hash) : " << hsh << std::endl;
                        good = false;
                    } else
                        population[hsh >> 6] |= 1ULL << (hsh & 63);
                }
                if (good && (na != nb || std::memcmp(a_codes.data(), b_codes.data(), na * 8) != 0)) {
                    std::cerr << "[Error] different hashes, maybe there were some Ns in the input (not supported as of now)\n";
                    good = false;
                }
            }
        };
        if (!streamed) {
            check_records(ptrs, lens);
        } else {  // one more pass over the file, a batch at a time
            int src_rc;
            while (good && (src_rc = file.advance(2)) > 0) check_records(file.ptrs, file.lens);
            if (good && src_rc < 0) {
                std::cerr << "[Error] cannot re-read " << input << "\n";
                good = false;
            }
        }
        if (good) {
            bool perfect = true;
            for (uint64_t i = 0; i < st.num_kmers; ++i)
                if (!((population[i >> 6] >> (i & 63)) & 1)) perfect = false;
            if (!perfect) {
                std::cerr << "[Error] Not all k-mers have been marked by a hash" << std::endl;
                rc = 5;
            } else
                std::cerr << "[Info] Everything is ok\n";
        } else
            rc = 5;
        lphq_close(idx);
    }
    if (verbose) {  // src/build.cpp:45-48 -> print_statistics(), src/partitioned_mphf.cpp:341-380 / src/unpartitioned_mphf.cpp:208-240
        std::cerr << "Statistics:\n";
        lphq_index* sidx = nullptr;
        lphq_info si;
        if (lphq_open(out.c_str(), kind, kmer_bits, &sidx) == LPHQ_OK && lphq_get_info(sidx, &si) == LPHQ_OK) {
            // Sizes are those of the serialized parts; the reference's num_bits() walks the in-memory objects and adds
            // sizeof(pthash::build_configuration), so its totals differ by a constant of a few hundred bits.
            const double total = (double)si.file_bytes * 8;
            auto pct = [&](uint64_t b) { return (double)b / total * 100; };
            std::cerr << "Total number of k-mers: " << si.num_kmers << "\n";
            std::cerr << "Total number of k-mers belonging to ambiguous minimizers: " << si.fallback_kmers << "\n";
            std::cerr << "xi = " << static_cast<double>(si.fallback_kmers) / si.num_kmers << "\n";
            std::cerr << "Minimizer MPHF size in bits : " << si.part_bits[0] << " (" << pct(si.part_bits[0]) << "%)\n";
            std::cerr << "\t = " << static_cast<double>(si.part_bits[0]) / si.distinct_minimizers << " bits/minimizer\n\n";
            if (kind == LPHQ_PARTITIONED) {
                std::cerr << "Wavelet tree size in bits : " << si.part_bits[1] << " (" << pct(si.part_bits[1]) << "%)\n";
                std::cerr << "\t = " << static_cast<double>(si.part_bits[1]) / si.distinct_minimizers << " bits/minimizer\n\n";
            }
            std::cerr << "Compressed arrays (EF) : " << si.part_bits[2] << " (" << pct(si.part_bits[2]) << "%)\n";
            std::cerr << "\t = " << static_cast<double>(si.part_bits[2]) / si.ef_entries << " bits/offset\n\n";
            std::cerr << "Fallback MPHF : " << si.part_bits[3] << " (" << pct(si.part_bits[3]) << "%)\n";
            std::cerr << "\t = " << static_cast<double>(si.part_bits[3]) / si.fallback_kmers << " bits/kmer\n\n";
            std::cerr << "Total size in bits : " << si.file_bytes * 8 << "\n";
            std::cerr << "\tequivalent to : " << static_cast<double>(si.file_bytes * 8) / si.num_kmers << " bits/k-mer\n";
            std::cerr << "\n";
        } else {
            std::cerr << "[Error] " << lphq_last_error() << "\n";
        }
        if (sidx) lphq_close(sidx);
    }
    if (temp_out) std::remove(out.c_str());
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Competitor harness, PTHash half: `ptbb_build` / `ptbb_query` of the reference (src/ptbb/ptbb_build.cpp,
// src/ptbb/ptbb_query.cpp) as sub-commands: same flags, same CSV.  The PTHash file is the reference's (a bare essentials::save of
// pthash::single_phf), the BBHash file is boomphf::mphf::save's.
// ---------------------------------------------------------------------------------------------------------------------------
int ptbb_build_main(int argc, char** argv) {
    Args a = parse(argc, argv, {"-i", "-k", "-p", "-b", "-a", "-c", "-g", "-d", "-t", "--kmer-bits", "--device"}, {"--verbose", "--check", "--cpu"});
    if (a.help || a.bad || !a.get("-i") || !a.get("-k")) {
        std::cerr << "Usage: " << argv[0] << " [-h,--help] -i input_filename -k k [-p pthash_filename] [-c c] [-d tmp_dirname] [-t threads] [--verbose] [--check] "
                     "[--kmer-bits 64|128] [--device N] [--cpu]\n";
        return 1;  // !parser.parse() -> 1, ptbb_build.cpp:42
    }
    const std::string input = *a.get("-i");
    const int k = std::atoi(a.get("-k")->c_str());
    const int kmer_bits = a.get("--kmer-bits") ? std::atoi(a.get("--kmer-bits")->c_str()) : 128;
    const double c = a.get("-c") ? std::atof(a.get("-c")->c_str()) : 3.0;  // constants::c
    const double gamma = a.get("-g") ? std::atof(a.get("-g")->c_str()) : 1.0;  // ptbb_build.cpp:143-148
    if (a.get("-b") && gamma < 1.0) {
        std::cerr << "BBHash gamma factor < 1\n";  // the reference throws std::runtime_error here (ptbb_build.cpp:149)
        return 3;
    }
    const int threads = a.get("-t") ? std::atoi(a.get("-t")->c_str()) : 1;
    if ((kmer_bits != 64 && kmer_bits != 128) || k < 1 || k > kmer_bits / 2 - 1) {
        std::cerr << "k cannot be larger than " << kmer_bits / 2 - 1 << std::endl;
        return 3;
    }
    if (a.get("-d")) (void)::mkdir(a.get("-d")->c_str(), 0777);  // essentials::create_directory; construction here is in memory
    lphq::FastxBatchReader rd(input.c_str(), std::max(1, threads));
    if (!rd.ok()) {
        std::cerr << "Unable to open the input file " << input << "\n";
        return 2;  // ptbb_build.cpp:48-51
    }
    lphq::FastxBatch whole;
    (void)rd.next(whole, ~(size_t)0);
    if (!rd.ok()) {
        std::cerr << "[Error] out of host memory while reading " << input << "\n";
        return 4;
    }
    std::vector<const char*> ptrs;
    std::vector<size_t> lens;
    batch_pointers(whole, ptrs, lens);
    const size_t n_recs = ptrs.size();
    if (!a.get("-p") && !a.get("-b")) {  // nothing to build: the k-mer count alone (ptbb_build.cpp:52-68), all other fields empty
        uint64_t total = 0;
        for (size_t i = 0; i < n_recs; ++i) {
            uint64_t since = 0;
            for (size_t j = 0; j < lens[i]; ++j) {
                switch (ptrs[i][j]) {
                    case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': case 'U': case 'u': total += ++since >= (uint64_t)k; break;
                    default: since = 0;
                }
            }
        }
        std::cout << input << "," << k << "," << total << ",,,," << std::endl;
        return 0;
    }
    // bijection check of ptbb_build.cpp:103-131 / :165-193 through the GPU path, for either function
    auto check_bijection = [&](lphq_index* idx, uint64_t nkeys, const char* what) -> int {
        std::cerr << "Checking " << what << "...";
        const int device = a.get("--device") ? std::atoi(a.get("--device")->c_str()) : 0;
        if (!idx || lphq_attach(idx, device) != LPHQ_OK) {
            std::cerr << "[Error] " << lphq_last_error() << "\n";
            return 4;
        }
        std::vector<uint64_t> population((nkeys + 63) / 64, 0), codes;
        uint64_t seen = 0;
        for (size_t i = 0; i < n_recs; ++i) {
            if (lens[i] < (size_t)k) continue;
            codes.resize(lens[i] - k + 1);
            size_t n = 0;
            if (lphq_query_contig(idx, ptrs[i], lens[i], 1, codes.data(), codes.size(), &n) != LPHQ_OK) {
                std::cerr << "[Error] " << lphq_last_error() << "\n";
                return 4;
            }
            for (size_t j = 0; j < n; ++j) {
                const uint64_t x = codes[j];
                if (x >= nkeys) {
                    std::cerr << "[Error] out of bounds" << std::endl;
                    return 2;
                }
                if ((population[x >> 6] >> (x & 63)) & 1) {
                    std::cerr << "[Error] collision" << std::endl;
                    return 2;
                }
                population[x >> 6] |= 1ULL << (x & 63);
                ++seen;
            }
        }
        if (seen != nkeys) {
            std::cerr << "[Error] hash is not perfect" << std::endl;
            return 2;
        }
        std::cerr << "EVERYTHING OK\n";
        return 0;
    };
    uint64_t total_kmers = 0;
    std::string pt_fields = ",,", bb_fields = ",,";
    int rc = 0;
    if (a.get("-p")) {
        const std::string out = *a.get("-p");
        int build_device = a.has("--cpu") ? -1 : (a.get("--device") ? std::atoi(a.get("--device")->c_str()) : 0);
        if (build_device >= 0) {
            int ndev = 0;
            if (hipGetDeviceCount(&ndev) != hipSuccess || build_device >= ndev) build_device = -1;
        }
        lphq_build_stats st;
        if (lphq_build_pthash(ptrs.data(), lens.data(), n_recs, k, kmer_bits, c, threads, build_device, out.c_str(), &st) != LPHQ_OK) {
            std::cerr << "[Error] " << lphq_last_error() << "\n";
            return 4;
        }
        total_kmers = st.num_kmers;
        // ptbb_build.cpp:100-101 (num_bits() of the reference counts the in-memory object; here the serialized size)
        pt_fields = "," + std::to_string(st.file_bytes * 8) + "," + std::to_string(static_cast<double>(st.file_bytes * 8) / st.num_kmers);
        if (a.has("--check")) {
            lphq_index* idx = nullptr;
            if (lphq_open_pthash(out.c_str(), k, kmer_bits, &idx) != LPHQ_OK) idx = nullptr;
            rc = check_bijection(idx, st.num_kmers, "PTHash");
            if (idx) lphq_close(idx);
        }
    }
    if (a.get("-b") && rc == 0) {
        const std::string out = *a.get("-b");
        lphq_build_stats st;
        if (lphq_build_bbhash(ptrs.data(), lens.data(), n_recs, k, kmer_bits, gamma, threads, out.c_str(), &st) != LPHQ_OK) {
            std::cerr << "[Error] " << lphq_last_error() << "\n";
            return 4;
        }
        total_kmers = st.num_kmers;
        // ptbb_build.cpp:161-162 (totalBitSize() of the reference adds 42 bytes per exact-map entry; here the serialized size)
        bb_fields = "," + std::to_string(st.file_bytes * 8) + "," + std::to_string(static_cast<double>(st.file_bytes * 8) / st.num_kmers);
        if (a.has("--check")) {
            lphq_index* idx = nullptr;
            if (lphq_open_bbhash(out.c_str(), k, kmer_bits, &idx) != LPHQ_OK) idx = nullptr;
            rc = check_bijection(idx, st.num_kmers, "BBHash");
            if (idx) lphq_close(idx);
        }
    }
    std::cout << input << "," << k << "," << total_kmers << pt_fields << bb_fields << std::endl;
    return rc;
}

int ptbb_query_main(int argc, char** argv) {
    Args a = parse(argc, argv, {"-q", "-k", "-p", "-b", "--kmer-bits", "--device", "--gpus", "--devices", "--batch-mb", "--threads"}, {"--pageable"});
    if (a.help || a.bad || !a.get("-q") || !a.get("-k")) {
        std::cerr << "Usage: " << argv[0] << " [-h,--help] -q input_filename -k k [-p pthash_filename] [--kmer-bits 64|128] [--device N | --gpus N | --devices a,b,..] [--batch-mb N] "
                     "[--threads N] [--pageable]\n";
        return 1;
    }
    const std::string input = *a.get("-q");
    const int k = std::atoi(a.get("-k")->c_str());
    const int kmer_bits = a.get("--kmer-bits") ? std::atoi(a.get("--kmer-bits")->c_str()) : 128;
    std::cout << input << "," << k;  // ptbb_query.cpp:23
    const int device = a.get("--device") ? std::atoi(a.get("--device")->c_str()) : 0;
    std::vector<int> devices;
    if (!device_list(a, device, devices)) return 1;
    const size_t batch_bases = (size_t)(a.get("--batch-mb") ? std::max(1l, std::atol(a.get("--batch-mb")->c_str())) : 32) << 20;
    const int threads = a.get("--threads") ? std::max(1, std::atoi(a.get("--threads")->c_str())) : default_parse_threads();
    const int gz_threads = a.get("--threads") ? threads : lphq::usable_cpus();  // (the .gz decoder scales further than the parser: fastx_batch.hpp)
    // one timed pass per function given (ptbb_query.cpp:40-72), each ",<file>,<ns per k-mer>", else ",,"
    for (const char* flag : {"-p", "-b"}) {
        if (!a.get(flag)) {
            std::cout << ",,";
            continue;
        }
        const std::string file = *a.get(flag);
        lphq_index* idx = nullptr;
        const int orc = flag[1] == 'p' ? lphq_open_pthash(file.c_str(), k, kmer_bits, &idx) : lphq_open_bbhash(file.c_str(), k, kmer_bits, &idx);
        if (orc != LPHQ_OK || lphq_attach_devices(idx, devices.data(), (int)devices.size()) != LPHQ_OK) {
            std::cerr << "\n[Error] " << lphq_last_error() << "\n";
            if (idx) lphq_close(idx);
            return 2;
        }
        lphq_info info;
        lphq_get_info(idx, &info);
        std::vector<GpuWorker> gpus(devices.size());
        for (size_t d = 0; d < devices.size(); ++d)
            if (!gpus[d].init(devices[d])) {
                std::cerr << "\n[Error] cannot create a stream on device " << devices[d] << "\n";
                lphq_close(idx);
                return 2;
            }
        {  // set-up outside the timer: first launch of the kernel (code object load)
            const std::string dummy(info.k, 'A');
            std::vector<uint64_t> one(1);
            size_t n1 = 0;
            for (size_t d = 0; d < devices.size(); ++d) {
                (void)hipSetDevice(devices[d]);
                (void)lphq_query_contig(idx, dummy.data(), dummy.size(), 1, one.data(), 1, &n1);
            }
        }
        uint64_t total = 0, us = 0;
        std::string err;
        const bool opened = query_pass(idx, info, input, 1, batch_bases, threads, gz_threads, !a.has("--pageable"), gpus, total, us, err, nullptr);
        for (GpuWorker& g : gpus) g.destroy();
        lphq_close(idx);
        if (!opened) {
            std::cerr << "\n[ptbb_file_itr] Unable to open file " << input << "\n";  // ptbb.hpp:84-85 throws here
            return 2;
        }
        if (!err.empty()) {
            std::cerr << "\n[Error] " << err << "\n";
            return 3;
        }
        std::cout << "," << file << "," << static_cast<double>(us * 1000) / total;
    }
    std::cout << "\n";
    return 0;
}

}  // namespace

int main(int argc, char* argv[]) {
    {  // the reference ships ptbb_build / ptbb_query as executables of their own (CMakeLists.txt:59-63): a link to this binary under those names behaves so
        const char* slash = std::strrchr(argv[0], '/');
        const std::string self = slash ? slash + 1 : argv[0];
        if (self == "ptbb_build") return ptbb_build_main(argc, argv);
        if (self == "ptbb_query") return ptbb_query_main(argc, argv);
    }
    if (argc < 2) return help(argv[0]);
    const std::string tool = argv[1];
    if (tool == "build-p") return build_main(LPHQ_PARTITIONED, argc - 1, argv + 1);
    if (tool == "build-u") return build_main(LPHQ_UNPARTITIONED, argc - 1, argv + 1);
    if (tool == "query-p") return query_main(LPHQ_PARTITIONED, argc - 1, argv + 1);
    if (tool == "query-u") return query_main(LPHQ_UNPARTITIONED, argc - 1, argv + 1);
    if (tool == "records") return records_main(argc - 1, argv + 1);
    if (tool == "inflate") return inflate_main(argc - 1, argv + 1);
    if (tool == "ptbb-build" || tool == "ptbb_build") return ptbb_build_main(argc - 1, argv + 1);
    if (tool == "ptbb-query" || tool == "ptbb_query") return ptbb_query_main(argc - 1, argv + 1);
    std::cerr << "Unsupported tool '" << tool << "'." << std::endl;
    return help(argv[0]);
}
