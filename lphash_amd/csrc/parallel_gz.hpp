// parallel_gz.hpp -- one gzip member decoded by many threads.
//
// A single-member .gz (what `gzip` writes and what the reference's inputs are: data/queries/*.fasta.gz, read through gzread by
// src/query.cpp:43-58 / external/kseq.h:192-240 inside the timer) has no index and every match may reach 32 KiB back, but it is not serial:
//
//   1. SPANS.  The compressed file is cut into spans of equal size.  The thread that gets span j looks for the first bit at or after the
//      span's first byte that can begin a dynamic-Huffman block -- not final, type 2, HLIT / HDIST in range, a COMPLETE code-length code, all
//      lengths decodable, complete literal and distance codes, an end-of-block code (find_block_start) -- and starts decoding there.  A start
//      whose first block fails, or whose second block header is invalid, was not one: the search goes on behind it.
//   2. PLACEHOLDERS.  Such a thread does not know the 32 KiB of text before its start.  It decodes into 16-bit symbols (fast_inflate.hpp,
//      decode_as<uint16_t>) behind a prefix of 32 Ki distinct placeholder values 256 + i: a match that reaches into the unknown text copies
//      placeholders, matches of matches copy them again.  Once the last 32 Ki symbols hold no placeholder any more the thread goes on in plain
//      bytes, at the speed of the serial decoder.  It stops at the first block border at or after the end of its span.
//   3. THE CHAIN.  The caller's thread walks the pieces in order.  It knows the bit where the text so far ends; the piece of the span that bit
//      lies in is accepted iff it started at exactly that bit.  Anything else -- no start found, a false one, an error, a stored or fixed block
//      where the span begins, a new gzip member, a piece that grew beyond its cap -- is decoded by the caller's thread itself from the known bit
//      with the known window, up to the end of that span: never wrong, only serial for that span.  What is accepted is therefore always what the
//      serial decoder would have produced up to placeholders, by construction and not by plausibility.
//   4. RESOLUTION.  With the window before a piece known (the last 32 KiB of what precedes it -- for which only the LAST 32 Ki symbols of each
//      piece have to be resolved in order, a few microseconds), its placeholders are replaced through a 33-KiB table, its bytes copied behind
//      them and its CRC-32 taken, all by the pool and straight into the caller's buffer; the members' CRC-32 and ISIZE are checked from the
//      pieces' CRCs (crc32_concat) when the caller collects them.
//
// Memory: at most `depth` pieces in flight, each capped at max(64 x its span, 32 MiB) of text.  Nothing is decoded before the first next_len() (a caller's
// timer includes all of the decoding, as the reference's does); prealloc() only allocates, touches the buffers and parks the threads.
// Checked against zlib in tests/test_host_cpu.py (byte equality on every corpus file with several thread counts and span sizes, members,
// garbage, damage) and under ASan / UBSan / TSan-free logic in tests/san.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "fast_inflate.hpp"
#include "gz_source.hpp"

namespace lphq {

// CRC-32 of A || B from crc(A), crc(B) and the length of B: crc(A) times x^(8 len) modulo the CRC polynomial (reflected bit order: the most
// significant bit of the word is x^0), plus crc(B).  (The pre- and post-inversions of the two CRCs cancel.)
inline uint32_t crc32_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
inline uint32_t crc32_concat(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    uint32_t pw = 0x00800000u, r = 0x80000000u;  // x^8, x^0
    for (uint64_t n = len_b; n; n >>= 1) {
        if (n & 1) r = crc32_mulmod(r, pw);
        pw = crc32_mulmod(pw, pw);
    }
    return crc32_mulmod(r, crc_a) ^ crc_b;
}

class ParallelGzReader {
public:
    static constexpr size_t W = DeflateDecoder::kWindow;
    static constexpr size_t kSlack = DeflateDecoder::kSlack + 64;
    static constexpr uint64_t kNone = ~(uint64_t)0;

    // nullptr: not a regular file that begins with a gzip member, or not worth the threads (one thread, a file of less than two spans).
    // span = compressed bytes per piece (0: chosen from the file size and the thread count).
    static ParallelGzReader* open(const char* path, int threads, size_t span = 0) {
        if (threads < 2) return nullptr;
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return nullptr;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) {
            ::close(fd);
            return nullptr;
        }
        const size_t size = (size_t)st.st_size;
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return nullptr;
        const size_t hdr = gzip_header_bytes((const uint8_t*)m, size);
        if (!span) span = std::min<size_t>(kMaxSpan, std::max<size_t>(kMinSpan, size / ((size_t)threads * 2)));
        if (!hdr || size < 2 * span) {
            munmap(m, size);
            return nullptr;
        }
        return new ParallelGzReader((const uint8_t*)m, size, hdr, threads, span);
    }
    ~ParallelGzReader() {
        {
            std::lock_guard<std::mutex> g(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto& t : pool_) t.join();
        for (size_t i = 0; i < depth_; ++i) {
            free(slots_[i].sym);
            free(slots_[i].text);
        }
        for (FbBuf* f : fbs_) {
            free(f->p);
            delete f;
        }
        delete dec_;
        munmap((void*)file_, size_);
    }
    ParallelGzReader(const ParallelGzReader&) = delete;
    ParallelGzReader& operator=(const ParallelGzReader&) = delete;

    // Set-up that belongs before a caller's timer (like gzopen's and kseq_init's allocations before t.start(), src/query.cpp:43-48): buffers
    // allocated and touched, threads created and parked.  No byte of the file is decoded here.
    void prealloc() { start_pool(false); }

    // Length of the next piece of text; 0 = end of the data (error() says whether it ended well).  Blocks until the piece is decoded.
    size_t next_len() {
        if (have_piece_) return piece_.nsym + piece_.ntext;
        start_pool(true);
        while (!ended_) {
            if (advance()) return piece_.nsym + piece_.ntext;
        }
        return 0;
    }
    // Write that piece to dst[0, next_len()): done by the pool, complete after flush().  dst must stay where it is until then.
    void emit(uint8_t* dst) {
        if (!have_piece_) return;
        have_piece_ = false;
        records_.push_back(Record{0, piece_.nsym + piece_.ntext, piece_.ends_member, piece_.want_crc, piece_.want_isize});
        EmitTask t;
        t.p = piece_;
        t.dst = dst;
        t.rec = &records_.back();
        {
            std::lock_guard<std::mutex> g(mu_);
            emit_q_.push_back(t);
            ++emits_pending_;
        }
        cv_.notify_all();
    }
    // Wait for the emits so far and check the members they completed; false = a CRC-32 or length does not match its trailer (error() stays true).
    bool flush() {
        {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&] { return emits_pending_ == 0; });
        }
        for (const Record& r : records_) {
            crc_ = crc32_concat(crc_, r.crc, r.len);
            isize_ += r.len;
            if (r.ends_member) {
                if (crc_ != r.want_crc || (uint32_t)isize_ != r.want_isize) error_ = true;
                crc_ = 0;
                isize_ = 0;
            }
        }
        records_.clear();
        return !error_;
    }
    bool error() const { return error_; }
    // (tests, diagnostics) pieces taken from the pool / decoded by the caller's thread
    uint64_t pieces_parallel() const { return n_accepted_; }
    uint64_t pieces_serial() const { return n_fallback_; }
    size_t span() const { return span_; }

private:
    static constexpr size_t kMinSpan = 128u << 10, kMaxSpan = 2u << 20;
    static constexpr size_t kCheck = 32768;  // symbols between two looks at whether the placeholders have died out

    enum SlotState { kFree, kBusy, kDone, kFailed, kEmitting };
    struct Slot {
        int state = kFree;
        size_t index = 0;
        bool started = false;
        std::atomic<bool> abandon{false};  // set by the caller's thread, polled by the span's worker
        uint64_t start_bit = 0, end_bit = 0;
        size_t end_byte = 0;  // first byte behind the deflate stream, when the final block ended in this piece
        bool final = false;
        uint16_t* sym = nullptr;  // W placeholders, then the piece's first nsym symbols
        size_t sym_cap = 0, nsym = 0;
        uint8_t* text = nullptr;  // W bytes of history (the last W symbols, narrowed), then the rest of the piece
        size_t text_cap = 0, ntext = 0;
        double t_search = 0, t_16 = 0, t_8 = 0;  // (LPHQ_PGZ_TRACE) seconds spent looking for the start, in 16-bit symbols, in bytes
        std::vector<uint8_t> table;  // 256 + W: what the placeholders of this piece stand for (filled when the piece is accepted)
    };
    struct FbBuf {
        uint8_t* p = nullptr;
        size_t cap = 0;
        bool busy = false;
    };
    struct Piece {
        const uint16_t* sym = nullptr;
        size_t nsym = 0;
        const uint8_t* text = nullptr;
        size_t ntext = 0;
        const uint8_t* table = nullptr;
        Slot* slot = nullptr;
        FbBuf* fb = nullptr;
        bool ends_member = false;
        uint32_t want_crc = 0, want_isize = 0;
    };
    struct Record {
        uint32_t crc;
        size_t len;
        bool ends_member;
        uint32_t want_crc, want_isize;
    };
    struct EmitTask {
        Piece p;
        uint8_t* dst;
        Record* rec;
    };

    ParallelGzReader(const uint8_t* file, size_t size, size_t hdr, int threads, size_t span)
        : file_(file), size_(size), threads_(threads), span_(span), nspans_((size + span - 1) / span), depth_((size_t)threads * 2 + 2) {
        slots_.reset(new Slot[depth_]);
        cur_bit_ = first_bit_ = (uint64_t)hdr * 8;
        member_start_ = true;
        max_piece_ = std::max<size_t>(64 * span_, (size_t)32 << 20);
        dec_ = new DeflateDecoder();
        memset(win_, 0, W);
    }
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
    static uint64_t load64(const uint8_t* p) {
        uint64_t v;
        memcpy(&v, p, 8);
        return v;
    }
    size_t est_text() const { return span_ * 5 + (64u << 10); }

    // ---------------------------------------------------------------- pool ----------------------------------------------------------------
    void start_pool(bool activate) {
        if (pool_.empty()) {
            touched_ = 0;
            for (int t = 0; t < threads_; ++t) pool_.emplace_back([this, t] { worker(t); });
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&] { return touched_ == threads_; });
        }
        if (activate && !active_) {
            t_start_ = now();
            {
                std::lock_guard<std::mutex> g(mu_);
                active_ = true;
            }
            cv_.notify_all();
        }
    }
    bool can_claim() const {
        return active_ && !stop_claims_ && next_claim_ < nspans_ && next_claim_ < consumer_span_ + depth_ && slots_[next_claim_ % depth_].state == kFree;
    }
    static bool grow16(Slot& s, size_t need) {
        if (need <= s.sym_cap) return true;
        const size_t cap = need + need / 2;
        uint16_t* p = (uint16_t*)realloc(s.sym, cap * 2);
        if (!p) return false;
        s.sym = p;
        s.sym_cap = cap;
        return true;
    }
    static bool grow8(uint8_t*& buf, size_t& cap_io, size_t need) {
        if (need <= cap_io) return true;
        const size_t cap = need + need / 2;
        uint8_t* p = (uint8_t*)realloc(buf, cap);
        if (!p) return false;
        buf = p;
        cap_io = cap;
        return true;
    }
    void worker(int t) {
        // this thread's share of the slots: allocated and touched once (first-touch page faults cost more than decoding into warm memory)
        for (size_t i = (size_t)t; i < depth_; i += (size_t)threads_) {
            Slot& s = slots_[i];
            // (placeholders die out after ~0.5 M symbols on DNA text: the 16-bit buffer starts at 1.5 M symbols and grows if a span needs more)
            const size_t n = W + est_text() + kSlack, n16 = W + std::min<size_t>(est_text(), (size_t)3 << 19) + kSlack;
            s.sym = (uint16_t*)malloc(n16 * 2);
            s.text = (uint8_t*)malloc(n);
            if (s.sym && s.text) {
                s.sym_cap = n16;
                s.text_cap = n;
                memset(s.sym, 0, n16 * 2);
                memset(s.text, 0, n);
                for (size_t k = 0; k < W; ++k) s.sym[k] = (uint16_t)(256 + k);
            } else {
                free(s.sym);
                free(s.text);
                s.sym = nullptr;
                s.text = nullptr;
            }
            s.table.assign(256 + W, 0);
        }
        DeflateDecoder* dec = new DeflateDecoder();
        {
            std::lock_guard<std::mutex> g(mu_);
            ++touched_;
        }
        cv_.notify_all();
        for (;;) {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&] { return quit_ || !emit_q_.empty() || can_claim(); });
            if (quit_) break;
            if (!emit_q_.empty()) {
                EmitTask task = emit_q_.front();
                emit_q_.pop_front();
                g.unlock();
                run_emit(task);
                g.lock();
                if (task.p.slot) task.p.slot->state = kFree;
                if (task.p.fb) task.p.fb->busy = false;
                --emits_pending_;
                g.unlock();
                cv_.notify_all();
                continue;
            }
            const size_t j = next_claim_++;
            Slot& s = slots_[j % depth_];
            s.state = kBusy;
            s.index = j;
            s.started = s.final = false;
            s.abandon = false;
            s.nsym = s.ntext = 0;
            g.unlock();
            const bool ok = s.sym && decode_span(s, *dec);
            g.lock();
            s.state = s.abandon ? kFree : (ok ? kDone : kFailed);
            g.unlock();
            cv_.notify_all();
        }
        delete dec;
    }
    static void run_emit(const EmitTask& t) {
        const Piece& p = t.p;
        uint8_t* d = t.dst;
        const uint8_t* tab = p.table;
        for (size_t i = 0; i < p.nsym; ++i) d[i] = tab[p.sym[i]];
        if (p.ntext) memcpy(d + p.nsym, p.text, p.ntext);
        t.rec->crc = crc32_slice16(0, d, p.nsym + p.ntext);
    }

    // ---------------------------------------------------- step 1: where a block may begin ----------------------------------------------------
    // Bits of the file from `bit` on (>= 57 of them valid); the caller keeps 8 readable bytes behind bit / 8.
    uint64_t peek(uint64_t bit) const { return load64(file_ + (bit >> 3)) >> (bit & 7); }

    // Everything RFC 1951, 3.2.7 lets one check about a dynamic block header without the data behind it.
    bool plausible_dynamic_header(uint64_t bit) const {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        if ((bit >> 3) + 8 + 8 > size_) return false;
        uint64_t v = peek(bit);
        if ((v & 7) != 4) return false;  // BFINAL = 0, BTYPE = 2 (bits 0, 0, 1 in stream order)
        const uint32_t hlit = (uint32_t)((v >> 3) & 31) + 257, hdist = (uint32_t)((v >> 8) & 31) + 1, hclen = (uint32_t)((v >> 13) & 15) + 4;
        if (hlit > 286 || hdist > 30) return false;
        uint64_t p = bit + 17;
        v = peek(p);  // 19 x 3 = 57 bits at most
        uint8_t cl[19] = {0};
        uint32_t kraft = 0;
        for (uint32_t i = 0; i < hclen; ++i) {
            const uint32_t l = (uint32_t)(v >> (3 * i)) & 7u;
            cl[order[i]] = (uint8_t)l;
            if (l) kraft += 128u >> l;
        }
        if (kraft != 128) return false;  // zlib refuses an incomplete code-length code as well as an over-subscribed one
        p += 3 * hclen;
        uint16_t tab[128];
        {
            int count[8] = {0};
            for (int i = 0; i < 19; ++i) count[cl[i]]++;
            count[0] = 0;
            uint32_t next_code[8], code = 0;
            for (int l = 1; l <= 7; ++l) {
                code = (code + (uint32_t)count[l - 1]) << 1;
                next_code[l] = code;
            }
            for (int s = 0; s < 19; ++s) {
                const int l = cl[s];
                if (!l) continue;
                const uint32_t c = next_code[l]++;
                uint32_t r = 0;
                for (int i = 0; i < l; ++i) r |= ((c >> i) & 1u) << (l - 1 - i);
                for (uint32_t i = r; i < 128; i += 1u << l) tab[i] = (uint16_t)((s << 4) | l);  // (complete: every entry is written)
            }
        }
        uint8_t lens[286 + 30 + 138];
        uint32_t n = 0;
        const uint32_t total = hlit + hdist;
        while (n < total) {
            if ((p >> 3) + 8 > size_) return false;
            v = peek(p);
            const uint16_t e = tab[v & 127];
            p += e & 15;
            v >>= e & 15;
            const uint32_t sym = e >> 4;
            if (sym < 16) {
                lens[n++] = (uint8_t)sym;
                continue;
            }
            uint32_t rep, val = 0;
            if (sym == 16) {
                if (n == 0) return false;
                val = lens[n - 1];
                rep = 3 + (uint32_t)(v & 3);
                p += 2;
            } else if (sym == 17) {
                rep = 3 + (uint32_t)(v & 7);
                p += 3;
            } else {
                rep = 11 + (uint32_t)(v & 127);
                p += 7;
            }
            if (n + rep > total) return false;
            while (rep--) lens[n++] = (uint8_t)val;
        }
        if (lens[256] == 0) return false;
        // complete codes, or the one exception zlib makes: a single codeword of length 1 (and, for distances, none at all)
        auto complete = [](const uint8_t* l, uint32_t cnt, bool may_be_empty) {
            uint32_t sum = 0, used = 0, maxl = 0;
            for (uint32_t i = 0; i < cnt; ++i)
                if (l[i]) {
                    sum += 32768u >> l[i];
                    ++used;
                    maxl = std::max<uint32_t>(maxl, l[i]);
                }
            if (sum == 32768u) return true;
            if (used == 0) return may_be_empty;
            return sum < 32768u && maxl == 1;
        };
        return complete(lens, hlit, false) && complete(lens + hlit, hdist, true);
    }
    // first bit in [from, to) that passes; kNone if there is none
    uint64_t find_block_start(uint64_t from, uint64_t to, const Slot& s) const {
        uint64_t bit = from;
        while (bit < to) {
            const size_t byte = (size_t)(bit >> 3);
            if (byte + 16 > size_) return kNone;
            if ((byte & 0xFFFF) == 0 && s.abandon) return kNone;
            const uint64_t v = load64(file_ + byte);
            for (uint32_t sft = (uint32_t)(bit & 7); sft < 8; ++sft) {
                const uint64_t x = v >> sft;
                if ((x & 7) == 4 && ((x >> 3) & 31) <= 29 && ((x >> 8) & 31) <= 29) {
                    const uint64_t b = (uint64_t)byte * 8 + sft;
                    if (b >= to) return kNone;
                    if (plausible_dynamic_header(b)) return b;
                }
            }
            bit = ((uint64_t)byte + 1) * 8;
        }
        return kNone;
    }

    // --------------------------------------------- steps 1 + 2: one span, by a thread of the pool ---------------------------------------------
    void publish_start(Slot& s, uint64_t bit) {
        {
            std::lock_guard<std::mutex> g(mu_);
            s.start_bit = bit;
            s.started = true;
        }
        cv_.notify_all();
    }
    bool decode_span(Slot& s, DeflateDecoder& dec) {
        const uint64_t from = (uint64_t)s.index * span_ * 8;
        const bool last = s.index + 1 >= nspans_;
        const uint64_t boundary = last ? (uint64_t)size_ * 8 : (uint64_t)(s.index + 1) * span_ * 8;
        uint64_t search = from;
        s.t_search = s.t_16 = s.t_8 = 0;
        if (s.index == 0) {
            // span 0 begins at the first block of the file's first member: a known start with nothing before it -- plain bytes from the first symbol
            // (round 4, second step: the caller's thread used to decode this span itself, 1.2-1.7 ms during which nothing was handed on)
            const double t1 = now();
            dec.reset_at_bit(file_, first_bit_, file_ + size_);
            if (!last) dec.set_stop_bit(boundary);
            publish_start(s, first_bit_);
            if (!grow8(s.text, s.text_cap, W + est_text() + kSlack)) return false;
            s.nsym = 0;
            size_t p8 = 0;  // (relative to text + W: a match that reaches before the data is an error, as in the serial decoder)
            int rc;
            for (;;) {
                const size_t step_end = std::min(s.text_cap - W - kSlack, p8 + (4u << 20));
                rc = dec.decode(s.text + W, p8, step_end - 1);
                if (rc != 0) break;
                if (s.abandon || p8 > max_piece_) return false;
                if (p8 + W + kSlack >= s.text_cap && !grow8(s.text, s.text_cap, s.text_cap + s.text_cap / 2)) return false;
            }
            if (rc < 0) return false;
            s.ntext = p8;
            s.final = rc == 1;
            s.end_bit = dec.bit_position();
            s.end_byte = (size_t)(dec.stream_end() - file_);
            s.t_8 = now() - t1;
            if (trace_) fprintf(stderr, "[pgz] span 0: known start, bytes %.2f ms (%zu), done at %.2f ms\n", s.t_8 * 1e3, s.ntext, (now() - t_start_) * 1e3);
            return true;
        }
        for (;;) {
            const double t0 = now();
            const uint64_t cand = find_block_start(search, boundary, s);
            s.t_search += now() - t0;
            if (cand == kNone) return false;
            dec.reset_at_bit(file_, cand, file_ + size_);
            if (!last) dec.set_stop_bit(boundary);
            size_t pos = W, scan_from = W, last_marker_end = W;
            bool published = false, retry = false;
            int rc;
            // ---- 16-bit symbols, until no placeholder is left among the last W of them
            for (;;) {
                if (!grow16(s, pos + kCheck + kSlack)) return false;
                rc = dec.decode_as<uint16_t>(s.sym, pos, pos + kCheck - 1);
                if (!published) {
                    if (rc < 0 && dec.blocks_started() <= 2) {  // not a block start after all
                        retry = true;
                        break;
                    }
                    if (rc != 0 || dec.blocks_started() >= 2) {
                        publish_start(s, cand);
                        published = true;
                    }
                }
                if (rc != 0) break;
                if (s.abandon) return false;
                for (size_t b = scan_from; b < pos;) {  // placeholders among the new symbols, to a granularity of 1 Ki
                    const size_t e = std::min(pos, b + 1024);
                    uint32_t acc = 0;
                    for (size_t i = b; i < e; ++i) acc |= s.sym[i];
                    if (acc >> 8) last_marker_end = e;
                    b = e;
                }
                scan_from = pos;
                if (published && pos - last_marker_end >= W) break;  // clean: bytes from here on
                if (pos - W > max_piece_) return false;
            }
            s.t_16 += now() - t0;
            if (retry) {
                search = cand + 1;
                continue;
            }
            if (rc < 0) return false;
            const double t1 = now();
            s.nsym = pos - W;
            s.ntext = 0;
            if (rc == 0) {
                // ---- plain bytes: the last W symbols are the history
                const size_t want = W + est_text() + kSlack;
                if (!grow8(s.text, s.text_cap, want)) return false;
                for (size_t i = 0; i < W; ++i) s.text[i] = (uint8_t)s.sym[pos - W + i];
                size_t p8 = W;
                for (;;) {
                    const size_t step_end = std::min(s.text_cap - kSlack, p8 + (4u << 20));  // (a look at `abandon` every 4 MiB)
                    rc = dec.decode(s.text, p8, step_end - 1);
                    if (rc != 0) break;
                    if (s.abandon || p8 - W + s.nsym > max_piece_) return false;
                    if (p8 + kSlack >= s.text_cap && !grow8(s.text, s.text_cap, s.text_cap + s.text_cap / 2)) return false;
                }
                if (rc < 0) return false;
                s.ntext = p8 - W;
            }
            s.final = rc == 1;
            s.end_bit = dec.bit_position();
            s.end_byte = (size_t)(dec.stream_end() - file_);
            s.t_8 = now() - t1;
            s.t_16 -= s.t_search;
            if (trace_) fprintf(stderr, "[pgz] span %zu: search %.2f ms, 16-bit %.2f ms (%zu symbols), bytes %.2f ms (%zu), done at %.2f ms\n", s.index, s.t_search * 1e3, s.t_16 * 1e3, s.nsym, s.t_8 * 1e3, s.ntext, (now() - t_start_) * 1e3);
            return true;
        }
    }

    // ------------------------------------------------ step 3: the chain, on the caller's thread ------------------------------------------------
    size_t span_of(uint64_t bit) const { return std::min<size_t>(nspans_ - 1, (size_t)((bit >> 3) / span_)); }

    // One more piece into piece_ (true), or nothing yet / the end (false; ended_ tells which).
    bool advance() {
        const uint64_t at = fb_cont_ ? dec_->bit_position() : cur_bit_;
        const size_t j = span_of(at);
        Slot* s = nullptr;
        {
            std::unique_lock<std::mutex> g(mu_);
            // spans the text has passed: their pieces are of no use any more
            while (consumer_span_ < j) {
                Slot& o = slots_[consumer_span_ % depth_];
                if (o.index == consumer_span_ && next_claim_ > consumer_span_) {
                    if (o.state == kBusy) o.abandon = true;
                    else if (o.state == kDone || o.state == kFailed) o.state = kFree;
                }
                ++consumer_span_;
            }
            if (next_claim_ < consumer_span_) next_claim_ = consumer_span_;
            cv_.notify_all();
            // (consumer_span_ == j: span j's piece has not been looked at yet -- a second visit to the same span, after a member ended in it or a
            // serial piece was cut, must not wait for a slot that may already belong to a later span)
            if (!fb_cont_ && consumer_span_ == j && !stop_claims_) {
                Slot& c = slots_[j % depth_];
                cv_.wait(g, [&] { return next_claim_ > j && c.index == j && (c.started || c.state == kDone || c.state == kFailed); });
                if (c.started && c.start_bit == cur_bit_) {
                    cv_.wait(g, [&] { return c.state != kBusy; });
                    if (c.state == kDone) s = &c;
                    else c.state = kFree;
                } else {
                    if (c.state == kBusy) c.abandon = true;
                    else c.state = kFree;
                }
                consumer_span_ = j + 1;  // whatever happens now, span j's slot is spoken for
                if (next_claim_ < consumer_span_) next_claim_ = consumer_span_;
                cv_.notify_all();
            }
        }
        piece_ = Piece();
        if (trace_) fprintf(stderr, "[pgz] caller at span %zu: %s at %.2f ms\n", j, s ? "accepted" : "serial", (now() - t_start_) * 1e3);
        if (s) {
            ++n_accepted_;
            memcpy(s->table.data() + 256, win_, W);
            for (int i = 0; i < 256; ++i) s->table[(size_t)i] = (uint8_t)i;
            piece_.sym = s->sym + W;
            piece_.nsym = s->nsym;
            piece_.text = s->text + W;
            piece_.ntext = s->ntext;
            piece_.table = s->table.data();
            piece_.slot = s;
            advance_window(piece_);
            cur_bit_ = s->end_bit;
            member_start_ = false;
            if (s->final) end_of_member(s->end_byte);
            if (piece_.nsym + piece_.ntext == 0) {  // nothing to hand out; an empty end of a member still has its trailer checked
                if (piece_.ends_member) records_.push_back(Record{0, 0, true, piece_.want_crc, piece_.want_isize});
                std::lock_guard<std::mutex> g(mu_);
                s->state = kFree;
                cv_.notify_all();
                return false;
            }
            {
                std::lock_guard<std::mutex> g(mu_);
                s->state = kEmitting;
            }
            have_piece_ = true;
            return true;
        }
        return serial_piece(j);
    }

    // The caller's thread decodes from the known bit with the known window to the end of span j (or to the end of the member, or to the cap).
    bool serial_piece(size_t j) {
        ++n_fallback_;
        if (n_fallback_ >= 64 && n_accepted_ < n_fallback_ / 4 && !stop_claims_) {
            // a stream the pool cannot help with (thousands of small members, stored blocks throughout): its threads would go on decoding spans that
            // are thrown away -- from here on the caller's thread alone
            std::lock_guard<std::mutex> g(mu_);
            stop_claims_ = true;
            for (size_t i = 0; i < depth_; ++i)
                if (slots_[i].state == kBusy) slots_[i].abandon = true;
        }
        FbBuf* fb = nullptr;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (FbBuf* f : fbs_)
                if (!f->busy) fb = f;
            if (!fb) {
                fb = new FbBuf();
                fbs_.push_back(fb);
            }
            fb->busy = true;
        }
        auto fail = [&] {
            error_ = true;
            ended_ = true;
        };
        if (!grow8(fb->p, fb->cap, W + est_text() + kSlack)) {
            fail();
            return false;
        }
        memcpy(fb->p, win_, W);
        if (!fb_cont_) {
            dec_->reset_at_bit(file_, cur_bit_, file_ + size_);
            if (j + 1 < nspans_) dec_->set_stop_bit((uint64_t)(j + 1) * span_ * 8);
        } else if (j + 1 < nspans_) {
            dec_->set_stop_bit((uint64_t)(j + 1) * span_ * 8);
        } else {
            dec_->set_stop_bit(kNone);
        }
        // at the start of a member nothing lies before the data: a match that reaches there is an error, as in the serial decoder
        const size_t lead = (member_start_ && !fb_cont_) ? W : 0;
        size_t pos = W - lead;
        int rc;
        for (;;) {
            rc = dec_->decode(fb->p + lead, pos, fb->cap - lead - kSlack - 1);
            if (rc != 0) break;
            if (pos + lead - W >= max_piece_) break;  // cut here (in the middle of a block): the next piece goes on with this decoder
            if (!grow8(fb->p, fb->cap, fb->cap + fb->cap / 2)) {
                fail();
                return false;
            }
        }
        piece_.text = fb->p + W;
        piece_.ntext = pos + lead - W;
        piece_.fb = fb;
        if (trace_) fprintf(stderr, "[pgz] caller decoded %zu bytes itself, done at %.2f ms\n", piece_.ntext, (now() - t_start_) * 1e3);
        advance_window(piece_);
        member_start_ = false;
        fb_cont_ = rc == 0;
        if (rc < 0) {  // what was decoded stands (a prefix of the text), the data ends here
            fail();
        } else if (rc == 1) {
            end_of_member((size_t)(dec_->stream_end() - file_));
        } else if (rc == 3) {
            cur_bit_ = dec_->bit_position();
        }
        if (piece_.ntext == 0) {
            if (piece_.ends_member) records_.push_back(Record{0, 0, true, piece_.want_crc, piece_.want_isize});
            std::lock_guard<std::mutex> g(mu_);
            fb->busy = false;
            return false;
        }
        have_piece_ = true;
        return true;
    }

    // The final block of a member ended at byte e: its trailer, and the member that may follow (anything else there is ignored, as gzread does).
    void end_of_member(size_t e) {
        if (size_ - e < 8) {
            error_ = true;
            ended_ = true;
            return;
        }
        piece_.ends_member = true;
        piece_.want_crc = le32(file_ + e);
        piece_.want_isize = le32(file_ + e + 4);
        const size_t at = e + 8;
        const size_t hdr = at < size_ ? gzip_header_bytes(file_ + at, size_ - at) : 0;
        if (!hdr) {
            ended_ = true;
            std::lock_guard<std::mutex> g(mu_);
            stop_claims_ = true;
            for (size_t i = 0; i < depth_; ++i)
                if (slots_[i].state == kBusy) slots_[i].abandon = true;
            return;
        }
        cur_bit_ = (uint64_t)(at + hdr) * 8;
        member_start_ = true;
        memset(win_, 0, W);
    }

    // win_ := the last W bytes of the text up to and including piece p (its placeholders stand for the bytes of the old win_)
    void advance_window(const Piece& p) {
        if (p.ntext >= W) {
            memcpy(win_, p.text + p.ntext - W, W);
            return;
        }
        uint8_t nw[W];
        const size_t need = W - p.ntext, from_sym = std::min(need, p.nsym), from_old = need - from_sym;
        if (from_old) memcpy(nw, win_ + W - from_old, from_old);
        for (size_t i = 0; i < from_sym; ++i) {
            const uint16_t v = p.sym[p.nsym - from_sym + i];
            nw[from_old + i] = v < 256 ? (uint8_t)v : win_[v - 256];
        }
        if (p.ntext) memcpy(nw + from_old + from_sym, p.text, p.ntext);
        memcpy(win_, nw, W);
    }

    const uint8_t* file_;
    size_t size_;
    int threads_;
    size_t span_, nspans_, depth_, max_piece_ = 0;
    std::unique_ptr<Slot[]> slots_;
    std::vector<FbBuf*> fbs_;
    std::vector<std::thread> pool_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<EmitTask> emit_q_;
    size_t emits_pending_ = 0;
    size_t next_claim_ = 0, consumer_span_ = 0;
    int touched_ = 0;
    bool quit_ = false, active_ = false, stop_claims_ = false;
    // caller's side
    DeflateDecoder* dec_ = nullptr;
    uint8_t win_[W];
    uint64_t cur_bit_ = 0, first_bit_ = 0;
    bool member_start_ = true, fb_cont_ = false, ended_ = false, have_piece_ = false;
    std::atomic<bool> error_{false};
    Piece piece_;
    std::deque<Record> records_;
    uint32_t crc_ = 0;
    uint64_t isize_ = 0;
    uint64_t n_accepted_ = 0, n_fallback_ = 0;
    const bool trace_ = getenv("LPHQ_PGZ_TRACE") != nullptr;
    double t_start_ = 0;
};

}  // namespace lphq
