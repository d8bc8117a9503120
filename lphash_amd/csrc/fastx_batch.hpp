// fastx_batch.hpp -- batches of FASTA/FASTQ records for the CLI, parsed by several threads when that is safe.
//
// The record semantics are kseq's (external/kseq.h:192-240, restated in fastx.hpp).  For a plain (not gzip) file whose
// first byte is '>', a header is exactly one line, so every later line that does not start with '>' or '@' is a
// sequence line -- as long as no sequence line starts with '+' (which would open a FASTQ quality string, where '>' and
// '@' are ordinary characters).  Such a file is mmap'ed; a batch is a run of whole records, cut at ANY line start into
// pieces that are parsed concurrently with the same per-record state machine (a piece that begins inside a record
// contributes its leading bases to the record begun in an earlier piece, so one 250-Mbase chromosome is still parsed by
// all threads).  A piece that meets a line starting with '+' -- or the one case where kseq's '\r' rule would need the
// length of a record begun elsewhere -- makes the reader fall back, from the start of the current batch, to the serial
// FastxReader for the rest of the file, so the records are the serial reader's in every case.  gzip input and files
// that do not start with '>' or '@' take the serial reader from the beginning -- except BGZF (bgzip / htslib: gzip members of
// <= 64 KiB whose compressed size is in the header's 'BC' extra field): its blocks are independent deflate streams, so they are inflated
// concurrently (raw inflate + CRC-32 and length check per block) into a WINDOW of text that stands in for the mapped file: the window
// always covers the batch being cut plus what it takes to find the next record start, slides forward with the batches and is as large
// as a few batches (one record larger than that grows it, as it grows the arenas) -- memory stays proportional to batch_bases however
// large the file is, like the plain path.  Any other regular .gz -- one member or several, what `gzip` and the reference's own inputs are --
// is decoded by all threads as well (parallel_gz.hpp: spans of the compressed stream decoded from guessed block starts with placeholders for
// the unknown history, accepted only where the chain of block borders confirms them) into the same sliding window; only its length is not
// known before the end.  Non-regular files, one thread, tiny files: the serial reader (gz_source.hpp ahead of FastxReader).
//
// A batch is a list of segments (bases in file order; a record may run across segments) plus the record lengths; segments
// point into one of two arenas owned by the reader, used alternately, so a batch stays valid while the NEXT one is
// being parsed (the CLI parses batch i+1 on a helper thread while batch i is on its way to the GPU) and dies with the
// call after that.
//
// Plain FASTQ (first byte '@') is parsed by several threads too when it is the usual four-line kind: a record starts at a line that begins
// with '@' and whose next-but-one line begins with '+' (a quality line may begin with '@', but then the line two further down is a sequence
// line, which never begins with '+'); batches and pieces are cut at such lines.  A piece is accepted only if every record in it is exactly
// header / one sequence line / '+' line / one quality line of the same length, without '\r' and with the next record (or the end of the
// file) right behind -- the one shape in which kseq_read (external/kseq.h:192-240) has no choices to make; anything else sends the rest of
// the file, from the start of the current batch, to the serial FastxReader.
//
// Arenas are allocated once and reused: first-touch page faults of fresh memory cost more than the
// parsing itself (measured: 40 ms per 100 MB in the build container, against 12 ms to parse them with 8 threads).
// They come from a caller-supplied allocator so that the CLI can use pinned host memory (hipHostMalloc).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "fastx.hpp"
#include "parallel_gz.hpp"
#include "work_pool.hpp"

namespace lphq {

struct FastxBatch {
    struct Segment {
        char* p;
        size_t n;
    };
    std::vector<Segment> segs;    // bases of the batch, in order (ascending addresses within one arena)
    std::vector<uint64_t> lens;   // length of every record, in order (sum = total bases of segs)
    size_t bases = 0;
    void clear() {
        segs.clear();
        lens.clear();
        bases = 0;
    }
    // one contiguous segment (for consumers that want a pointer per record): slides the segments together in place
    void compact() {
        if (segs.size() <= 1) return;
        char* dst = segs[0].p + segs[0].n;
        for (size_t i = 1; i < segs.size(); ++i) {
            memmove(dst, segs[i].p, segs[i].n);
            dst += segs[i].n;
        }
        segs.assign(1, Segment{segs[0].p, (size_t)(dst - segs[0].p)});
    }
};

class FastxBatchReader {
public:
    using AllocFn = void* (*)(size_t);
    using FreeFn = void (*)(void*);

    // min_piece: smallest file span worth a thread of its own (tests pass 1 to force many pieces)
    // gz_threads: threads of the span-parallel .gz decoder (0: as many as `threads`).  Decoding is compute-bound and scales to twice the CPU-time quota
    // of a container (14.2 GB/s with 32 threads against 9.9 with 16 on the GPU box), parsing is memory-bound and does not (plain FASTA: 0.044 ns per k-mer
    // with 16 threads, 0.12 with 32): a caller may give the two stages different numbers.
    FastxBatchReader(const char* path, int threads, AllocFn alloc = nullptr, FreeFn dealloc = nullptr, size_t min_piece = 1 << 16, int gz_threads = 0)
        : path_(path), threads_(std::max(1, threads)), gz_threads_(gz_threads > 0 ? gz_threads : std::max(1, threads)), min_piece_(std::max<size_t>(1, min_piece)),
          alloc_(alloc ? alloc : &::malloc), free_(dealloc ? dealloc : &::free) {
        fd_ = ::open(path, O_RDONLY);
        if (fd_ < 0) return;
        struct stat st;
        unsigned char head = 0;
        if (fstat(fd_, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && ::pread(fd_, &head, 1, 0) == 1 && (head == '>' || head == '@')) {  // (gzip starts with 0x1f)
            fastq_ = head == '@';
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (m != MAP_FAILED) {
                map_ = (const char*)m;
                size_ = lim_ = (size_t)st.st_size;
                (void)madvise(m, size_, MADV_SEQUENTIAL);
            }
        }
        if (!map_ && fd_ >= 0 && fstat(fd_, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 28) bgzf_detect((size_t)st.st_size);
        if (!map_ && !bgzf_ && fd_ >= 0 && !getenv("LPHQ_GZ_ZLIB") && !getenv("LPHQ_GZ_SERIAL")) {
            const char* sp = getenv("LPHQ_GZ_SPAN");  // (tests) compressed bytes per span
            pgz_ = ParallelGzReader::open(path, gz_threads_, sp ? (size_t)atol(sp) : 0);
        }
        if (!map_ && !bgzf_ && !pgz_) serial_ = new FastxReader(path);
        ok_ = map_ || bgzf_ || pgz_ || serial_->ok();
    }
    ~FastxBatchReader() {
        delete serial_;
        delete pgz_;
        delete pool_;
        delete inflate_pool_;
        if (bgzf_) munmap((void*)bgzf_, bgzf_size_);
        if (wbuf_) {
            free(wbuf_);
            map_ = nullptr;
        }
        if (map_) munmap((void*)map_, size_);
        if (fd_ >= 0) ::close(fd_);
        for (Arena& a : arenas_)
            if (a.p) free_(a.p);
    }
    FastxBatchReader(const FastxBatchReader&) = delete;
    FastxBatchReader& operator=(const FastxBatchReader&) = delete;

    // How many batches the caller holds at once (default 2: the one being consumed and the one being filled; a driver that keeps depth x GPUs batches in
    // flight asks for that many + 1).  Before the first next() / prealloc().
    void set_live_batches(int n) {
        if (n >= 2 && arenas_[0].p == nullptr && arenas_[1].p == nullptr) arenas_.resize((size_t)n);
    }
    // Allocate both arenas for batches of about batch_bases bases now (e.g. before a caller starts its timer: pinned host memory
    // is expensive to allocate); next() grows them if a batch needs more.
    void prealloc(size_t batch_bases) {
        const size_t want = map_ ? std::min(size_, batch_bases + (batch_bases >> 3) + (1 << 16)) : batch_bases + (batch_bases >> 3) + (1 << 16);
        for (size_t i = 0; i < arenas_.size() && ok_; ++i) {
            which_ = (which_ + 1) % (int)arenas_.size();
            (void)reserve(want);
        }
        if (!pool_ && threads_ > 1 && !serial_) pool_ = new WorkPool(threads_);  // the parse threads, parked
        if (bgzf_ && !serial_) (void)inflate_pool();
        if (pgz_ && !wbuf_) {  // the decoder's buffers and threads, and a window of one batch and a few pieces: allocated and touched here, nothing decoded
            pgz_->prealloc();
            const size_t cap = want + 16 * pgz_->span() + kWindowSlack;
            wbuf_ = (char*)malloc(cap);
            if (wbuf_) {
                wcap_ = cap;
                memset(wbuf_, 0, cap);
            }
        }
    }
    bool ok() const { return ok_; }
    bool parallel() const { return map_ != nullptr && serial_ == nullptr; }  // (tests) which path produced the last batch

    // Next batch of roughly batch_bases bases (whole records; at least one record unless the input is exhausted).
    // Returns false when there is nothing left.
    bool next(FastxBatch& b, size_t batch_bases) {
        b.clear();
        which_ = (which_ + 1) % (int)arenas_.size();
        if (bgzf_ && !windowed_ && !serial_ && !bgzf_index()) {  // first call on a BGZF file; on any irregularity the serial reader takes the whole file
            serial_ = new FastxReader(path_.c_str());
            if (!serial_->ok()) ok_ = false;
        }
        if (pgz_ && !windowed_ && !serial_ && !gz_begin()) {  // first call on any other .gz; text that does not begin like FASTA / FASTQ, or damage: the serial reader decides
            delete pgz_;  // (its threads would go on decoding ahead)
            pgz_ = nullptr;
            serial_ = new FastxReader(path_.c_str());
            if (!serial_->ok()) ok_ = false;
        }
        while (ok_ && parallel()) {
            if (off_ >= size_) return false;
            if (!next_parallel(b, batch_bases)) {
                // a '+' line (or a damaged BGZF block): the rest of the file goes through the serial reader, starting at this batch's first record.
                // Plain text: by offset.  BGZF: a fresh reader skips the records already handed out (by the equivalence of the two readers they are
                // the first `emitted_` records of the serial one; seeking zlib to an uncompressed offset would drop what precedes a damaged block)
                delete pgz_;
                pgz_ = nullptr;
                serial_ = new FastxReader(path_.c_str());
                if (!serial_->ok()) ok_ = false;
                else if (!windowed_) ok_ = serial_->seek(off_);
                else {
                    std::string skip;
                    for (uint64_t i = 0; i < emitted_ && serial_->next(skip); ++i) {}
                }
                b.clear();
                break;
            }
            emitted_ += b.lens.size();
            if (!b.lens.empty()) return true;  // (an empty window: a last header without a line end)
        }
        return ok_ && next_serial(b, batch_bases);
    }

    bool bgzf() const { return windowed_ && bgzf_ != nullptr; }  // (tests) the input is inflated block-parallel
    bool parallel_gz() const { return windowed_ && pgz_ != nullptr; }  // (tests) ... or span-parallel (parallel_gz.hpp)
    size_t window_capacity() const { return wcap_; }  // (tests) bytes of inflated text held at once

private:
    // ---- BGZF (SAM/BAM specification, section 4.1): gzip member = 10-byte header with FLG.FEXTRA, XLEN, extra subfields (one of them
    // SI1 'B', SI2 'C', SLEN 2, BSIZE = member size - 1), raw deflate data, CRC32, ISIZE ----
    struct BgzfBlock {
        size_t in, in_len;    // deflate payload within the file
        size_t out, out_len;  // where it inflates to
        uint32_t crc;
    };
    static uint32_t le32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
    // total size of the member at `at` and where its deflate payload starts; 0 if it is not a BGZF member
    static size_t bgzf_member(const unsigned char* f, size_t size, size_t at, size_t& payload) {
        if (size - at < 26 || f[at] != 0x1f || f[at + 1] != 0x8b || f[at + 2] != 8 || !(f[at + 3] & 4)) return 0;
        const size_t xlen = (size_t)f[at + 10] | ((size_t)f[at + 11] << 8);
        if (size - at < 12 + xlen + 8) return 0;
        size_t bsize = 0;
        for (size_t x = at + 12; x + 4 <= at + 12 + xlen;) {
            const size_t slen = (size_t)f[x + 2] | ((size_t)f[x + 3] << 8);
            if (f[x] == 'B' && f[x + 1] == 'C' && slen == 2 && x + 6 <= at + 12 + xlen) bsize = ((size_t)f[x + 4] | ((size_t)f[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (f[at + 3] & ~4u) return 0;  // FNAME / FCOMMENT / FHCRC: legal gzip, not what bgzip writes; leave it to zlib
        if (bsize < 12 + xlen + 8 || bsize > size - at) return 0;
        payload = at + 12 + xlen;
        return bsize;
    }
    void bgzf_detect(size_t size) {
        unsigned char head[18];
        size_t payload;
        if (::pread(fd_, head, 18, 0) != 18 || head[0] != 0x1f || head[1] != 0x8b || !(head[3] & 4)) return;
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) return;
        if (!bgzf_member((const unsigned char*)m, size, 0, payload)) {
            munmap(m, size);
            return;
        }
        bgzf_ = (const unsigned char*)m;
        bgzf_size_ = size;
    }
    // Index every member of the file (no inflation yet); true = it is BGZF from the first byte to the last, size_ is the length of its text and
    // the first window is in place and begins like a FASTA / FASTQ file.
    bool bgzf_index() {
        const unsigned char* f = bgzf_;
        const size_t size = bgzf_size_;
        size_t at = 0, total = 0;
        while (at < size) {
            size_t payload = 0;
            const size_t bs = bgzf_member(f, size, at, payload);
            if (!bs) return false;
            BgzfBlock bl;
            bl.in = payload;
            bl.in_len = at + bs - 8 - payload;
            bl.crc = le32(f + at + bs - 8);
            bl.out = total;
            bl.out_len = le32(f + at + bs - 4);
            if (bl.out_len > 65536) return false;
            total += bl.out_len;
            blocks_.push_back(bl);
            at += bs;
        }
        if (total == 0) return false;  // empty text: the serial reader decides
        size_ = total;
        windowed_ = true;
        if (!bgzf_window(0, std::min<size_t>(total, 1 << 16)) || (map_[0] != '>' && map_[0] != '@')) {  // damaged or headerless text: the serial reader decides
            windowed_ = false;
            map_ = nullptr;
            size_ = lim_ = 0;
            return false;
        }
        fastq_ = map_[0] == '@';
        return true;
    }
    // Make the window hold the text [from, upto') with upto' >= min(upto, size_) (whole blocks).  Text before `from` is given up; what is
    // already inflated beyond it is kept (moved to the front of the buffer only when the buffer would otherwise overflow, so that the copying
    // is of the look-ahead tail, not of the batch).  The new blocks are inflated by threads_ threads.  false = a block is damaged.
    bool bgzf_window(size_t from, size_t upto) {
        upto = std::min(upto, size_);
        size_t b_end = blk_next_;
        size_t new_bytes = 0;
        while (b_end < blocks_.size() && blocks_[b_end].out < upto) new_bytes += blocks_[b_end++].out_len;
        if (b_end == blk_next_) return true;
        if (from > wend_) from = wend_;
        if (!window_room(from, new_bytes)) return false;
        const unsigned char* f = bgzf_;
        char* const base = wbuf_;
        const size_t wbase = wbase_, b0 = blk_next_, nb = b_end - blk_next_;
        const int T = (int)std::min<size_t>((size_t)gz_threads_, nb);  // (inflating is compute-bound like the span decoder: the larger of the two thread counts)
        std::vector<int> bad((size_t)T, 0);
        auto work = [&](int t) {
            // every block through the own decoder (fast_inflate.hpp) into a private buffer -- the decoder writes a few bytes past what it has
            // decoded, which must not land in a neighbour's part of the text -- then into place; CRC-32 and length as the member's trailer says
            DeflateDecoder* dec = new DeflateDecoder();
            std::vector<uint8_t> tmp((size_t)65536 + DeflateDecoder::kSlack + 64);
            for (size_t i = b0 + (size_t)t; i < b0 + nb; i += (size_t)T) {
                const BgzfBlock& bl = blocks_[i];
                dec->reset(f + bl.in, f + bl.in + bl.in_len);
                size_t pos = 0;
                const bool okb = dec->decode(tmp.data(), pos, 65536) == 1 && pos == bl.out_len && dec->stream_end() <= f + bl.in + bl.in_len &&
                                 crc32_slice16(0, tmp.data(), pos) == bl.crc;
                if (!okb) {
                    bad[(size_t)t] = 1;
                    break;
                }
                memcpy(base + (bl.out - wbase), tmp.data(), pos);
            }
            delete dec;
        };
        inflate_pool().run((size_t)T, [&](size_t t) { work((int)t); });
        for (int x : bad)
            if (x) return false;
        blk_next_ = b_end;
        wend_ = blocks_[b_end - 1].out + blocks_[b_end - 1].out_len;
        lim_ = wend_;
        map_ = (const char*)((uintptr_t)wbuf_ - (uintptr_t)wbase_);  // map_[p] is the text at absolute position p for p in [wbase_, wend_)
        return true;
    }
    static constexpr size_t kWindowSlack = 1 << 12;
    // Room for new_bytes more bytes of text behind wend_, keeping [from, wend_): the retained tail slides to the front of the buffer only when
    // the buffer would otherwise overflow (so that the copying is of the look-ahead tail, not of the batch), and the buffer grows if that is not enough.
    bool window_room(size_t from, size_t new_bytes) {
        const size_t keep = wend_ - from;  // already inflated text that is still needed
        if ((wend_ - wbase_) + new_bytes + kWindowSlack > wcap_) {
            if (keep + new_bytes + kWindowSlack > wcap_) {
                const size_t cap = std::max<size_t>(2 * wcap_, keep + new_bytes + new_bytes / 2 + kWindowSlack);
                char* nb = (char*)malloc(cap);
                if (!nb) return false;
                if (keep) memcpy(nb, wbuf_ + (from - wbase_), keep);
                free(wbuf_);
                wbuf_ = nb;
                wcap_ = cap;
            } else if (keep && from != wbase_) {
                memmove(wbuf_, wbuf_ + (from - wbase_), keep);
            }
            wbase_ = from;
        }
        return true;
    }
    bool window_fill(size_t from, size_t upto) { return bgzf_ ? bgzf_window(from, upto) : (pgz_ ? gz_window(from, upto) : false); }

    // ---- any other .gz: pieces of text from the span-parallel decoder (parallel_gz.hpp), appended to the window in order ----
    static constexpr size_t kUnknownSize = ~(size_t)0 >> 2;
    bool gz_begin() {
        size_ = kUnknownSize;
        windowed_ = true;
        if (!gz_window(0, 1) || wend_ == 0 || (map_[0] != '>' && map_[0] != '@')) {
            windowed_ = false;
            map_ = nullptr;
            size_ = lim_ = 0;
            return false;
        }
        fastq_ = map_[0] == '@';
        return true;
    }
    // Make the window hold the text [from, upto') with upto' >= upto, or up to the end of the data (size_ is known from then on).  The pieces are
    // written into the window by the decoder's threads; false = the stream is damaged (or a CRC-32 / length is wrong).
    bool gz_window(size_t from, size_t upto) {
        if (from > wend_) from = wend_;
        while (wend_ < upto && !gz_done_) {
            const size_t n = pgz_->next_len();
            if (n == 0) {
                gz_done_ = true;
                break;
            }
            if ((wend_ - wbase_) + n + kWindowSlack > wcap_) {  // the buffer is about to move: the pieces on their way into it first
                if (!pgz_->flush()) return false;
                if (!window_room(from, n)) return false;
            }
            pgz_->emit((uint8_t*)wbuf_ + (wend_ - wbase_));
            wend_ += n;
        }
        if (!pgz_->flush() || pgz_->error()) return false;
        if (gz_done_) size_ = wend_;
        lim_ = wend_;
        map_ = (const char*)((uintptr_t)wbuf_ - (uintptr_t)wbase_);
        return true;
    }

    struct Piece {
        size_t begin, end;   // file span; begin is a line start
        char* out;           // output position in the arena (capacity end - begin)
        size_t n = 0;        // bases written
        size_t head = 0;     // of which: bases before the first header of the piece (they belong to the previous piece's last record)
        std::vector<uint64_t> lens;  // records that BEGIN in this piece
        bool unsafe = false;
    };

    struct Arena {
        char* p = nullptr;
        size_t cap = 0;
    };
    // current arena with room for n bytes, keeping its first `keep` bytes when it has to grow
    bool reserve(size_t n, size_t keep = 0) {
        Arena& a = arenas_[which_];
        if (n <= a.cap) return true;
        const size_t cap = n + n / 4 + 4096;
        char* np = (char*)alloc_(cap);
        if (!np) {
            ok_ = false;
            return false;
        }
        if (keep) memcpy(np, a.p, keep);
        if (a.p) free_(a.p);
        a.p = np;
        a.cap = cap;
        return true;
    }

    // first position p in [from, size_) with map_[p] == '>' and map_[p - 1] == '\n'; size_ if there is none
    // FASTQ: first line start p >= from with map_[p] == '@' whose next-but-one line begins with '+'; size_ if there is none
    // (searches end at lim_: the end of the file for mapped text, the end of the inflated window for BGZF -- the caller extends the window and asks again)
    size_t next_fastq_record(size_t from) const {
        size_t p = from;
        if (p > 0 && map_[p - 1] != '\n') p = next_line_start(p, lim_);
        while (p < lim_) {
            if (map_[p] == '@') {
                const size_t l1 = next_line_start(p, lim_), l2 = l1 < lim_ ? next_line_start(l1, lim_) : lim_;
                if (l2 < lim_ && map_[l2] == '+') return p;
            }
            p = next_line_start(p, lim_);
        }
        return lim_;
    }

    size_t next_record_start(size_t from) const {
        if (fastq_) return next_fastq_record(from);
        size_t p = std::max<size_t>(from, 1);
        while (p < lim_) {
            const char* q = (const char*)memchr(map_ + p, '>', lim_ - p);
            if (!q) return lim_;
            p = (size_t)(q - map_);
            if (map_[p - 1] == '\n') return p;
            ++p;
        }
        return lim_;
    }

    // first line start in (from, end]; `end` if the rest holds no line end
    size_t next_line_start(size_t from, size_t end) const {
        if (from >= end) return end;
        const char* q = (const char*)memchr(map_ + from, '\n', end - from);
        return q ? (size_t)(q - map_) + 1 : end;
    }

    // Sequence lines from p on, appended at out, until a line that starts a record ('>' / '@') or the end of the piece.
    // `rec` = where the record's bases begin if that is inside this piece, else nullptr (record begun in an earlier piece).
    // Returns false if the piece cannot be parsed on its own.
    bool sequence_lines(size_t& p, size_t e, char*& out, char* rec) const {
        const char* s = map_;
        char* const local = rec ? rec : out;
        while (p < e) {
            const char c = s[p];
            if (c == '>' || c == '@') break;  // next record
            if (c == '+') return false;       // FASTQ quality string ahead
            if (c == '\n') {
                ++p;
                continue;
            }
            const char* ln = (const char*)memchr(s + p, '\n', e - p);
            const size_t n = ln ? (size_t)(ln - (s + p)) : e - p;
            memcpy(out, s + p, n);
            out += n;
            // kseq: `\r` before the line end is dropped once the sequence has > 1 characters -- but not when the line's first
            // character is the last byte of the input (ks_getuntil2 then returns before that rule)
            if (out[-1] == '\r' && (ln || n > 1)) {
                if (out - local > 1) --out;
                else if (!rec) return false;  // ... and how long the record is so far is only known to an earlier piece
            }
            p += n + (ln ? 1 : 0);
        }
        return true;
    }

    // four-line FASTQ records only (see the file header); anything else marks the piece unsafe
    void parse_piece_fastq(Piece& pc) const {
        const char* s = map_;
        size_t p = pc.begin;
        const size_t e = pc.end;
        char* out = pc.out;
        while (p < e) {
            if (s[p] != '@') {
                pc.unsafe = true;
                return;
            }
            const char* h = (const char*)memchr(s + p, '\n', e - p);                              // header line
            const char* q = h ? (const char*)memchr(h + 1, '\n', (size_t)(s + e - (h + 1))) : nullptr;  // sequence line
            if (!h || !q || q + 1 >= s + e || q[1] != '+') {
                pc.unsafe = true;
                return;
            }
            const char* pl = (const char*)memchr(q + 1, '\n', (size_t)(s + e - (q + 1)));  // '+' line
            if (!pl) {
                pc.unsafe = true;
                return;
            }
            const size_t n = (size_t)(q - (h + 1));
            const char* ql = pl + 1;  // quality line: exactly n characters, then a line end (or the end of the file)
            const size_t left = (size_t)(s + e - ql);
            const bool at_eof = pc.end == size_ && left == n;
            if (n == 0 || left < n || (!at_eof && (left == n || ql[n] != '\n')) || memchr(h + 1, '\r', n) || memchr(ql, '\r', n) ||
                memchr(ql, '\n', n) != nullptr) {
                pc.unsafe = true;
                return;
            }
            memcpy(out, h + 1, n);
            out += n;
            pc.lens.push_back((uint64_t)n);
            p = (size_t)(ql - s) + n + (at_eof ? 0 : 1);
        }
        pc.n = (size_t)(out - pc.out);
    }

    // kseq's record loop over one piece.
    void parse_piece(Piece& pc) const {
        if (fastq_) {
            parse_piece_fastq(pc);
            return;
        }
        const char* s = map_;
        size_t p = pc.begin;
        const size_t e = pc.end;
        char* out = pc.out;
        if (p < e && s[p] != '>' && s[p] != '@') {  // the piece begins inside a record
            if (!sequence_lines(p, e, out, nullptr)) {
                pc.unsafe = true;
                return;
            }
            pc.head = (size_t)(out - pc.out);
        }
        while (p < e) {
            // header: s[p] is '>' or '@'
            const char* nl = (const char*)memchr(s + p, '\n', e - p);
            if (!nl) {
                // last header of the file without a line end: kseq still yields an (empty) record, unless the header
                // character is the very last byte of the input (fastx.hpp: next())
                if (pc.end != size_) pc.unsafe = true;  // cannot happen: pieces end at line starts (defensive)
                else if (e - p > 1) pc.lens.push_back(0);
                break;
            }
            p = (size_t)(nl - s) + 1;
            char* rec = out;
            if (!sequence_lines(p, e, out, rec)) {
                pc.unsafe = true;
                return;
            }
            pc.lens.push_back((uint64_t)(out - rec));
        }
        pc.n = (size_t)(out - pc.out);
    }

    // (LPHQ_READER_TRACE: where a batch's time goes, on stderr)
    static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    bool next_parallel(FastxBatch& b, size_t batch_bases) {
        static const bool trace = getenv("LPHQ_READER_TRACE") != nullptr;
        const double t_in = trace ? now_ms() : 0;
        const size_t want = std::max<size_t>(batch_bases, 1);
        size_t end;
        if (!windowed_) {
            end = want >= size_ - off_ ? size_ : next_record_start(off_ + want);
        } else {
            // BGZF: the window must hold the batch and the start of the record after it; a search that runs into the end of the window while the
            // file goes on asks for more
            // (a quarter more each time, not twice as much: the window is filled synchronously for BGZF, and for a plain .gz waiting for text nobody needs
            // yet delays the batch that could already be on its way to the GPU)
            for (size_t ahead = want + (1 << 16);; ahead += std::max<size_t>(ahead / 4, (size_t)1 << 16)) {
                if (!window_fill(off_, ahead >= size_ - off_ ? size_ : off_ + ahead)) return false;  // damaged block: the serial reader takes over at off_
                if (want >= size_ - off_) {
                    end = size_;
                    if (lim_ == size_) break;
                    continue;
                }
                end = off_ + want < lim_ ? next_record_start(off_ + want) : lim_;
                if (end < lim_ || lim_ == size_) break;
            }
        }
        const double t_text = trace ? now_ms() : 0;
        // sized for a full batch from the start, so that a short first batch does not cause a second cold allocation
        if (!reserve(std::max(end - off_, std::min(want, size_ - off_)))) return true;
        char* const arena = arenas_[which_].p;
        // cut [off_, end) into pieces at line starts
        // (text that is still being decoded by a pool of its own: pieces of >= 1 MB -- the decoder's threads hold the cores, and waking sixteen parse
        // threads for a 4-MB batch cost more than the parsing; the tests' tiny min_piece_ still cuts finely)
        const size_t piece = (pgz_ && windowed_ && !gz_done_) ? std::max<size_t>(min_piece_ < (1 << 16) ? min_piece_ : (size_t)1 << 20, 1) : min_piece_;
        const int T = (int)std::min<size_t>((size_t)threads_, std::max<size_t>(1, (end - off_) / piece));
        std::vector<Piece> pieces;
        size_t from = off_;
        for (int t = 1; t <= T && from < end; ++t) {
            size_t to = t == T ? end : (fastq_ ? std::min(end, next_fastq_record(off_ + (end - off_) / T * t)) : next_line_start(off_ + (end - off_) / T * t, end));
            if (to <= from) continue;
            Piece pc;
            pc.begin = from;
            pc.end = to;
            pc.out = arena + (from - off_);
            pieces.push_back(std::move(pc));
            from = to;
        }
        const double t_cut = trace ? now_ms() : 0;
        pool().run(pieces.size(), [&](size_t i) { parse_piece(pieces[i]); });
        if (trace)
            fprintf(stderr, "[reader] batch of %zu bytes of text: text ready after %.2f ms, cut into %zu pieces after %.2f, parsed after %.2f\n", end - off_, t_text - t_in,
                    pieces.size(), t_cut - t_in, now_ms() - t_in);
        for (const Piece& pc : pieces)
            if (pc.unsafe) return false;
        for (const Piece& pc : pieces) {
            if (pc.n) b.segs.push_back({pc.out, pc.n});
            if (pc.head) {
                if (b.lens.empty()) return false;  // cannot happen: the batch begins with a header (defensive)
                b.lens.back() += pc.head;
            }
            b.lens.insert(b.lens.end(), pc.lens.begin(), pc.lens.end());
            b.bases += pc.n;
        }
        off_ = end;
        return true;
    }

    bool next_serial(FastxBatch& b, size_t batch_bases) {
        if (serial_done_) return false;
        size_t used = 0;
        std::string seq;
        while (true) {
            if (!serial_->next(seq)) {
                serial_done_ = true;
                break;
            }
            if (!reserve(std::max<size_t>(used + seq.size(), std::min<size_t>(batch_bases, (size_t)1 << 28)), used)) return false;
            memcpy(arenas_[which_].p + used, seq.data(), seq.size());
            used += seq.size();
            b.lens.push_back(seq.size());
            if (used >= batch_bases) break;
        }
        if (used) b.segs.push_back({arenas_[which_].p, used});
        b.bases = used;
        return !b.lens.empty();
    }

    std::string path_;
    int threads_, gz_threads_;
    size_t min_piece_;
    AllocFn alloc_;
    FreeFn free_;
    int fd_ = -1;
    const char* map_ = nullptr;
    size_t size_ = 0, off_ = 0;
    bool fastq_ = false;                   // the mapped text begins with '@': four-line FASTQ pieces
    size_t lim_ = 0;                       // end of the text map_ can be read at: size_ for a mapped file, the end of the window for BGZF
    const unsigned char* bgzf_ = nullptr;  // mapped BGZF file
    size_t bgzf_size_ = 0;
    std::vector<BgzfBlock> blocks_;        // its members (filled by the first next())
    size_t blk_next_ = 0;                  // first member not inflated yet
    char* wbuf_ = nullptr;                 // window of inflated text [wbase_, wend_); map_ = wbuf_ - wbase_
    size_t wcap_ = 0, wbase_ = 0, wend_ = 0;
    bool windowed_ = false;
    uint64_t emitted_ = 0;                 // records handed out by the parallel path so far
    WorkPool* pool_ = nullptr;             // the parse (and BGZF inflate) threads: created by prealloc() or at the first batch that needs them
    WorkPool& pool() {
        if (!pool_) pool_ = new WorkPool(threads_);
        return *pool_;
    }
    WorkPool* inflate_pool_ = nullptr;     // BGZF blocks: gz_threads_ threads when that is more than the parser's
    WorkPool& inflate_pool() {
        if (gz_threads_ <= threads_) return pool();
        if (!inflate_pool_) inflate_pool_ = new WorkPool(gz_threads_);
        return *inflate_pool_;
    }
    ParallelGzReader* pgz_ = nullptr;      // a .gz that is not BGZF, decoded by all threads into the same window
    bool gz_done_ = false;
    FastxReader* serial_ = nullptr;
    bool serial_done_ = false, ok_ = false;
    std::vector<Arena> arenas_ = std::vector<Arena>(2);  // the batches a caller may hold at once (set_live_batches)
    int which_ = 0;
};

}  // namespace lphq
