// gz_source.hpp -- a .gz file as a byte stream, decoded ahead of its reader: the compressed file is mapped, one thread inflates it
// (fast_inflate.hpp) into a small ring of 4-MiB chunks, a second thread checks every member's CRC-32 and length (the
// inflating thread is the bottleneck and does nothing else), and read() hands the bytes out in order.  What gzread() would deliver:
// concatenated members are concatenated text, bytes after the last member that do not begin another one are ignored, a damaged or
// truncated stream ends the data where the damage is (error() then says so; the callers treat it like gzread's -1: end of input).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "fast_inflate.hpp"

namespace lphq {

// CRC-32 (IEEE 802.3, the one of RFC 1952) sixteen bytes per step ("slicing by 16": sixteen 256-entry tables, generated from the reflected polynomial
// at first use).  zlib 1.2.11's crc32 (four bytes per step) could not keep up with the decoder on one thread.  Checked against zlib's in the tests.
inline uint32_t crc32_slice16(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t T[16][256];
    static bool ready = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 16; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFFu];
        return true;
    }();
    (void)ready;
    crc = ~crc;
    while (n >= 16) {
        uint32_t a, b, c, d;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        memcpy(&c, p + 8, 4);
        memcpy(&d, p + 12, 4);
        a ^= crc;
        crc = T[15][a & 0xFFu] ^ T[14][(a >> 8) & 0xFFu] ^ T[13][(a >> 16) & 0xFFu] ^ T[12][a >> 24] ^ T[11][b & 0xFFu] ^ T[10][(b >> 8) & 0xFFu] ^
              T[9][(b >> 16) & 0xFFu] ^ T[8][b >> 24] ^ T[7][c & 0xFFu] ^ T[6][(c >> 8) & 0xFFu] ^ T[5][(c >> 16) & 0xFFu] ^ T[4][c >> 24] ^ T[3][d & 0xFFu] ^
              T[2][(d >> 8) & 0xFFu] ^ T[1][(d >> 16) & 0xFFu] ^ T[0][d >> 24];
        p += 16;
        n -= 16;
    }
    while (n--) crc = (crc >> 8) ^ T[0][(crc ^ *p++) & 0xFFu];
    return ~crc;
}

class GzFastSource {
public:
    // nullptr when the path is not a regular file that begins with a gzip member (the caller then lets zlib have it)
    static GzFastSource* open(const char* path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return nullptr;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) {
            ::close(fd);
            return nullptr;
        }
        void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return nullptr;
        if (!gzip_header_bytes((const uint8_t*)m, (size_t)st.st_size)) {
            munmap(m, (size_t)st.st_size);
            return nullptr;
        }
        (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        GzFastSource* s = new GzFastSource((const uint8_t*)m, (size_t)st.st_size);
        if (!s->allocate()) {
            delete s;
            return nullptr;
        }
        return s;  // (the threads start with the first read: nothing is decoded ahead of a caller's timer)
    }
    ~GzFastSource() {
        {
            std::lock_guard<std::mutex> g(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        if (producer_.joinable()) producer_.join();
        if (checker_.joinable()) checker_.join();
        for (Chunk& c : ring_) free(c.buf);
        munmap((void*)file_, size_);
    }
    GzFastSource(const GzFastSource&) = delete;
    GzFastSource& operator=(const GzFastSource&) = delete;

    // up to n bytes into dst; 0 = end of the data
    size_t read(char* dst, size_t n) {
        start();
        size_t got = 0;
        while (got < n && !done_) {
            if (!cur_) {
                std::unique_lock<std::mutex> g(mu_);
                Chunk& c = ring_[read_seq_ % kRing];
                cv_.wait(g, [&] { return (c.state == kFilled && c.seq == read_seq_) || quit_; });
                if (quit_) break;
                cur_ = &c;
                cur_off_ = 0;
            }
            const size_t take = std::min(n - got, cur_->len - cur_off_);
            memcpy(dst + got, cur_->buf + DeflateDecoder::kWindow + cur_off_, take);
            got += take;
            cur_off_ += take;
            if (cur_off_ == cur_->len && !retire_current()) break;
        }
        return got;
    }
    // The same bytes without the copy: the next stretch of text (at most `most` bytes) inside the source's own memory, valid until the following
    // call of next_span() or read(); n = 0 (and nullptr) at the end of the data.
    const char* next_span(size_t& n, size_t most) {
        start();
        n = 0;
        while (!done_) {
            if (!cur_) {
                std::unique_lock<std::mutex> g(mu_);
                Chunk& c = ring_[read_seq_ % kRing];
                cv_.wait(g, [&] { return (c.state == kFilled && c.seq == read_seq_) || quit_; });
                if (quit_) return nullptr;
                cur_ = &c;
                cur_off_ = 0;
            }
            if (cur_off_ < cur_->len) {
                n = std::min(most, cur_->len - cur_off_);
                const char* p = (const char*)cur_->buf + DeflateDecoder::kWindow + cur_off_;
                cur_off_ += n;
                return p;
            }
            if (!retire_current()) return nullptr;
        }
        return nullptr;
    }
    bool error() const { return error_; }

private:
    // the current chunk has been handed out completely: it goes back once the checker is through with it as well (error() is final after the last
    // chunk).  false: that was the last chunk, or the source is shutting down.
    bool retire_current() {
        {
            std::unique_lock<std::mutex> g(mu_);
            Chunk* c = cur_;
            cv_.wait(g, [&] { return c->checked || quit_; });
            if (quit_) return false;
            if (c->last) {
                done_ = true;
                return false;
            }
            c->state = kFree;
        }
        cv_.notify_all();
        cur_ = nullptr;
        ++read_seq_;
        return true;
    }
    static constexpr size_t kChunk = 4u << 20;
    static constexpr int kRing = 4;
    enum ChunkState { kFree, kFilled };
    struct Segment {  // a stretch of one member inside a chunk
        size_t off, len;
        bool ends_member;
        uint32_t crc, isize;  // the member's trailer, when it ends here
    };
    struct Chunk {
        uint8_t* buf = nullptr;  // kWindow bytes of history, kChunk of payload, slack
        size_t len = 0;          // payload bytes
        uint64_t seq = 0;
        ChunkState state = kFree;
        bool checked = false, last = false;
        std::vector<Segment> segs;
    };

    GzFastSource(const uint8_t* file, size_t size) : file_(file), size_(size) {}
    bool allocate() {
        for (Chunk& c : ring_) {
            c.buf = (uint8_t*)malloc(DeflateDecoder::kWindow + kChunk + DeflateDecoder::kSlack + 64);
            if (!c.buf) return false;
        }
        return true;
    }
    void start() {
        if (started_) return;
        started_ = true;
        producer_ = std::thread([this] { produce(); });
        checker_ = std::thread([this] { check(); });
    }
    static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

    // wait for ring slot `seq` to be free; false: shutting down
    Chunk* acquire(uint64_t seq) {
        std::unique_lock<std::mutex> g(mu_);
        Chunk& c = ring_[seq % kRing];
        cv_.wait(g, [&] { return c.state == kFree || quit_; });
        if (quit_) return nullptr;
        c.len = 0;
        c.seq = seq;
        c.checked = false;
        c.last = false;
        c.segs.clear();
        return &c;
    }
    void publish(Chunk* c, bool last) {
        {
            std::lock_guard<std::mutex> g(mu_);
            c->last = last;
            c->state = kFilled;
        }
        cv_.notify_all();
    }

    void produce() {
        DeflateDecoder* dec = new DeflateDecoder();
        const size_t W = DeflateDecoder::kWindow;
        uint64_t seq = 0;
        Chunk* c = acquire(seq);
        size_t at = 0;  // file offset of the next member
        bool failed = false;
        while (c && !failed) {
            const size_t hdr = at < size_ ? gzip_header_bytes(file_ + at, size_ - at) : 0;
            if (!hdr) break;  // the end, or bytes that do not begin a member: ignored, as gzread does
            dec->reset(file_ + at + hdr, file_ + size_);
            size_t floor = W + c->len;  // earliest byte of the chunk buffer a match of this member may reach
            size_t seg_start = c->len;
            uint64_t member_bytes = 0;
            for (;;) {
                size_t pos = W + c->len - floor;
                const int r = dec->decode(c->buf + floor, pos, W + kChunk - floor);
                const size_t new_len = floor + pos - W;
                member_bytes += new_len - c->len;
                c->len = new_len;
                if (r == 0) {  // chunk full: the next one begins with this one's last 32 KiB
                    c->segs.push_back({seg_start, c->len - seg_start, false, 0, 0});
                    Chunk* next = acquire(seq + 1);
                    if (!next) {
                        c = nullptr;
                        break;
                    }
                    const size_t h = (size_t)std::min<uint64_t>(W, member_bytes);
                    memcpy(next->buf + W - h, c->buf + W + c->len - h, h);
                    floor = W - h;
                    publish(c, false);
                    c = next;
                    ++seq;
                    seg_start = 0;
                    continue;
                }
                if (r < 0) {
                    failed = true;
                    break;
                }
                // member complete: its trailer
                const uint8_t* e = dec->stream_end();
                if ((size_t)(file_ + size_ - e) < 8) {
                    failed = true;
                    break;
                }
                c->segs.push_back({seg_start, c->len - seg_start, true, le32(e), le32(e + 4)});
                at = (size_t)(e - file_) + 8;
                break;
            }
            if (c && !failed && c->len >= kChunk) {  // (a member ended exactly at the chunk's end)
                Chunk* next = acquire(seq + 1);
                if (!next) {
                    c = nullptr;
                    break;
                }
                publish(c, false);
                c = next;
                ++seq;
            }
        }
        if (c) {
            if (failed) {
                if (c->len > (c->segs.empty() ? 0 : c->segs.back().off + c->segs.back().len)) {
                    const size_t s0 = c->segs.empty() ? 0 : c->segs.back().off + c->segs.back().len;
                    c->segs.push_back({s0, c->len - s0, false, 0, 0});
                }
                error_ = true;
            }
            publish(c, true);
        }
        delete dec;
    }

    void check() {
        uint64_t seq = 0;
        uint32_t crc = 0;
        uint64_t isize = 0;
        for (;;) {
            Chunk* c;
            {
                std::unique_lock<std::mutex> g(mu_);
                c = &ring_[seq % kRing];
                cv_.wait(g, [&] { return (c->state == kFilled && c->seq == seq && !c->checked) || quit_; });
                if (quit_) return;
            }
            for (const Segment& s : c->segs) {
                crc = crc32_slice16(crc, c->buf + DeflateDecoder::kWindow + s.off, s.len);
                isize += s.len;
                if (s.ends_member) {
                    if (crc != s.crc || (uint32_t)isize != s.isize) error_ = true;
                    crc = 0;
                    isize = 0;
                }
            }
            const bool last = c->last;
            {
                std::lock_guard<std::mutex> g(mu_);
                c->checked = true;
            }
            cv_.notify_all();
            if (last) return;
            ++seq;
        }
    }

    const uint8_t* file_;
    size_t size_;
    Chunk ring_[kRing];
    std::mutex mu_;
    std::condition_variable cv_;
    std::thread producer_, checker_;
    bool quit_ = false;
    volatile bool error_ = false;
    // reader side
    Chunk* cur_ = nullptr;
    size_t cur_off_ = 0;
    uint64_t read_seq_ = 0;
    bool done_ = false, started_ = false;
};

}  // namespace lphq
