// fastx.hpp -- FASTA/FASTQ(.gz) record reader with the record semantics of the reference's kseq
// (external/kseq.h:192-240): a header starts at any '>' or '@' once the previous record is complete; the
// sequence is every following line, concatenated, until a line that starts with '>', '@' or '+'; '\r' before
// '\n' is dropped; a FASTQ quality string must have the length of the sequence or reading stops.
#pragma once
#include <zlib.h>

#include <cstring>
#include <string>
#include <vector>

namespace lphq {

class FastxReader {
public:
    explicit FastxReader(const char* path) : fp_(gzopen(path, "r")), buf_(1 << 22) {
        if (fp_) gzbuffer(fp_, 1 << 20);
    }
    ~FastxReader() {
        if (fp_) gzclose(fp_);
    }
    bool ok() const { return fp_ != nullptr; }
    // Restart between records at an (uncompressed) byte offset; the next record is the first one that begins at or after it.
    bool seek(size_t offset) {
        if (!fp_ || gzseek(fp_, (z_off_t)offset, SEEK_SET) < 0) return false;
        pos_ = end_ = 0;
        eof_ = false;
        last_char_ = 0;
        return true;
    }

    // Next record's sequence into seq; false at end of input or on a malformed FASTQ record (kseq_read < 0).
    bool next(std::string& seq) {
        int c;
        if (last_char_ == 0) {
            while ((c = getc()) >= 0 && c != '>' && c != '@') {
            }
            if (c < 0) return false;
            last_char_ = c;
        }
        seq.clear();
        // name up to the first white space, then the rest of the header line (comment)
        while ((c = getc()) >= 0 && c != ' ' && c != '\t' && c != '\n' && c != '\v' && c != '\f' && c != '\r') {
        }
        if (c < 0) return false;  // header without a line end: kseq returns EOF here
        if (c != '\n') skip_line();
        while ((c = getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            seq.push_back((char)c);
            append_line(seq);
        }
        if (c == '>' || c == '@') last_char_ = c;
        if (c != '+') return true;  // FASTA
        skip_line();
        size_t qlen = 0;
        bool eof = false;
        while (qlen < seq.size() && !eof) qlen += count_line(eof);
        last_char_ = 0;
        return qlen == seq.size();
    }

private:
    int getc() {
        if (pos_ >= end_) {
            if (eof_) return -1;
            int n = gzread(fp_, buf_.data(), (unsigned)buf_.size());
            pos_ = 0;
            end_ = n > 0 ? (size_t)n : 0;
            if (n <= 0) {
                eof_ = true;
                return -1;
            }
        }
        return (unsigned char)buf_[pos_++];
    }
    // Rest of the current line, a buffer span at a time (memchr): visit(ptr, len) per span; returns false at end of input.
    template <typename Visit>
    bool rest_of_line(Visit visit) {
        for (;;) {
            if (pos_ >= end_) {
                int c = getc();  // refill
                if (c < 0) return false;
                --pos_;
            }
            const char* p = buf_.data() + pos_;
            const char* nl = (const char*)memchr(p, '\n', end_ - pos_);
            const size_t n = nl ? (size_t)(nl - p) : end_ - pos_;
            visit(p, n);
            pos_ += n + (nl ? 1 : 0);
            if (nl) return true;
        }
    }
    void skip_line() {
        rest_of_line([](const char*, size_t) {});
    }
    void append_line(std::string& s) {
        rest_of_line([&](const char* p, size_t n) { s.append(p, n); });
        if (s.size() > 1 && s.back() == '\r') s.pop_back();
    }
    size_t count_line(bool& eof) {
        size_t n = 0;
        int last = 0;
        if (!rest_of_line([&](const char* p, size_t len) {
                n += len;
                if (len) last = (unsigned char)p[len - 1];
            }))
            eof = true;
        if (n > 1 && last == '\r') --n;
        return n;
    }
    gzFile fp_;
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
    int last_char_ = 0;
};

}  // namespace lphq
