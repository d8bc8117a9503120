// fastx.hpp -- FASTA/FASTQ(.gz) record reader with the record semantics of the reference's kseq
// (external/kseq.h:192-240): a header starts at any '>' or '@' once the previous record is complete; the
// sequence is every following line, concatenated, until a line that starts with '>', '@' or '+'; '\r' before
// '\n' is dropped; a FASTQ quality string must have the length of the sequence or reading stops.
// Pinned against the real kseq.h on crafted and fuzzed inputs (oracle/kseq_dump.c, tests/test_host_cpu.py).
//
// FastxReader::next() follows the control flow of kseq_read step by step (its corner cases are observable behaviour), so
// this file carries klib's notice for that function:
//   kseq.h -- The MIT License.  Copyright (c) 2008, 2009, 2011 Attractive Chaos <attractor@live.co.uk>
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//   documentation files (the "Software"), to deal in the Software without restriction, including without limitation the
//   rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and to permit
//   persons to whom the Software is furnished to do so, subject to the following conditions: The above copyright notice and
//   this permission notice shall be included in all copies or substantial portions of the Software.
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE
//   WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT.  IN NO EVENT SHALL THE AUTHORS OR
//   COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR
//   OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
#pragma once
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gz_source.hpp"

namespace lphq {

class FastxReader {
public:
    // A regular .gz file is inflated ahead of the parser by its own threads (gz_source.hpp); anything else -- and everything when
    // LPHQ_GZ_ZLIB is set (A/B timing, tests) -- goes through zlib's gzread, which also passes plain text through.
    explicit FastxReader(const char* path) : fp_(gzopen(path, "r")), buf_(1 << 22) {
        if (fp_) gzbuffer(fp_, 1 << 20);
        if (fp_ && !getenv("LPHQ_GZ_ZLIB")) fast_ = GzFastSource::open(path);
    }
    ~FastxReader() {
        delete fast_;
        if (fp_) gzclose(fp_);
    }
    FastxReader(const FastxReader&) = delete;
    FastxReader& operator=(const FastxReader&) = delete;
    bool ok() const { return fp_ != nullptr; }
    // Restart between records at an (uncompressed) byte offset; the next record is the first one that begins at or after it.
    bool seek(size_t offset) {
        delete fast_;  // (seeking is for plain files handed over by the parallel reader; zlib does it from here on)
        fast_ = nullptr;
        if (!fp_ || gzseek(fp_, (z_off_t)offset, SEEK_SET) < 0) return false;
        pos_ = end_ = 0;
        eof_ = false;
        last_char_ = 0;
        return true;
    }

    // Next record's sequence into seq; false at end of input or on a malformed FASTQ record (kseq_read < 0).
    // Operation by operation what kseq_read does (external/kseq.h:192-240), because its corner cases are observable:
    // a last header without a line end still yields an (empty) record if anything follows the header character; the
    // '\r' before a line end is dropped only once the string holds more than one character, and not at all when the
    // line's first character is the last byte of the input; a '+' line always consumes one quality line, even for an
    // empty sequence.
    bool next(std::string& seq) {
        int c;
        if (last_char_ == 0) {  // jump to the next header line
            while ((c = getc()) >= 0 && c != '>' && c != '@') {
            }
            if (c < 0) return false;
            last_char_ = c;
        }  // else: the first header character was read by the previous call
        seq.clear();
        int delim = 0;
        if (get_until(kSpace, scratch_, &delim, false) < 0) return false;  // name; nothing at all after the header character: EOF
        if (delim != '\n') (void)get_until(kLine, scratch_, nullptr, false);  // comment: the rest of the header line
        while ((c = getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;  // skip empty lines
            seq.push_back((char)c);
            (void)get_until(kLine, seq, nullptr, true);  // the rest of the line
        }
        if (c == '>' || c == '@') last_char_ = c;  // the first header character has been read
        if (c != '+') return true;                 // FASTA
        while ((c = getc()) >= 0 && c != '\n') {  // skip the rest of the '+' line
        }
        if (c < 0) return false;  // no quality string (-2)
        qual_.clear();
        while (get_until(kLine, qual_, nullptr, true) >= 0 && qual_.size() < seq.size()) {
        }
        last_char_ = 0;  // we have not come to the next header line
        return qual_.size() == seq.size();  // (-2 otherwise: the caller's read loop stops)
    }

private:
    enum Delim { kSpace, kLine };
    // the next stretch of text: data_[0, n); the fast source lends its own chunk memory (no copy), zlib fills buf_
    int fill() {
        if (fast_) {
            size_t n = 0;
            data_ = fast_->next_span(n, (size_t)1 << 30);
            if (!n && fast_->error() && !warned_) {  // (gzread would return -1 here; the records read so far stand, as with kseq)
                warned_ = true;
                fprintf(stderr, "[warning] the gzip stream is damaged or truncated: input ends early\n");
            }
            return (int)n;
        }
        data_ = buf_.data();
        return gzread(fp_, buf_.data(), (unsigned)buf_.size());
    }
    int getc() {
        if (pos_ >= end_) {
            if (eof_) return -1;
            int n = fill();
            pos_ = 0;
            end_ = n > 0 ? (size_t)n : 0;
            if (n <= 0) {
                eof_ = true;
                return -1;
            }
        }
        return (unsigned char)data_[pos_++];
    }
    // ks_getuntil2 (external/kseq.h:99-158): append (or assign) the bytes up to the next delimiter -- any isspace()
    // character for kSpace, '\n' for kLine -- consume the delimiter and report it in *dret (0 if the input ended first).
    // Returns -1 if the input was already exhausted, else the length of str; for kLine a trailing '\r' is dropped when
    // str holds more than one character.
    long get_until(Delim d, std::string& str, int* dret, bool append) {
        bool gotany = false;
        if (dret) *dret = 0;
        if (!append) str.clear();
        for (;;) {
            if (pos_ >= end_) {
                if (eof_) break;
                int n = fill();
                pos_ = 0;
                end_ = n > 0 ? (size_t)n : 0;
                if (n <= 0) {
                    eof_ = true;
                    break;
                }
            }
            const char* p = data_ + pos_;
            size_t i;
            if (d == kLine) {
                const char* nl = (const char*)memchr(p, '\n', end_ - pos_);
                i = nl ? (size_t)(nl - p) : end_ - pos_;
            } else {
                for (i = 0; i < end_ - pos_; ++i) {
                    const unsigned char ch = (unsigned char)p[i];
                    if (ch == ' ' || (ch >= '\t' && ch <= '\r')) break;  // isspace() in the C locale
                }
            }
            gotany = true;
            str.append(p, i);
            const bool hit = i < end_ - pos_;
            if (hit && dret) *dret = (unsigned char)p[i];
            pos_ += i + (hit ? 1 : 0);
            if (hit) break;
        }
        if (!gotany) return -1;
        if (d == kLine && str.size() > 1 && str.back() == '\r') str.pop_back();
        return (long)str.size();
    }
    std::string scratch_, qual_;
    gzFile fp_;
    GzFastSource* fast_ = nullptr;
    std::vector<char> buf_;
    const char* data_ = nullptr;
    bool warned_ = false;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
    int last_char_ = 0;
};

}  // namespace lphq
