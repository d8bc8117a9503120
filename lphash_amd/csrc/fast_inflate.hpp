// fast_inflate.hpp -- a DEFLATE (RFC 1951) decoder and the gzip (RFC 1952) member framing around it, for a single-member .gz (the format
// the reference's usage text names, src/query.cpp:14-22).  zlib 1.2.11's inflate ran at ~490 MB/s of text on the GPU box and made the whole
// `lphash query-*` call 2 ns per k-mer.  This decoder works on the whole compressed file in memory (mmap), which removes the input refill
// logic, and is written for throughput; it can start at any bit of the file and stop at the first block border at or after a given bit, and
// it can write 16-bit symbols instead of bytes (a match then copies whatever stands in the 32 Ki symbols before it, known byte or
// placeholder) -- the two things parallel_gz.hpp needs to decode one stream with many threads.  Per thread:
//   * a 64-bit bit buffer refilled with one unaligned 8-byte load (no per-byte loop);
//   * table-driven Huffman decoding: an 11-bit primary table for literals / lengths and an 8-bit one for distances, longer codes through
//     second-level tables; one table entry carries the symbol's value (literal, length base, distance base), its extra-bit count and its
//     codeword length, so a symbol costs one lookup (two for the rare long codes);
//   * matches copied eight bytes at a time when the distance allows, into an output buffer with slack so that no copy needs a bound check.
// Output goes to large chunks, each preceded by the last 32 KiB of the one before (the match window).  Everything here is written from the
// two RFCs; zlib is the checker in the tests (tests/test_host_cpu.py: byte equality on stored / fixed / dynamic blocks, every header flag,
// several members, truncated and corrupted streams) and stays the fallback for anything this path does not take (non-regular files).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace lphq {

class DeflateDecoder {
public:
    static constexpr size_t kWindow = 32768;  // bytes of history a match may reach back
    static constexpr size_t kSlack = 320;     // room past `limit` a call may write into: one match (258) and an 8-byte copy step, rounded up

    void reset(const uint8_t* in, const uint8_t* in_end) {
        base_ = in_ = in;
        end_ = in_end;
        bitbuf_ = 0;
        bitsleft_ = 0;
        state_ = kHeader;
        final_ = false;
        stored_left_ = 0;
        stop_bit_ = ~(uint64_t)0;
        blocks_ = 0;
    }
    // start at bit `bit` of the data at `base` (bit 0 = least significant bit of base[0]: the order in which DEFLATE packs them)
    void reset_at_bit(const uint8_t* base, uint64_t bit, const uint8_t* in_end) {
        reset(base + (bit >> 3), in_end);
        base_ = base;
        if ((bit & 7) && need((int)(bit & 7))) take((int)(bit & 7));
    }
    // position of the next unread bit, counted like reset_at_bit's argument
    uint64_t bit_position() const { return (uint64_t)(in_ - base_) * 8 - (uint64_t)bitsleft_; }
    // decode() returns 3 instead of reading a block header at or beyond this bit
    void set_stop_bit(uint64_t bit) { stop_bit_ = bit; }
    // block headers read so far
    uint32_t blocks_started() const { return blocks_; }
    bool in_final_block() const { return final_; }
    // Decode into buf[pos ...): stops once pos > limit (the caller provides kSlack bytes of room past limit), at the end of the stream, or
    // on an error.  `hist` = valid bytes before buf + pos at the time of the call... a match may reach min(kWindow, bytes before its start).
    // Returns 0: out of room, call again with a new buffer whose first bytes repeat the last kWindow bytes; 1: the final block ended;
    // -1: invalid or truncated stream; 3: the next block would begin at or beyond the stop bit (set_stop_bit).
    int decode(uint8_t* buf, size_t& pos, size_t limit) { return decode_as<uint8_t>(buf, pos, limit); }
    // The same with T = uint16_t: one 16-bit symbol per byte of text.  Literals are 0..255; what a match copies is whatever the caller put (or this
    // decoder wrote) in the symbols before it, so a caller that does not know the 32 KiB before its starting point puts 32 Ki distinct placeholder
    // values >= 256 there and replaces them once the text before is known (parallel_gz.hpp).
    template <typename T>
    int decode_as(T* buf, size_t& pos, size_t limit) {
        for (;;) {
            if (state_ == kHeader) {
                if (final_) return 1;
                if (bit_position() >= stop_bit_) return 3;
                if (!need(3)) return -1;
                final_ = take(1) != 0;
                const uint32_t type = take(2);
                if (type == 0) {
                    // stored: skip to the byte boundary, LEN, NLEN, bytes
                    take(bitsleft_ & 7);
                    in_ -= bitsleft_ >> 3;  // give the whole bytes back
                    bitbuf_ = 0;
                    bitsleft_ = 0;
                    if (end_ - in_ < 4) return -1;
                    const uint32_t len = (uint32_t)in_[0] | ((uint32_t)in_[1] << 8), nlen = (uint32_t)in_[2] | ((uint32_t)in_[3] << 8);
                    if ((len ^ nlen) != 0xFFFFu) return -1;
                    in_ += 4;
                    stored_left_ = len;
                    state_ = kStored;
                } else if (type == 1) {
                    fixed_tables();
                    state_ = kHuffman;
                } else if (type == 2) {
                    if (!dynamic_tables()) return -1;
                    state_ = kHuffman;
                } else {
                    return -1;
                }
                ++blocks_;
            }
            if (state_ == kStored) {
                while (stored_left_) {
                    if (pos > limit) return 0;
                    size_t n = stored_left_;
                    if (n > limit + 1 - pos) n = limit + 1 - pos;
                    if ((size_t)(end_ - in_) < n) return -1;
                    if (sizeof(T) == 1) memcpy(buf + pos, in_, n);
                    else
                        for (size_t i = 0; i < n; ++i) buf[pos + i] = (T)in_[i];
                    pos += n;
                    in_ += n;
                    stored_left_ -= (uint32_t)n;
                }
                state_ = kHeader;
                continue;
            }
            const int r = huffman_block<T>(buf, pos, limit);
            if (r != 2) return r;  // 0: out of room, -1: error
            state_ = kHeader;      // 2: end of block
        }
    }
    // first byte after the deflate stream (valid once decode() returned 1): the bytes the bit buffer holds beyond the last symbol go back
    const uint8_t* stream_end() const { return in_ - (bitsleft_ >> 3); }

protected:
    enum State { kHeader, kStored, kHuffman };
    // table entry: value << 16 | type << 12 | extra << 8 | nbits, nbits = the bits the symbol takes from the stream INCLUDING its extra bits
    // (one shift on the critical path; the extra bits are picked out of a copy of the buffer on the side)
    enum : uint32_t { kLiteral = 0, kBase = 1, kEndOfBlock = 2, kSubtable = 3, kInvalid = 4, kLiterals = 5 /* 2-4 literals from one lookup: value = how many, the bytes in the high word */ };
    static constexpr int kLitBits = 11, kDistBits = 8;
    static uint32_t entry(uint32_t value, uint32_t type, uint32_t extra, uint32_t nbits) { return (value << 16) | (type << 12) | (extra << 8) | nbits; }

    static uint64_t load64(const uint8_t* p) {
        uint64_t v;
        memcpy(&v, p, 8);
        return v;  // little-endian host (x86-64)
    }
    // bit buffer: bit 0 of bitbuf_ is the next bit of the stream
    void refill_fast() {  // needs 8 readable bytes at in_
        bitbuf_ |= load64(in_) << bitsleft_;
        in_ += (63 - bitsleft_) >> 3;
        bitsleft_ |= 56;
    }
    void refill_slow() {
        while (bitsleft_ <= 56 && in_ < end_) {
            bitbuf_ |= (uint64_t)*in_++ << bitsleft_;
            bitsleft_ += 8;
        }
    }
    bool need(int n) {
        if (bitsleft_ < n) refill_slow();
        return bitsleft_ >= n;
    }
    uint32_t take(int n) {  // n <= 32, available
        const uint32_t v = (uint32_t)(bitbuf_ & ((1ull << n) - 1));
        bitbuf_ >>= n;
        bitsleft_ -= n;
        return v;
    }

    // ---- Huffman tables ----
    // lens[0 .. n): code lengths (0 = unused).  kind 0: literal/length alphabet, 1: distances.  false: over-subscribed.
    bool build(const uint8_t* lens, int n, int kind) {
        static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dist_base[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                               193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        const int tb = kind == 0 ? kLitBits : kDistBits;
        auto get = [&](uint32_t i) -> uint32_t { return kind == 0 ? (uint32_t)lit_[i] : dist_[i]; };
        auto set = [&](uint32_t i, uint32_t e) {
            if (kind == 0) lit_[i] = e;
            else dist_[i] = e;
        };
        const uint32_t tsize = 1u << tb;
        int count[16] = {0};
        for (int i = 0; i < n; ++i) count[lens[i]]++;
        count[0] = 0;
        // canonical codes: first code of every length; reject an over-subscribed set
        uint32_t next_code[16];
        uint32_t code = 0;
        int64_t left = 1;
        for (int l = 1; l <= 15; ++l) {
            left = (left << 1) - count[l];
            if (left < 0) return false;
            code = (code + (uint32_t)count[l - 1]) << 1;
            next_code[l] = code;
        }
        for (uint32_t i = 0; i < tsize; ++i) set(i, entry(0, kInvalid, 0, 1));
        // symbol -> entry (without the codeword length)
        auto symbol_entry = [&](int sym, uint32_t nbits) -> uint32_t {
            if (kind == 0) {
                if (sym < 256) return entry((uint32_t)sym, kLiteral, 0, nbits);
                if (sym == 256) return entry(0, kEndOfBlock, 0, nbits);
                if (sym - 257 < 29) return entry(len_base[sym - 257], kBase, len_extra[sym - 257], nbits + len_extra[sym - 257]);
                return entry(0, kInvalid, 0, nbits);  // 286, 287: never valid in the data
            }
            if (sym < 30) return entry(dist_base[sym], kBase, dist_extra[sym], nbits + dist_extra[sym]);
            return entry(0, kInvalid, 0, nbits);  // 30, 31
        };
        auto reverse = [](uint32_t c, int l) {
            uint32_t r = 0;
            for (int i = 0; i < l; ++i) r |= ((c >> i) & 1u) << (l - 1 - i);
            return r;
        };
        // pass 1: short codes straight into the primary table; for long ones, the longest code behind every primary prefix
        uint8_t longest[1u << kLitBits];
        memset(longest, 0, tsize);
        uint32_t codes[288];
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t r = reverse(next_code[l]++, l);
            codes[s] = r;
            if (l <= tb) {
                const uint32_t e = symbol_entry(s, (uint32_t)l);
                for (uint32_t i = r; i < tsize; i += 1u << l) set(i, e);
            } else {
                const uint32_t p = r & (tsize - 1);
                if ((int)longest[p] < l) longest[p] = (uint8_t)l;
            }
        }
        // pass 2: a second-level table behind every prefix that has long codes
        uint32_t next_free = tsize;
        for (uint32_t p = 0; p < tsize; ++p) {
            if (!longest[p]) continue;
            const uint32_t sub = (uint32_t)longest[p] - (uint32_t)tb;
            set(p, entry(next_free, kSubtable, sub, (uint32_t)tb));
            for (uint32_t i = 0; i < (1u << sub); ++i) set(next_free + i, entry(0, kInvalid, 0, 1));
            next_free += 1u << sub;
        }
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (l <= tb) continue;
            const uint32_t r = codes[s], p = r & (tsize - 1);
            const uint32_t start = get(p) >> 16, sub = (get(p) >> 8) & 15u;
            const uint32_t e = symbol_entry(s, (uint32_t)(l - tb));
            for (uint32_t i = r >> tb; i < (1u << sub); i += 1u << (l - tb)) set(start + i, e);
        }
        if (kind == 0) {
            // Up to four literals per lookup where their codewords fit into the index together: DNA text is almost all literals of 2-3 bits,
            // and the cost of a symbol is the lookup, not the bits.  A following literal is read where the bits of the ones before end; it is
            // determined by the index bits alone iff its codeword is no longer than what is left of them.
            uint32_t single[1u << kLitBits];
            for (uint32_t i = 0; i < tsize; ++i) single[i] = (uint32_t)lit_[i];
            for (uint32_t i = 0; i < tsize; ++i) {
                const uint32_t a = single[i];
                if (((a >> 12) & 15u) != kLiteral) continue;
                uint32_t used = a & 0xFFu, count = 1;
                uint64_t bytes = a >> 16;
                while (count < 4 && used < (uint32_t)tb) {
                    const uint32_t b = single[i >> used];
                    if (((b >> 12) & 15u) != kLiteral || (b & 0xFFu) > (uint32_t)tb - used) break;
                    bytes |= (uint64_t)(b >> 16) << (8 * count);
                    used += b & 0xFFu;
                    ++count;
                }
                if (count > 1) lit_[i] = (bytes << 32) | entry(count, kLiterals, 0, used);
            }
        }
        return true;
    }
    void fixed_tables() {
        uint8_t lens[288];
        for (int i = 0; i < 144; ++i) lens[i] = 8;
        for (int i = 144; i < 256; ++i) lens[i] = 9;
        for (int i = 256; i < 280; ++i) lens[i] = 7;
        for (int i = 280; i < 288; ++i) lens[i] = 8;
        (void)build(lens, 288, 0);
        for (int i = 0; i < 32; ++i) lens[i] = 5;
        (void)build(lens, 32, 1);
    }
    bool dynamic_tables() {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        if (!need(14)) return false;
        const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
        if (hlit > 286 || hdist > 30) return false;
        uint8_t cl[19] = {0};
        for (uint32_t i = 0; i < hclen; ++i) {
            if (!need(3)) return false;
            cl[order[i]] = (uint8_t)take(3);
        }
        // the code-length code: at most 7 bits, decoded with a small direct table
        uint16_t cltab[128];
        {
            int count[8] = {0};
            for (int i = 0; i < 19; ++i) count[cl[i]]++;
            count[0] = 0;
            uint32_t next_code[8], code = 0;
            int left = 1;
            for (int l = 1; l <= 7; ++l) {
                left = (left << 1) - count[l];
                if (left < 0) return false;
                code = (code + (uint32_t)count[l - 1]) << 1;
                next_code[l] = code;
            }
            for (int i = 0; i < 128; ++i) cltab[i] = 0xFFFF;
            for (int s = 0; s < 19; ++s) {
                const int l = cl[s];
                if (!l) continue;
                const uint32_t c = next_code[l]++;
                uint32_t r = 0;
                for (int i = 0; i < l; ++i) r |= ((c >> i) & 1u) << (l - 1 - i);
                for (uint32_t i = r; i < 128; i += 1u << l) cltab[i] = (uint16_t)((s << 4) | l);
            }
        }
        uint8_t lens[286 + 30 + 138];
        uint32_t n = 0;
        while (n < hlit + hdist) {
            if (!need(7 + 7)) {
                if (bitsleft_ < 1) return false;  // (the last symbols may sit in the final bits of the input)
            }
            const uint16_t e = cltab[bitbuf_ & 127];
            if (e == 0xFFFF || (int)(e & 15) > bitsleft_) return false;
            take(e & 15);
            const uint32_t sym = e >> 4;
            if (sym < 16) {
                lens[n++] = (uint8_t)sym;
            } else {
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (n == 0 || bitsleft_ < 2) return false;
                    val = lens[n - 1];
                    rep = 3 + take(2);
                } else if (sym == 17) {
                    if (bitsleft_ < 3) return false;
                    rep = 3 + take(3);
                } else {
                    if (bitsleft_ < 7) return false;
                    rep = 11 + take(7);
                }
                if (n + rep > hlit + hdist) return false;
                while (rep--) lens[n++] = (uint8_t)val;
            }
        }
        if (lens[256] == 0) return false;  // no end-of-block code
        if (!build(lens, (int)hlit, 0)) return false;
        return build(lens + hlit, (int)hdist, 1);
    }

    // ---- one Huffman-coded block: 2 = end of block, 0 = out of room, -1 = error ----
    // four bytes -> four 16-bit symbols
    static uint64_t widen4(uint32_t four) {
        uint64_t x = four;
        x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
        return (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    }
    template <typename T>
    int huffman_block(T* buf, size_t& pos_io, size_t limit) {
        constexpr size_t E = 8 / sizeof(T);  // symbols per 8-byte copy step
        T* out = buf + pos_io;
        T* const stop = buf + limit;  // symbols may start while out <= stop
        const uint32_t lmask = (1u << kLitBits) - 1, dmask = (1u << kDistBits) - 1;
        int rc;
        // ---- the fast loop: bit state in registers; at its top the buffer holds >= 56 bits and the table entry of the next symbol is loaded
        // already (a refill does not change the bits the entry was looked up with), so that the lookup of the next symbol overlaps with the
        // copy of the current match.  It runs while 16 input bytes are left (two 8-byte refills per turn at most) and there is output room.
        if (end_ - in_ >= 16) {
            const uint8_t* in = in_;
            const uint8_t* const in_safe = end_ - 16;
            uint64_t bb = bitbuf_;
            int bl = bitsleft_;
#define LPHQ_REFILL()                     \
    do {                                  \
        bb |= load64(in) << bl;           \
        in += (63 - bl) >> 3;             \
        bl |= 56;                         \
    } while (0)
            LPHQ_REFILL();
            uint64_t e = lit_[bb & lmask];
            rc = 1;  // 1: left the loop without a verdict
            while (out <= stop && in <= in_safe) {
                uint32_t t = ((uint32_t)e >> 12) & 15u;
                if (t == kLiterals || t == kLiteral) {  // up to four lookups of literals (<= 11 bits each) before the next refill
                    int k = 0;
                    do {
                        if (t == kLiterals) {
                            const uint32_t four = (uint32_t)(e >> 32);
                            if (sizeof(T) == 1) {
                                memcpy(out, &four, 4);  // (the bytes beyond the count are overwritten by what follows)
                            } else {
                                const uint64_t w = widen4(four);
                                memcpy(out, &w, 8);
                            }
                            out += ((uint32_t)e >> 16) & 7u;
                        } else {
                            *out++ = (T)(uint8_t)(e >> 16);
                        }
                        bb >>= (e & 0xFFu);
                        bl -= (int)(e & 0xFFu);
                        e = lit_[bb & lmask];
                        t = ((uint32_t)e >> 12) & 15u;
                    } while (++k < 4 && (t == kLiterals || t == kLiteral));
                    LPHQ_REFILL();
                    continue;
                }
                if (t == kSubtable) {
                    bb >>= kLitBits;
                    bl -= kLitBits;
                    e = lit_[((uint32_t)e >> 16) + (uint32_t)(bb & ((1u << (((uint32_t)e >> 8) & 15u)) - 1))];
                    t = ((uint32_t)e >> 12) & 15u;
                    if (t == kLiteral) {
                        *out++ = (T)(uint8_t)(e >> 16);
                        bb >>= (e & 0xFFu);
                        bl -= (int)(e & 0xFFu);
                        LPHQ_REFILL();
                        e = lit_[bb & lmask];
                        continue;
                    }
                }
                if (t == kBase) {
                    uint32_t nb = (uint32_t)e & 0xFFu, xb = ((uint32_t)e >> 8) & 15u;
                    const uint32_t len = ((uint32_t)e >> 16) + ((uint32_t)(bb >> (nb - xb)) & ((1u << xb) - 1));
                    bb >>= nb;
                    bl -= (int)nb;
                    uint32_t d = dist_[bb & dmask];
                    if (((d >> 12) & 15u) == kSubtable) {
                        bb >>= kDistBits;
                        bl -= kDistBits;
                        d = dist_[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 15u)) - 1))];
                    }
                    if (((d >> 12) & 15u) != kBase) {
                        rc = -1;
                        break;
                    }
                    nb = d & 0xFFu;
                    xb = (d >> 8) & 15u;
                    const uint32_t dist = (d >> 16) + ((uint32_t)(bb >> (nb - xb)) & ((1u << xb) - 1));
                    bb >>= nb;
                    bl -= (int)nb;
                    if ((size_t)(out - buf) < dist) {  // a match reaching before the data
                        rc = -1;
                        break;
                    }
                    LPHQ_REFILL();
                    e = lit_[bb & lmask];  // the next symbol's entry travels while the bytes are copied
                    const T* src = out - dist;
                    T* const mend = out + len;
                    if (dist >= E) {
                        memcpy(out, src, 8);
                        memcpy(out + E, src + E, 8);
                        if (len > 2 * E) {
                            out += 2 * E;
                            src += 2 * E;
                            do {
                                memcpy(out, src, 8);
                                out += E;
                                src += E;
                            } while (out < mend);
                        }
                    } else if (dist == 1) {
                        if (sizeof(T) == 1) memset(out, (int)*src, len);
                        else
                            for (uint32_t i = 0; i < len; ++i) out[i] = *src;
                    } else {
                        do *out++ = *src++;
                        while (out < mend);
                    }
                    out = mend;
                    continue;
                }
                if (t == kEndOfBlock) {
                    bb >>= (e & 0xFFu);
                    bl -= (int)(e & 0xFFu);
                    rc = 2;
                    break;
                }
                rc = -1;  // an unused codeword
                break;
            }
#undef LPHQ_REFILL
            in_ = in;
            bitbuf_ = bb;
            bitsleft_ = bl;
            if (rc != 1) {
                pos_io = (size_t)(out - buf);
                return rc;
            }
        }
        // ---- the careful loop: the last bytes of the input, one symbol per turn, every bit accounted for ----
        for (;;) {
            if (out > stop) {
                rc = 0;
                break;
            }
            if (end_ - in_ >= 16) {  // (out of room sent us here, not the end of the input)
                rc = 0;
                break;
            }
            refill_slow();
            uint64_t e = lit_[bitbuf_ & lmask];
            // literal(s) / length / end of block
            if ((((uint32_t)e >> 12) & 15u) == kSubtable) {
                bitbuf_ >>= kLitBits;
                bitsleft_ -= kLitBits;
                e = lit_[((uint32_t)e >> 16) + (uint32_t)(bitbuf_ & ((1u << (((uint32_t)e >> 8) & 15u)) - 1))];
            }
            uint32_t nb = (uint32_t)e & 0xFFu, type = ((uint32_t)e >> 12) & 15u;
            if (type == kLiteral || type == kLiterals) {
                bitbuf_ >>= nb;
                bitsleft_ -= (int)nb;
                if (bitsleft_ < 0) {  // (literals that lie beyond the input end up here as well)
                    rc = -1;
                    break;
                }
                if (type == kLiteral) {
                    *out++ = (T)(uint8_t)(e >> 16);
                } else {
                    for (uint32_t i = 0, cnt = ((uint32_t)e >> 16) & 7u; i < cnt; ++i) *out++ = (T)(uint8_t)(e >> (32 + 8 * i));
                }
                continue;
            }
            if (type == kEndOfBlock) {
                bitbuf_ >>= nb;
                bitsleft_ -= (int)nb;
                rc = bitsleft_ < 0 ? -1 : 2;
                break;
            }
            if (type != kBase) {
                rc = -1;
                break;
            }
            uint32_t xb = ((uint32_t)e >> 8) & 15u;
            const uint32_t len = ((uint32_t)e >> 16) + ((uint32_t)(bitbuf_ >> (nb - xb)) & ((1u << xb) - 1));
            bitbuf_ >>= nb;
            bitsleft_ -= (int)nb;
            refill_slow();
            // distance
            uint32_t d = dist_[bitbuf_ & dmask];
            if (((d >> 12) & 15u) == kSubtable) {
                bitbuf_ >>= kDistBits;
                bitsleft_ -= kDistBits;
                d = dist_[(d >> 16) + (uint32_t)(bitbuf_ & ((1u << ((d >> 8) & 15u)) - 1))];
            }
            if (((d >> 12) & 15u) != kBase) {
                rc = -1;
                break;
            }
            nb = d & 0xFFu;
            xb = (d >> 8) & 15u;
            const uint32_t dist = (d >> 16) + ((uint32_t)(bitbuf_ >> (nb - xb)) & ((1u << xb) - 1));
            bitbuf_ >>= nb;
            bitsleft_ -= (int)nb;
            if (bitsleft_ < 0 || (size_t)(out - buf) < dist) {  // bits that were never there, or a match reaching before the data
                rc = -1;
                break;
            }
            const T* src = out - dist;
            T* const mend = out + len;
            if (dist >= E) {
                do {
                    memcpy(out, src, 8);
                    out += E;
                    src += E;
                } while (out < mend);
                out = mend;
            } else if (dist == 1) {
                if (sizeof(T) == 1) memset(out, (int)*src, len);
                else
                    for (uint32_t i = 0; i < len; ++i) out[i] = *src;
                out = mend;
            } else {
                do *out++ = *src++;
                while (out < mend);
            }
        }
        pos_io = (size_t)(out - buf);
        return rc;
    }

    const uint8_t *in_ = nullptr, *end_ = nullptr, *base_ = nullptr;
    uint64_t stop_bit_ = ~(uint64_t)0;
    uint32_t blocks_ = 0;
    uint64_t bitbuf_ = 0;
    int bitsleft_ = 0;
    State state_ = kHeader;
    bool final_ = false;
    uint32_t stored_left_ = 0;
    uint64_t lit_[(1u << kLitBits) + 288 * 16];  // low word: the entry; high word: the literals of a kLiterals entry
    uint32_t dist_[(1u << kDistBits) + 32 * 128];
};

// gzip member header at p (RFC 1952, section 2.3): bytes it occupies, or 0 if it is not one / incomplete
inline size_t gzip_header_bytes(const uint8_t* p, size_t n) {
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
    const uint8_t flg = p[3];
    size_t at = 10;
    if (flg & 4) {  // FEXTRA
        if (at + 2 > n) return 0;
        at += 2 + ((size_t)p[at] | ((size_t)p[at + 1] << 8));
    }
    for (int f = 8; f <= 16; f <<= 1)  // FNAME, FCOMMENT: zero-terminated
        if (flg & f) {
            while (at < n && p[at]) ++at;
            ++at;
        }
    if (flg & 2) at += 2;  // FHCRC
    return at < n ? at : 0;
}

}  // namespace lphq
