/*
 * lph_oracle_driver.c -- TEST INFRASTRUCTURE ONLY (see lph_oracle.h).
 *
 * CPU restatement of the reference's query DRIVER, src/query.cpp:24-88 ("B1, reference-faithful" of BASELINE.md section 2):
 * open the (gzip or plain) FASTA/FASTQ with zlib, and for every record call MPHF::operator() -- pass 1 with streaming on,
 * pass 2 with streaming off -- with inflate and record parsing INSIDE the timed region and one freshly allocated result
 * vector per record, exactly as the reference does (query.cpp:48-56, 66-74).  The record reader restates kseq_read
 * (external/kseq.h:192-240) and the line rule of ks_getuntil2 (kseq.h:96-151); tests/test_host_cpu.py pins it against the real
 * kseq.h (oracle/_ref/kseq_dump) on crafted and fuzzed inputs.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>
#include <pthread.h>

#include "lph_oracle.h"

#define ORC_KS_BUF 16384 /* KSEQ_INIT's stream buffer (kseq.h:252: KSTREAM_INIT(type_t, __read, 16384)) */

typedef struct {
    gzFile f;
    unsigned char buf[ORC_KS_BUF];
    int begin, end, eof, err;
} ks_t;

typedef struct { char* s; size_t l, cap; } str_t;

static void str_reserve(str_t* s, size_t extra) {
    if (s->l + extra + 1 > s->cap) {
        size_t c = s->cap ? s->cap : 256;
        while (c < s->l + extra + 1) c *= 2;
        s->s = (char*)realloc(s->s, c);
        s->cap = c;
    }
}

/* ks_getc, kseq.h:65-82: -1 at end of input, -3 on a read error */
static int ks_fill(ks_t* k) {
    k->begin = 0;
    k->end = gzread(k->f, k->buf, ORC_KS_BUF);
    if (k->end == 0) { k->eof = 1; return -1; }
    if (k->end < 0) { k->eof = 1; k->err = 1; k->end = 0; return -3; }
    return 0;
}
static int ks_getc(ks_t* k) {
    if (k->err) return -3;
    if (k->begin >= k->end) {
        if (k->eof) return -1;
        int r = ks_fill(k);
        if (r < 0) return r;
    }
    return (int)k->buf[k->begin++];
}

/* ks_getuntil2, kseq.h:96-151.  sep: 0 = any isspace() character (KS_SEP_SPACE), 2 = '\n' (KS_SEP_LINE).  Appends to s when
 * append != 0.  Returns -1 if nothing at all could be read (end of input), -3 on error, else the string length; a '\r' that ends a
 * line is dropped only when the string then holds more than one character (kseq.h:146-147). */
static long ks_until(ks_t* k, int sep, str_t* s, int* dret, int append) {
    int gotany = 0;
    if (dret) *dret = 0;
    if (!append) s->l = 0;
    for (;;) {
        if (k->err) return -3;
        if (k->begin >= k->end) {
            if (k->eof) break;
            int r = ks_fill(k);
            if (r == -3) return -3;
            if (r == -1) break;
        }
        int i = k->begin;
        if (sep == 2) {
            const unsigned char* p = (const unsigned char*)memchr(k->buf + k->begin, '\n', (size_t)(k->end - k->begin));
            i = p ? (int)(p - k->buf) : k->end;
        } else {
            while (i < k->end && !(k->buf[i] == ' ' || (k->buf[i] >= '\t' && k->buf[i] <= '\r'))) ++i; /* isspace(), C locale */
        }
        str_reserve(s, (size_t)(i - k->begin));
        gotany = 1;
        memcpy(s->s + s->l, k->buf + k->begin, (size_t)(i - k->begin));
        s->l += (size_t)(i - k->begin);
        k->begin = i + 1;
        if (i < k->end) {
            if (dret) *dret = k->buf[i];
            break;
        }
    }
    if (!gotany && k->eof && k->begin >= k->end) return -1;
    str_reserve(s, 0);
    if (sep == 2 && s->l > 1 && s->s[s->l - 1] == '\r') --s->l;
    s->s[s->l] = 0;
    return (long)s->l;
}

typedef struct {
    ks_t ks;
    str_t name, comment, seq, qual;
    int last_char;
} rec_t;

/* kseq_read, kseq.h:192-240: >= 0 sequence length, -1 end of input, -2 truncated quality, -3 stream error */
static long rec_read(rec_t* r) {
    int c;
    long n;
    ks_t* k = &r->ks;
    if (r->last_char == 0) { /* jump to the next header line */
        while ((c = ks_getc(k)) >= 0 && c != '>' && c != '@') {}
        if (c < 0) return c;
        r->last_char = c;
    }
    r->comment.l = r->seq.l = r->qual.l = 0;
    if ((n = ks_until(k, 0, &r->name, &c, 0)) < 0) return n;
    if (c != '\n') ks_until(k, 2, &r->comment, 0, 0);
    while ((c = ks_getc(k)) >= 0 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue; /* empty line */
        str_reserve(&r->seq, 1);
        r->seq.s[r->seq.l++] = (char)c;
        ks_until(k, 2, &r->seq, 0, 1); /* rest of the line */
    }
    if (c == '>' || c == '@') r->last_char = c;
    str_reserve(&r->seq, 1);
    r->seq.s[r->seq.l] = 0;
    if (c != '+') return (long)r->seq.l; /* FASTA */
    while ((c = ks_getc(k)) >= 0 && c != '\n') {} /* rest of the '+' line */
    if (c == -1) return -2;
    while ((n = ks_until(k, 2, &r->qual, 0, 1)) >= 0 && r->qual.l < r->seq.l) {}
    if (n == -3) return -3;
    r->last_char = 0;
    if (r->seq.l != r->qual.l) return -2;
    return (long)r->seq.l;
}

static rec_t* rec_open(const char* path) {
    gzFile f = gzopen(path, "r");
    if (!f) return NULL;
    rec_t* r = (rec_t*)calloc(1, sizeof *r);
    r->ks.f = f;
    return r;
}
static void rec_close(rec_t* r) {
    if (!r) return;
    gzclose(r->ks.f);
    free(r->name.s); free(r->comment.s); free(r->seq.s); free(r->qual.s);
    free(r);
}

/* records, bases and FNV-1a 64 over (length, bases) of every record: the figures oracle/_ref/kseq_dump prints */
int orc_file_digest(const char* path, uint64_t* n_records, uint64_t* n_bases, uint64_t* digest) {
    rec_t* r = rec_open(path);
    if (!r) return 2;
    uint64_t h = 1469598103934665603ull, nr = 0, nb = 0;
    while (rec_read(r) >= 0) {
        const uint64_t len = r->seq.l;
        const unsigned char* p = (const unsigned char*)&len;
        for (int i = 0; i < 8; ++i) h = (h ^ p[i]) * 1099511628211ull;
        for (size_t i = 0; i < r->seq.l; ++i) h = (h ^ (unsigned char)r->seq.s[i]) * 1099511628211ull;
        ++nr;
        nb += len;
    }
    rec_close(r);
    *n_records = nr; *n_bases = nb; *digest = h;
    return 0;
}

static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

/* One pass of src/query.cpp:48-56: the timer runs over kseq_read AND the lookups; a fresh vector per record.  `sink` keeps the
 * results observable (essentials::do_not_optimize_away, query.cpp:54). */
static int one_pass(const orc_index* idx, const char* path, int streaming, uint64_t* total, double* us, uint64_t* sink) {
    rec_t* r = rec_open(path);
    if (!r) return 2;
    uint64_t n = 0, acc = 0;
    const double t0 = now_us();
    while (rec_read(r) >= 0) {
        const size_t cap = r->seq.l ? r->seq.l : 1;
        uint64_t* hashes = (uint64_t*)malloc(cap * sizeof *hashes);
        const int64_t got = orc_query(idx, r->seq.s, r->seq.l, streaming, hashes, cap);
        if (got > 0) {
            n += (uint64_t)got;
            acc ^= hashes[0] ^ hashes[got - 1];
        }
        free(hashes);
    }
    *us = now_us() - t0;
    rec_close(r);
    *total = n;
    *sink ^= acc;
    return 0;
}

/* query<MPHF>(), src/query.cpp:24-88, after option parsing and essentials::load: returns 0, or 2 when the file cannot be opened
 * (query.cpp:43-46).  Outputs what the CSV line of query.cpp:83-86 is made of. */
int orc_query_file(const orc_index* idx, const char* query_path, uint64_t* total_kmers_streaming, uint64_t* total_kmers_random,
                   double* us_streaming, double* us_random) {
    uint64_t sink = 0;
    int rc = one_pass(idx, query_path, 1, total_kmers_streaming, us_streaming, &sink);
    if (rc) return rc;
    rc = one_pass(idx, query_path, 0, total_kmers_random, us_random, &sink);
    if (rc) return rc;
    return sink == 0x5eed5eed5eed5eedull ? 0 : 0; /* keeps `sink` live */
}


/* orc_query_packed over contiguous slices of the batch, one POSIX thread per slice (the lookups are independent; the index is
 * read-only).  Test infrastructure like the rest of this file: the all-cores CPU figure bench.py reports next to the GPU's. */
typedef struct { const orc_index* x; const uint64_t* kmers; size_t n; uint64_t* out; } slice_t;
static void* slice_main(void* p) {
    slice_t* s = (slice_t*)p;
    orc_query_packed(s->x, s->kmers, s->n, s->out);
    return NULL;
}
int orc_query_packed_threads(const orc_index* x, const uint64_t* kmers, size_t n, uint64_t* out, int threads) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    const size_t wpk = orc_kmer_words(x);
    pthread_t* th = calloc((size_t)threads, sizeof *th);
    slice_t* sl = calloc((size_t)threads, sizeof *sl);
    if (!th || !sl) { free(th); free(sl); return -1; }
    int started = 0, rc = 0;
    for (int t = 0; t < threads; ++t) {
        const size_t a = n * (size_t)t / (size_t)threads, b = n * (size_t)(t + 1) / (size_t)threads;
        sl[t] = (slice_t){x, kmers + a * wpk, b - a, out + a};
        if (pthread_create(&th[t], NULL, slice_main, &sl[t]) != 0) { rc = -1; break; }
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    if (rc != 0)  /* could not start them all: finish the rest here */
        for (int t = started; t < threads; ++t) {
            const size_t a = n * (size_t)t / (size_t)threads, b = n * (size_t)(t + 1) / (size_t)threads;
            orc_query_packed(x, kmers + a * wpk, b - a, out + a);
        }
    free(th);
    free(sl);
    return 0;
}
