/* kseq_dump.c -- TEST INFRASTRUCTURE ONLY.  Pins the product's FASTA/FASTQ record reader against the REAL reference
 * dependency: this driver includes the reference's own external/kseq.h (from /root/reference, where it lies -- it is
 * never copied into this repo) and reads a file exactly as src/query.cpp:43-56 does:
 *     KSEQ_INIT(gzFile, gzread);  fp = gzopen(path, "r");  seq = kseq_init(fp);  while (kseq_read(seq) >= 0) ...
 * It prints "<records>,<bases>,<FNV-1a 64 of every record's (length, bases)>" -- the first three fields of
 * `lphash records`, so tests/test_host_cpu.py can compare the two readers on crafted and fuzzed inputs.
 * Built by oracle/Makefile into oracle/_ref/kseq_dump only where /root/reference exists (the build container). */
#include <stdint.h>
#include <stdio.h>
#include <zlib.h>

#include "kseq.h"
KSEQ_INIT(gzFile, gzread)

static uint64_t h = 1469598103934665603ull;
static void mix(const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 1099511628211ull;
}

int main(int argc, char** argv) {
    if (argc != 2) return 1;
    gzFile fp = gzopen(argv[1], "r");
    if (!fp) return 2;
    kseq_t* seq = kseq_init(fp);
    uint64_t n_rec = 0, n_bases = 0;
    while (kseq_read(seq) >= 0) {
        const uint64_t len = seq->seq.l;
        mix(&len, 8);
        mix(seq->seq.s, len);
        ++n_rec;
        n_bases += len;
    }
    kseq_destroy(seq);
    gzclose(fp);
    printf("%llu,%llu,%llu\n", (unsigned long long)n_rec, (unsigned long long)n_bases, (unsigned long long)h);
    return 0;
}
