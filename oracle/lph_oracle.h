/*
 * lph_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the LPHash query path and of the build path that
 * produces the `.lph` files the query path reads.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load this; the product
 * (lphash_amd/, include/lphq.h) never links, imports or executes it.
 *
 * PARITY UNPINNED: the reference cannot be compiled in this image (its PTHash
 * dependency, github.com/jermp/pthash -- v1-era API, pin unknown -- is an empty
 * un-vendored submodule, /root/reference/.gitmodules:1-3) and the reference's
 * tests hold no golden vectors for the query path.
 * LPHash-owned logic is restated line by line from the cited files; PTHash /
 * essentials behaviour is restated from the published PTHash v1 algorithm.
 */
#ifndef LPH_ORACLE_H
#define LPH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_PARTITIONED = 0, ORC_UNPARTITIONED = 1 };

typedef struct orc_index orc_index;

/* essentials::load(hf, path)  (src/query.cpp:37) -> visit() order of include/partitioned_mphf.hpp:204-219
 * / include/unpartitioned_mphf.hpp:199-211.  kmer_bits selects the compile-time kmer_t
 * (include/compile_constants.tpd:1-2): 64 or 128. Returns NULL on error (see orc_last_error). */
orc_index* orc_load(const char* path, int kind, int kmer_bits);
orc_index* orc_load_mem(const uint8_t* buf, size_t len, int kind, int kmer_bits);
void orc_free(orc_index* idx);
const char* orc_last_error(void);

uint64_t orc_k(const orc_index* idx);
size_t orc_kmer_words(const orc_index* idx); /* 64-bit words per packed k-mer */
uint64_t orc_m(const orc_index* idx);
uint64_t orc_mm_seed(const orc_index* idx);
uint64_t orc_kmer_count(const orc_index* idx);   /* get_kmer_count()   */
uint64_t orc_minimizer_L0(const orc_index* idx); /* get_minimizer_L0() */
uint64_t orc_fallback_keys(const orc_index* idx);

/* MPHF::operator()(contig, length, streaming)  include/partitioned_mphf.hpp:73-197,
 * include/unpartitioned_mphf.hpp:72-192.  Writes at most cap codes, returns the number of
 * codes the reference's vector would hold (-1 on error). length < k returns 0 in both modes
 * (the reference's random branch is undefined there, partitioned_mphf.hpp:186). */
int64_t orc_query(const orc_index* idx, const char* contig, size_t length, int streaming,
                  uint64_t* out, size_t cap);

/* The random branch applied to already packed k-mers: words_per_kmer = 1 (kmer_bits 64) or 2
 * (little-endian low,high).  compute_minimizer_triplet + query(), include/mphf_utils.hpp:118-137,
 * src/partitioned_mphf.cpp:292-339, src/unpartitioned_mphf.cpp:191-206. */
void orc_query_packed(const orc_index* idx, const uint64_t* kmers, size_t n, uint64_t* out);

/* Build path restated (src/partitioned_mphf.cpp:33-268, src/unpartitioned_mphf.cpp:23-189,
 * include/minimizer.hpp:11-319, src/minimizer.cpp:5-50, include/ef_sequence.hpp:37-75,
 * include/rs_bit_vector.hpp:120-171, src/quartet_wtree.cpp:5-54) + essentials::save.
 * Contigs are in-memory records (what kseq_read yields).  Returns 0 on success. */
int orc_build(const char* const* contigs, const size_t* lengths, size_t n_contigs, int k, int m,
              uint64_t mm_seed, double c, int kind, int kmer_bits, const char* out_path);

/* The reference's query driver, src/query.cpp:24-88 (lph_oracle_driver.c): both passes over a FASTA/FASTQ(.gz) file with inflate
 * and record parsing inside the timers, one result vector per record.  Returns 0, or 2 if the file cannot be opened. */
int orc_query_file(const orc_index* idx, const char* query_path, uint64_t* total_kmers_streaming, uint64_t* total_kmers_random,
                   double* us_streaming, double* us_random);
/* records / bases / FNV-1a digest of a file as the driver's reader sees it (pinned against the real kseq.h by the tests) */
int orc_file_digest(const char* path, uint64_t* n_records, uint64_t* n_bases, uint64_t* digest);
/* orc_query_packed on `threads` POSIX threads over contiguous slices (the all-cores CPU figure) */
int orc_query_packed_threads(const orc_index* idx, const uint64_t* kmers, size_t n, uint64_t* out, int threads);

/* Competitor harness, PTHash half (src/ptbb/ptbb.hpp:16-34, ptbb_build.cpp:83-99, ptbb_query.cpp:40-55): a plain
 * pthash::single_phf<PTHasher, dictionary_dictionary, true> over the k-mers of the records, serialized by essentials::save. */
typedef struct orc_ptbb orc_ptbb;
int orc_ptbb_build(const char* const* contigs, const size_t* lengths, size_t n_contigs, int k, int kmer_bits, double c, const char* out_path);
orc_ptbb* orc_ptbb_load(const char* path, int k, int kmer_bits);
void orc_ptbb_free(orc_ptbb* h);
uint64_t orc_ptbb_num_keys(const orc_ptbb* h);
void orc_ptbb_query_packed(const orc_ptbb* h, const uint64_t* kmers, size_t n, uint64_t* out);
int64_t orc_ptbb_query_contig(const orc_ptbb* h, const char* contig, size_t length, uint64_t* out, size_t cap);

/* pieces exported for known-answer / model tests */
uint64_t orc_murmur64a(const void* key, size_t len, uint64_t seed);
uint64_t orc_minimizer_order(const orc_index* idx, uint64_t minimizer); /* PTHash over minimizers */

#ifdef __cplusplus
}
#endif
#endif
