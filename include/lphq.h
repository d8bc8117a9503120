/*
 * lphq.h -- C ABI of the MI355X-native LPHash query engine (liblphq.so).
 *
 * This is the drop-in boundary for LPHash's `query-p` / `query-u` hot path.  The reference has
 * no FFI layer; what a binding would wrap is (paths relative to the reference tree):
 *   - essentials::load(hf, path)                                   src/query.cpp:35-37
 *   - std::vector<uint64_t> MPHF::operator()(contig, length, streaming)
 *                                      include/partitioned_mphf.hpp:21-23,73-197 (class mphf,     query-p)
 *                                      include/unpartitioned_mphf.hpp:22-23,72-192 (class mphf_alt, query-u)
 *   - MPHF::get_kmer_count() / get_minimizer_L0()                  src/partitioned_mphf.cpp:270-272
 * and the on-disk format fixed by the visit() order (include/partitioned_mphf.hpp:204-219,
 * include/unpartitioned_mphf.hpp:199-211).  See INTEGRATION.md for the reference-side stub.
 *
 * All query entry points run on the GPU (hand-written gfx950 kernels).  There is NO CPU
 * fallback: every call that needs a device fails with LPHQ_ERR_NO_DEVICE / LPHQ_ERR_HIP when
 * none is usable.  Plain pointers and sizes only; no exceptions cross this boundary.
 */
#ifndef LPHQ_H
#define LPHQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this interface.  An entry point never changes its signature: new behaviour gets a new name and the number goes up
 * (2: lphq_count_valid_on_stream, lphq_abi_version; 3: lphq_session_*, lphq_host_alloc / lphq_host_free, lphq_replica_info; 4: the run-encoded
 * streaming result -- lphq_query_records_runs, lphq_expand_runs, lphq_session_submit_runs / lphq_session_collect_runs --, the fused lookup -> payload
 * pass lphq_query_records_gather, and sessions over several replicas, lphq_session_open_devices).  A caller compiled against version N works with
 * any library reporting >= N. */
#define LPHQ_ABI_VERSION 4
int lphq_abi_version(void);

/* ---- status codes ---- */
enum {
    LPHQ_OK = 0,
    LPHQ_ERR_ARG = 1,        /* bad argument (NULL, kind, kmer_bits, k/m out of range)       */
    LPHQ_ERR_IO = 2,         /* cannot open / read the file (CLI exit code 2, query.cpp:43-46) */
    LPHQ_ERR_FORMAT = 3,     /* truncated file or a redundant field disagrees                 */
    LPHQ_ERR_NO_DEVICE = 4,  /* no usable HIP device                                          */
    LPHQ_ERR_HIP = 5,        /* a HIP call failed; lphq_last_error() has the text             */
    LPHQ_ERR_STATE = 6,      /* index not attached to a device / wrong order of calls         */
    LPHQ_ERR_UNSUPPORTED = 7 /* index exceeds a limit of the device layout    */
};

/* which reference class wrote the file: build-p -> class mphf, build-u -> class mphf_alt
 * (src/lphash.cpp:25-32).  The file carries no magic, so the caller says. */
enum { LPHQ_PARTITIONED = 0, LPHQ_UNPARTITIONED = 1, LPHQ_PTHASH = 2, LPHQ_BBHASH = 3 /* competitor harness, see lphq_open_pthash / lphq_open_bbhash */ };

/* Sentinel written by the device streaming entry point for a window that contains a non-ACGT
 * base (the reference's streaming branch emits nothing for those windows,
 * include/partitioned_mphf.hpp:179-183).  The host entry point compacts them away. */
#define LPHQ_INVALID UINT64_MAX

typedef struct lphq_index lphq_index;

typedef struct lphq_info {
    uint32_t kind;       /* LPHQ_PARTITIONED / LPHQ_UNPARTITIONED                       */
    uint32_t kmer_bits;  /* 64 or 128: the reference's compile-time kmer_t width          */
    uint32_t k, m;
    uint64_t mm_seed;
    uint64_t num_kmers;            /* get_kmer_count()                                  */
    uint64_t distinct_minimizers;  /* get_minimizer_L0()                                */
    uint64_t fallback_kmers;       /* fallback_kmer_order.num_keys()                    */
    uint64_t file_bytes;           /* bytes consumed by the loader (essentials::load)   */
    uint64_t image_bytes;          /* size of the flat device image                     */
    int32_t device;                /* attached device, -1 if none                       */
    uint32_t reserved;
    /* census of the partitioned function (0 for unpartitioned): k-mers whose minimizer is LEFT, RIGHT, MAXIMAL,
     * NONE (include/quartet_wtree.hpp:7); with fallback_kmers they add up to num_kmers */
    uint64_t kmers_by_type[4];
    uint64_t minimizer_table_size; /* PTHash table_size of minimizer_order (>= distinct_minimizers) */
    /* serialized bits of the parts print_statistics() reports (src/partitioned_mphf.cpp:341-380, src/unpartitioned_mphf.cpp:208-240):
     * minimizer MPHF, wavelet tree (0 for query-u), Elias-Fano arrays, fallback MPHF; 0 for handles opened from a device image */
    uint64_t part_bits[4];
    uint64_t ef_entries;           /* sizes_and_positions.size() (query-p) / sizes.size() + positions.size() (query-u) */
} lphq_info;

/* Thread-local text of the last error raised on the calling thread. */
const char* lphq_last_error(void);

/* ---- load: replaces `MPHF hf; essentials::load(hf, path)` (src/query.cpp:35-37) ----
 * Parses the serialized mphf / mphf_alt, cross-checks its redundant fields and builds the flat
 * device image on the host.  kmer_bits is the reference's compile-time kmer_t width
 * (include/compile_constants.tpd:1-2): 64 or 128; it changes the fallback hash
 * (include/constants.hpp:56-70) and is not recorded in the file. */
int lphq_open(const char* path, int kind, int kmer_bits, lphq_index** out);
int lphq_open_memory(const void* buf, size_t len, int kind, int kmer_bits, lphq_index** out);
void lphq_close(lphq_index* idx);
int lphq_get_info(const lphq_index* idx, lphq_info* info);

/* ---- flat image (position independent; what gets RCCL-broadcast between GPUs) ---- */
int lphq_image_bytes(const lphq_index* idx, size_t* nbytes);
int lphq_image_copy(const lphq_index* idx, void* host_dst, size_t cap);
/* Create an index handle around an image that is ALREADY resident on `device` (e.g. received
 * by ncclBroadcast).  The library reads the header back and keeps the pointer; the caller keeps
 * ownership of the memory and must keep it alive until lphq_close. */
int lphq_open_device_image(const void* d_image, size_t nbytes, int device, lphq_index** out);

/* Put a replica of the image on `device`.  The first attach uploads the host image (hipMalloc + hipMemcpy) and makes that
 * device the PRIMARY one; attaching further devices copies the primary replica peer to peer (hipMemcpyPeer: xGMI between the
 * GPUs of one node).  Attaching a device twice is a no-op.  Attach before sharing the handle between threads. */
int lphq_attach(lphq_index* idx, int device);
/* lphq_attach for devices[0..n): the single-process form of "index replicated on every GPU, query batches sharded"
 * (one process per GPU uses lphq_open_device_image after an RCCL broadcast instead).  The reference object a
 * binding replaces is one immutable `MPHF hf` shared by all callers (include/partitioned_mphf.hpp:21-23).
 * On failure the replicas whose copy did not complete are NOT attached (a retry makes them again); those that completed stay --
 * lphq_attached_devices says what is there.  Attach rule: lphq_attach / lphq_attach_devices may run while sessions are open (a session holds its
 * replica's device and image by value) but not concurrently with the other query entry points of the same handle, which read the replica list. */
int lphq_attach_devices(lphq_index* idx, const int* devices, int n);
/* Attached devices in attach order (primary first): writes min(*n, cap) ordinals, *n = number attached. */
int lphq_attached_devices(const lphq_index* idx, int* devices, int cap, int* n);
/* Device pointer of the replica the calling thread would use (for broadcast from the root rank). */
int lphq_device_image(const lphq_index* idx, const void** d_image, size_t* nbytes);

/* ---- queries on device-resident buffers (stream = hipStream_t, NULL = default stream) ----
 * All are asynchronous with respect to the host.  Device selection: a call runs on the calling thread's CURRENT HIP device if
 * the index has a replica there, otherwise on the primary device; buffers and stream must belong to that device.  Any number
 * of host threads may call concurrently on one handle (the index is immutable, as `const` MPHF::operator() is). */

/* Random (non-streaming) lookups of n packed k-mers: the `streaming == false` branch
 * (include/partitioned_mphf.hpp:185-195) applied to kmer_t values as produced by
 * debug::string_to_integer_no_reverse (include/mphf_utils.hpp:110-116): first base in the most
 * significant used bits.  kmer_bits 64: one uint64 per k-mer; 128: two (low, high). */
int lphq_query_packed(const lphq_index* idx, const void* d_kmers, size_t n, uint64_t* d_out, void* stream);

/* All windows of a batch of records laid out back to back in d_seq (ASCII, no separators):
 * record r = d_seq[d_seq_offsets[r] .. d_seq_offsets[r+1]), d_seq_offsets[0] = 0,
 * d_seq_offsets[n_records] = total_bases (n_records + 1 entries) ; its codes go to
 * d_out[d_out_offsets[r] + i] for window i.  d_out_offsets[r+1]-d_out_offsets[r] must be
 * max(0, len_r - k + 1).  streaming != 0: one minimizer hash per base, super-k-mer leaders do the
 * lookup (the reference's streaming branch, include/partitioned_mphf.hpp:78-184); windows holding
 * a non-ACGT base get LPHQ_INVALID (a NON-member k-mer can also evaluate to 2^64-1 through the
 * reference's wrapping arithmetic; callers that query non-members and need to tell the two apart
 * derive validity from the bases, as lphq_query_contig does).  streaming == 0: every window does the full random lookup
 * with non-ACGT bases read as 'A' (include/mphf_utils.hpp:108). */
int lphq_query_records(const lphq_index* idx, const char* d_seq, size_t total_bases, const uint64_t* d_seq_offsets,
                       size_t n_records, uint64_t* d_out, const uint64_t* d_out_offsets, int streaming, void* stream);

/* Pack windows of d_seq into kmer_t values: out[i] = string_to_integer_no_reverse(d_seq + d_pos[i], k). */
int lphq_pack_kmers(const lphq_index* idx, const char* d_seq, const uint64_t* d_pos, size_t n, void* d_kmers, void* stream);

/* Number of codes != LPHQ_INVALID in d_codes[0..n) (what vector.size() of the reference's streaming pass, src/query.cpp:53,
 * would be).  Synchronous: counts on the default stream and waits for it. */
int lphq_count_valid(const lphq_index* idx, const uint64_t* d_codes, size_t n, uint64_t* count);
/* The same on `stream` (ordered after the query that produced d_codes there); waits for that stream only. */
int lphq_count_valid_on_stream(const lphq_index* idx, const uint64_t* d_codes, size_t n, void* stream, uint64_t* count);
/* The same count left in *d_count (8 bytes of device memory, overwritten) without waiting: for callers that keep several
 * batches in flight and read the count back with their own copy on `stream`. */
int lphq_count_valid_async(const lphq_index* idx, const uint64_t* d_codes, size_t n, uint64_t* d_count, void* stream);

/* ---- result consumer ----
 * The reference discards the codes (essentials::do_not_optimize_away, src/query.cpp:54); the use of an MPHF is to index per-k-mer data by
 * them.  d_out[i] = d_table[d_codes[i]] for elements of elem_bytes = 1, 2, 4 or 8 bytes (d_table holds num_kmers of them, in the order of
 * the hash codes), without leaving the device; codes that are not slots of the function -- LPHQ_INVALID, or what a NON-member k-mer's
 * wrapped arithmetic produced when it is >= num_kmers -- yield the low elem_bytes bytes of `missing`.  (An MPHF cannot reject non-members
 * whose code happens to be < num_kmers; that is the nature of the structure, not of this call.) */
int lphq_gather_payload(const lphq_index* idx, const uint64_t* d_codes, size_t n, const void* d_table, uint32_t elem_bytes, uint64_t missing,
                        void* d_out, void* stream);

/* ---- the streaming result as the reference COMPUTES it: runs (ABI 4) ----
 * The reference's streaming branch asks the function once per super-k-mer and then counts: the k-mers that follow under the same minimizer get the
 * previous code + 1 or - 1 (include/partitioned_mphf.hpp:131-145, case 0: `++mm_ctx.local_rank` for RIGHT / NONE minimizers, `--` for LEFT / MAXIMAL;
 * include/unpartitioned_mphf.hpp:130-137: `++hval`), and only k-mers under a colliding minimizer are looked up one by one in the fallback function
 * (:133-135).  A streaming pass therefore IS a list of affine runs -- (k - m + 2) / 2 windows each on random sequence: 6.5 at k = 31 / m = 20, 22.5
 * at k = 63 -- which src/query.cpp:52-54 expands into a vector and throws away.  lphq_query_records_runs hands out the runs themselves: 16 bytes per
 * run instead of 8 per k-mer leave the kernel (and, through a session, cross PCIe).
 *
 * One lphq_run = a run of windows of ONE record whose codes are affine (or explicit):
 *   code    kind UP / DOWN: the code of the run's first window; window first + i has code + i (UP) or code - i (DOWN), wrapping like the reference's
 *           uint64_t arithmetic.  kind EXPLICIT (colliding minimizer): index of the run's first code in the explicit-code array of the run's arena;
 *           window first + i has d_codes[arena * arena_codes + code + i]
 *   where   LPHQ_RUN_FIRST: index of the run's first window in the code space lphq_query_records writes (d_out_offsets[r] + window within record r);
 *           LPHQ_RUN_LENGTH: windows in the run (>= 1); LPHQ_RUN_KIND
 * Windows that hold a non-ACGT base belong to no run (the reference emits nothing for them).  Every other window of the batch belongs to exactly
 * one run.  Runs are cut where the kernel's tiles and waves end (every 512 window starts at most), so a long super-k-mer may arrive as two or three
 * adjacent runs of the same kind; nothing else splits it.
 *
 * Layout: LPHQ_RUN_ARENAS arenas.  A workgroup appends the runs of its tile to arena (tile % LPHQ_RUN_ARENAS) with ONE atomic add per tile; one
 * counter for the whole batch would serialise -- measured 83 M atomics/s on one address, 4.2 G/s over 64 (tools/gather_ceiling.hip,
 * profiles/gather_ceiling.json) -- where a 2 G-window batch has 2.1 M tiles in 6 ms.  Arena a holds d_counts[a].runs runs at
 * d_runs[a * arena_runs ...] (in no particular order: each run says where it goes) and d_counts[a].codes explicit codes at d_codes[a * arena_codes ...],
 * arena_runs = cap_runs / LPHQ_RUN_ARENAS, arena_codes = cap_codes / LPHQ_RUN_ARENAS (both capacities: multiples of LPHQ_RUN_ARENAS).  A counter may
 * END ABOVE its arena's capacity: the entries beyond were not written and the result is incomplete -- check after the stream has finished, or let
 * lphq_expand_runs report it.  Sizing: a batch of n windows yields about 2 n / (k - m + 2) + n / 500 runs on random sequence, at most n; a tile
 * (~1000 windows) puts all its runs into one arena, so the arenas of a SMALL batch fill unevenly -- an arena that holds
 * min(n, 1024 * ceil(tiles / LPHQ_RUN_ARENAS)) entries cannot overflow, whatever the sequence. */
typedef struct lphq_run {
    uint64_t code;
    uint64_t where;
} lphq_run;
enum { LPHQ_RUN_UP = 0, LPHQ_RUN_DOWN = 1, LPHQ_RUN_EXPLICIT = 2 };
#define LPHQ_RUN_FIRST(r) ((r).where & 0xFFFFFFFFFFULL)                     /* < 2^40: batches of up to 2^40 windows */
#define LPHQ_RUN_LENGTH(r) ((uint32_t)(((r).where >> 40) & 0x3FFFFFu))
#define LPHQ_RUN_KIND(r) ((uint32_t)((r).where >> 62))
#define LPHQ_RUN_ARENAS 32
typedef struct lphq_run_counts { /* one per arena, a cache line of its own (the arenas' atomics must not meet on a line) */
    uint64_t runs, codes, reserved[14];
} lphq_run_counts;
/* The streaming pass of lphq_query_records (same arguments, same tiles, same lookups) with the run-encoded result.  d_counts: LPHQ_RUN_ARENAS
 * entries, zeroed by the call on `stream`.  LPHQ_ERR_ARG for competitor handles (kind >= LPHQ_PTHASH: their codes are not affine) and for batches
 * of 2^40 windows or more. */
int lphq_query_records_runs(const lphq_index* idx, const char* d_seq, size_t total_bases, const uint64_t* d_seq_offsets, size_t n_records,
                            const uint64_t* d_out_offsets, lphq_run* d_runs, size_t cap_runs, uint64_t* d_codes, size_t cap_codes,
                            lphq_run_counts* d_counts, void* stream);
/* d_out[0..n_out) from the runs: what lphq_query_records(streaming = 1) writes for the same batch, LPHQ_INVALID included (d_out is filled with it
 * first).  Asynchronous; *d_status (8 bytes of device memory, may be NULL) is set to 0, or to 1 when an arena had overflowed or a run points
 * outside [0, n_out) -- such runs are skipped. */
int lphq_expand_runs(const lphq_index* idx, const lphq_run* d_runs, size_t cap_runs, const uint64_t* d_codes, size_t cap_codes,
                     const lphq_run_counts* d_counts, uint64_t* d_out, size_t n_out, uint64_t* d_status, void* stream);

/* ---- streaming lookup fused with its consumer (ABI 4) ----
 * d_payload_out[d_out_offsets[r] + i] = d_table[code of window i of record r], elements of elem_bytes = 1, 2, 4 or 8 bytes: lphq_query_records
 * (streaming) followed by lphq_gather_payload, in one kernel -- the codes never reach memory.  The k-mers of a super-k-mer have consecutive codes,
 * so their payload is contiguous in d_table: the reads this pass adds are what "locality-preserving" buys.  Windows that hold a non-ACGT base and
 * codes that are not slots of the function (>= num_kmers: a non-member k-mer's wrapped arithmetic) yield the low elem_bytes bytes of `missing`,
 * as in lphq_gather_payload.  LPHQ_ERR_ARG for competitor handles. */
int lphq_query_records_gather(const lphq_index* idx, const char* d_seq, size_t total_bases, const uint64_t* d_seq_offsets, size_t n_records,
                              const uint64_t* d_out_offsets, const void* d_table, uint32_t elem_bytes, uint64_t missing, void* d_payload_out,
                              void* stream);

/* ---- host-buffer convenience: mirrors MPHF::operator()(contig, length, streaming) ----
 * Copies the record to the device, runs the GPU path, copies the codes back and (streaming
 * only) drops the windows that hold a non-ACGT base -- decided from the input bases, not from the
 * LPHQ_INVALID sentinel -- so *n_out is what the reference's vector.size() would be for
 * ACGT input.  length < k yields 0 codes in both modes.  Each call works on its own stream and device buffers (recycled
 * between calls), so concurrent callers do not serialise on the library. */
int lphq_query_contig(const lphq_index* idx, const char* contig, size_t length, int streaming,
                      uint64_t* out, size_t cap, size_t* n_out);

/* ---- batched host-buffer queries: MPHF::operator() over MANY records per call, pipelined (ABI 3) ----
 * lphq_query_contig pays its copies and one stream synchronisation per record; a driver that loops over records like
 * src/query.cpp:48-59 gets 0.5-1 G k-mers/s from it.  A session keeps `depth` batches in flight on one stream of one device: submit copies the
 * records of a batch to the device (one asynchronous copy per run of records that are adjacent in memory -- a batch parsed into one arena is one
 * copy), runs the kernel and copies the codes into pinned host memory owned by the session, all without waiting; collect waits for the OLDEST
 * batch in flight and hands out its codes.  Typical loop (depth 2): submit(A); submit(B); collect(A) ... use A ...; submit(C); collect(B); ...
 *
 * A session belongs to one thread at a time; any number of sessions (and other calls) may use one index concurrently.  The device is the calling
 * thread's current HIP device at lphq_session_open if the index has a replica there, otherwise the primary one.
 * Lifetime: close sessions before the index.  If lphq_close comes first the handle is dead for every other call, but the index's device memory
 * is only released by the lphq_session_close of its last session -- a session never reads freed memory, whatever order a garbage collector picks.
 *
 * submit: the records' memory must stay valid and unchanged until the batch is collected (pinned memory -- lphq_host_alloc -- makes the copies
 *   truly asynchronous; pageable memory works, the runtime stages it).  LPHQ_ERR_STATE when `depth` batches are already in flight.
 *   When submit fails nothing of the batch is in flight any more (the stream is drained before the call returns): the records may be freed.
 * collect: *codes holds the batch's codes record after record, record r at [(*offsets)[r], (*offsets)[r + 1]) -- (*offsets)[r + 1] - (*offsets)[r]
 *   is what the reference's vector.size() would be: max(0, len_r - k + 1) windows, minus (streaming only) the windows that hold a non-ACGT base,
 *   decided from the input bases as in lphq_query_contig.  The pointers are valid until the next lphq_session_submit on this session.
 *   LPHQ_ERR_STATE when nothing is in flight. */
typedef struct lphq_session lphq_session;
int lphq_session_open(const lphq_index* idx, int depth /* 1..4 */, lphq_session** out);
/* ABI 4: one session over SEVERAL replicas of the index (lphq_attach_devices first): batches go to devices[0], devices[1], ... in turn -- batch b to
 * devices[b % n_devices], each device with a stream of its own and `depth` batches in flight -- and are collected in submission order, so the caller's
 * submit / collect loop spreads over the GPUs of a node unchanged (up to depth * n_devices submits before the first collect).  The reference object this
 * stands for is the one immutable `MPHF hf` all callers share (include/partitioned_mphf.hpp:21-23).  LPHQ_ERR_STATE if a device has no replica. */
int lphq_session_open_devices(const lphq_index* idx, int depth /* 1..4, per device */, const int* devices, int n_devices, lphq_session** out);
void lphq_session_close(lphq_session* s);
/* Allocate now what batches of up to total_bases bases in up to n_records records need (device buffers and the pinned result buffer of EVERY
 * slot): page-locked allocations are slow (tens of ms for hundreds of MB), a caller with a timer does them before it starts the timer, like the
 * reference's gzopen + kseq_init before t.start() (src/query.cpp:43-48).  Optional: submit grows the buffers when a batch needs more. */
int lphq_session_reserve(lphq_session* s, size_t total_bases, size_t n_records);
int lphq_session_submit(lphq_session* s, const char* const* records, const size_t* lengths, size_t n_records, int streaming);
int lphq_session_collect(lphq_session* s, const uint64_t** codes, const uint64_t** offsets /* n_records + 1 entries */, size_t* n_records);
/* The same pipeline with the streaming result left as the reference COMPUTES it (ABI 4; lphq_run above): submit_runs = submit(streaming = 1) whose
 * batch crosses PCIe as runs -- 16 B per super-k-mer instead of 8 B per k-mer --; collect_runs hands out ONE dense array of runs (no arenas on the
 * host; in no particular order; kind EXPLICIT: code indexes *explicit_codes), *n_windows = the windows all runs hold together (the reference's
 * vector.size() summed over the batch: windows holding a non-ACGT base are in no run), and *offsets[r] = where record r's windows begin in the
 * space LPHQ_RUN_FIRST counts in: max(0, len_r - k + 1) slots per record, non-ACGT windows included.  Pointers valid until the next submit.
 * lphq_session_collect works on such a batch too: it expands the runs on the host (all usable threads, up to 16) and returns codes and compacted
 * offsets exactly as for a batch submitted with lphq_session_submit.  (lphq_session_submit itself keeps the 8-byte codes on the wire: with 16 cores'
 * worth of host threads the copy engine delivers expanded codes faster than the host expands runs -- 0.18 against 0.20 ns per k-mer; runs pay off when
 * the caller consumes them AS runs: 0.10.  LPHQ_SESSION_TRANSPORT=runs in the environment makes lphq_session_submit(streaming) use runs + host
 * expansion: A/B runs, hosts with many cores.)  collect_runs on a batch that was not submitted as runs: LPHQ_ERR_STATE.  A batch of low-complexity sequence (up to one run per window) that outgrows the arenas sized for random sequence is
 * answered again, synchronously, with worst-case arenas: slower, never wrong. */
int lphq_session_submit_runs(lphq_session* s, const char* const* records, const size_t* lengths, size_t n_records);
int lphq_session_collect_runs(lphq_session* s, const lphq_run** runs, size_t* n_runs, const uint64_t** explicit_codes, size_t* n_codes,
                              uint64_t* n_windows /* may be NULL */, const uint64_t** offsets /* n_records + 1 entries */, size_t* n_records);
/* Page-locked host memory for record buffers handed to lphq_session_submit (hipHostMalloc); NULL when it cannot be had. */
void* lphq_host_alloc(size_t bytes);
void lphq_host_free(void* p);

/* How replica i (attach order) got its image: 0 = uploaded from the host image, 1 = peer copy with peer access enabled (xGMI / PCIe P2P),
 * 2 = peer copy staged by the runtime (no peer access between the two devices), 3 = adopted from device memory (lphq_open_device_image). */
int lphq_replica_info(const lphq_index* idx, int i, int* device, int* how);

/* ---- index construction (writes the reference's format) ----
 * In-memory build of a partitioned / unpartitioned LP-MPHF over the k-mers of the records
 * (src/partitioned_mphf.cpp:33-145, src/unpartitioned_mphf.cpp:23-136) and essentials::save
 * (src/build.cpp:28).  Host-side, multi-threaded. threads <= 0: all cores. */
typedef struct lphq_build_stats {
    uint64_t num_kmers;             /* nkmers                                               */
    uint64_t num_records;           /* records read (total_contigs before the reference's --) */
    uint64_t num_super_kmers;       /* all_minimizers.size()   (partitioned_mphf.cpp:82)     */
    uint64_t distinct_minimizers;   /* unique_mms.size()                                     */
    uint64_t colliding_super_kmers; /* coll_ids.size()         (partitioned_mphf.cpp:108)    */
    uint64_t fallback_kmers;        /* k-mers handed to the fallback MPHF                    */
    uint64_t file_bytes;            /* size of the serialized function                       */
} lphq_build_stats;
int lphq_build(const char* const* records, const size_t* lengths, size_t n_records, int k, int m,
               uint64_t mm_seed, double c, int kind, int kmer_bits, int threads, const char* out_path,
               lphq_build_stats* stats /* may be NULL */);

/* Same function, same bytes written, with the PTHash pilot search (95 % of the CPU build time) run on HIP device
 * `device` (speculative search of a window of buckets + in-order commit, lphash_amd/csrc/search.hip).  device < 0: CPU. */
int lphq_build_device(const char* const* records, const size_t* lengths, size_t n_records, int k, int m,
                      uint64_t mm_seed, double c, int kind, int kmer_bits, int threads, int device, const char* out_path,
                      lphq_build_stats* stats /* may be NULL */);

/* Construction from a stream of record batches with bounded memory: the reference's two passes over the input
 * (src/partitioned_mphf.cpp:58-81 kseq loop into external_memory_vector<mm_record_t>, :110-130 second pass collecting the k-mers of
 * repeated minimizers) with its `--max-memory` / `-d tmp_dirname` behaviour (src/parser_build.cpp:74-88, include/external_memory_vector.hpp:
 * sorted runs spilled to temporary files, merged by a heap).  `next` is called until it returns 0; every call that returns 1 hands over
 * one batch of WHOLE records (pointers valid until the following call).  The input is read twice: next(ctx, 0, ...) serves the first
 * pass, next(ctx, 1, ...) the second (the first call with pass = 1 restarts the input from its beginning).  < 0 from `next` aborts with
 * LPHQ_ERR_IO.  The sequence is never held whole: a batch at a time.  With max_memory_bytes != 0 everything whose size follows the input lives
 * in `tmp_dir` (created if absent; files lphash.tmp.<pid>.*, removed before returning), as in the reference (src/partitioned_mphf.cpp:58-130):
 * the super-k-mer records (32 B each, ~2 / (k - m + 2) per k-mer; sorted pieces of at most max_memory_bytes / 4, merged by a heap), the distinct
 * minimizers, their order under the minimizer MPHF (sorted pieces again), the super-k-mers of repeated minimizers, and the (bucket, hash) pairs of
 * both PTHash constructions (single_phf::build_in_external_memory: the merge of the sorted pieces files every bucket under its size, the pilot
 * search -- on `device` or on the CPU -- reads those files from the largest size down, a piece at a time, with the table of taken positions carried
 * over).  What stays in memory: one piece, the index under construction (its bit vectors and, until they are encoded, <= 3 B per distinct minimizer: type, size, position),
 * 8 B of pilot per PTHash bucket and the bitmaps of the two tables.  max_memory_bytes = 0: no limit, nothing touches the disk.  The bytes written
 * are those of lphq_build on the same records, whatever the batch boundaries and the memory limit. */
typedef int (*lphq_record_source)(void* ctx, int pass, const char* const** records, const size_t** lengths, size_t* n_records);
int lphq_build_stream(lphq_record_source next, void* ctx, int k, int m, uint64_t mm_seed, double c, int kind, int kmer_bits, int threads,
                      int device /* < 0: CPU pilot search */, const char* tmp_dir /* NULL: "." */, uint64_t max_memory_bytes,
                      const char* out_path, lphq_build_stats* stats /* may be NULL */);

/* ---- competitor harness: a plain PTHash MPHF over whole k-mers ----
 * The PTHash half of the reference's `ptbb_build` / `ptbb_query` (src/ptbb/ptbb_build.cpp:83-99, src/ptbb/ptbb_query.cpp:40-55,
 * src/ptbb/ptbb.hpp:16-34): pthash::single_phf<PTHasher, dictionary_dictionary, true> with
 * PTHasher::hash(kmer, seed) = {MurmurHash2_64(&kmer, sizeof kmer, seed), MurmurHash2_64(&kmer, sizeof kmer, ~seed)}, built over every window
 * of k valid bases (forward strand) with seed 1, alpha 0.94, and serialized alone by essentials::save.  The file holds neither k nor
 * the k-mer width, so the caller says.  The handle works with lphq_attach*, lphq_query_packed, lphq_query_records (any `streaming`
 * value: one code per window, LPHQ_INVALID for windows holding a non-ACGT base, which ptbb_file_itr skips), lphq_query_contig,
 * lphq_count_valid, lphq_get_info (kind = LPHQ_PTHASH, m = k, num_kmers = distinct_minimizers = number of keys). */
int lphq_open_pthash(const char* path, int k, int kmer_bits, lphq_index** out);
int lphq_build_pthash(const char* const* records, const size_t* lengths, size_t n_records, int k, int kmer_bits, double c, int threads,
                      int device /* < 0: CPU pilot search */, const char* out_path, lphq_build_stats* stats /* may be NULL */);

/* The BBHash half of the same harness (src/ptbb/ptbb_build.cpp:140-199, src/ptbb/ptbb_query.cpp:56-72): boomphf::mphf<kmer_t, ptbb::BBHasher>
 * (external/BooPHF.hpp) with BBHasher(kmer, seed) = MurmurHash2_64(&kmer, sizeof kmer, seed) ^ MurmurHash2_64(&kmer, sizeof kmer, ~seed), in the
 * file format of boomphf::mphf::save.  lphq_open_bbhash reads a file written by the reference (or by lphq_build_bbhash) and cross-checks its
 * redundant fields; lphq_build_bbhash writes one the reference's load() accepts (gamma >= 1; same 25-level cascade; the exact map of the last level
 * numbers its keys in input order, where the reference numbers them in arrival order of its threads).  The handle behaves like a PTHash one; a k-mer
 * that is not a key yields whatever bit it meets -- or LPHQ_INVALID (ULLONG_MAX in the reference) when it reaches the exact map and is not in it. */
int lphq_open_bbhash(const char* path, int k, int kmer_bits, lphq_index** out);
int lphq_build_bbhash(const char* const* records, const size_t* lengths, size_t n_records, int k, int kmer_bits, double gamma, int threads,
                      const char* out_path, lphq_build_stats* stats /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif /* LPHQ_H */
