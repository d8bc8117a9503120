// capi.cpp -- the C ABI of include/lphq.h over the loader, the flat image and the gfx950 kernels.
// No CPU query path exists here: every query entry point launches HIP kernels or fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/lphq.h"
#include "bbhash.hpp"
#include "host_cpus.hpp"
#include "image.hpp"
#include "kernels.hpp"
#include "lph_format.hpp"
#include "runs_host.hpp"
#include "work_pool.hpp"

using namespace lphq;

// One replica of the flat image per attached device.  Replicas are added by lphq_attach / lphq_attach_devices before the
// index is shared between threads and never change afterwards, so the query entry points read them without a lock.
struct Replica {
    int device = -1;
    uint8_t* d_image = nullptr;
    bool owns = false;
    int how = 0;  // lphq_replica_info: 0 uploaded, 1 peer copy with peer access, 2 peer copy staged by the runtime, 3 adopted
};
// Per-call scratch of the host-buffer path and of lphq_count_valid: buffers plus a private non-blocking stream, so that
// concurrent callers neither serialise on a lock nor meet on the null stream.  Slots are recycled through a free list.
struct Scratch {
    int device = -1;
    hipStream_t stream = nullptr;
    char* d_seq = nullptr;
    size_t d_seq_cap = 0;
    uint64_t* d_out = nullptr;
    size_t d_out_cap = 0;
    uint64_t* d_offs = nullptr;  // 4 record offsets + 1 counter
    uint64_t* h_pin = nullptr;   // pinned: the five words above on their way to the device, then the counter on its way back ([5])
};
struct lphq_index {
    lphq_image_header hdr;
    std::vector<uint8_t> image;  // host copy of the flat image (empty for lphq_open_device_image handles)
    uint64_t file_bytes = 0;
    uint64_t part_bits[4] = {0, 0, 0, 0}, ef_entries = 0;  // for lphq_get_info (print_statistics of the CLI)
    size_t image_bytes = 0;
    std::vector<Replica> reps;   // reps[0] = primary device
    std::mutex pool_mu;          // guards pool only (pop / push of a slot), never held across device work
    std::vector<Scratch*> pool;
    // ONE count of the users of this object: the caller's handle holds one reference (dropped by lphq_close), every open session one (dropped by
    // lphq_session_close).  Whoever takes it from 1 to 0 destroys the index -- exactly once, whatever the interleaving of a closing main thread and a
    // worker closing the last session (two flags checked in two orders could both see "I am last": a double free).
    std::atomic<int> refs{1};
    std::atomic<bool> closing{false};   // lphq_close was called: the handle is dead for new sessions (the memory lives until the last reference goes)
};

static thread_local std::string g_err;
extern "C" const char* lphq_last_error(void) { return g_err.c_str(); }
namespace lphq {
void set_last_error(const std::string& s) { g_err = s; }
}  // namespace lphq
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
static int fail_hip(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? LPHQ_ERR_NO_DEVICE : LPHQ_ERR_HIP;
}
#define HIP_TRY(call)                                   \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) return fail_hip(e_, #call); \
    } while (0)

// For calls made AFTER work was queued on `stream` that reads or writes caller memory (or a slot that goes back to its pool on return): on
// failure nothing may still be in flight when the caller learns that the call did not happen.
#define HIP_TRY_DRAIN(call, stream)                        \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) {                            \
            (void)hipStreamSynchronize(stream);            \
            return fail_hip(e_, #call);                    \
        }                                                  \
    } while (0)

static bool header_is_consistent(const lphq_image_header& h, size_t nbytes, std::string& why);

struct DeviceGuard {  // make idx->device current for the duration of a call
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

// kind LPHQ_PTHASH: `k` is the caller's (a bare single_phf file does not hold it); otherwise k is read from the file
static int open_memory(const void* buf, size_t len, int kind, int kmer_bits, int k, lphq_index** out) {
    if (!buf || !out) return fail(LPHQ_ERR_ARG, "null argument");
    *out = nullptr;
    if (kind != LPHQ_PARTITIONED && kind != LPHQ_UNPARTITIONED && kind != LPHQ_PTHASH)
        return fail(LPHQ_ERR_ARG, "kind must be LPHQ_PARTITIONED or LPHQ_UNPARTITIONED");
    if (kmer_bits != 64 && kmer_bits != 128) return fail(LPHQ_ERR_ARG, "kmer_bits must be 64 or 128");
    LphFile file;
    std::string err;
    lphq_index* idx = nullptr;
    try {  // no exception may cross the C ABI (a hostile length field could otherwise surface as std::bad_alloc)
        const bool parsed = kind == LPHQ_PTHASH ? parse_pthash((const uint8_t*)buf, len, k, kmer_bits, file, err)
                                                : parse_lph((const uint8_t*)buf, len, kind, kmer_bits, file, err);
        if (!parsed) return fail(LPHQ_ERR_FORMAT, err);
        idx = new (std::nothrow) lphq_index();
        if (!idx) return fail(LPHQ_ERR_ARG, "out of memory");
        if (!build_image(file, idx->image, err)) {
            delete idx;
            return fail(err.find("beyond the device layout") != std::string::npos ? LPHQ_ERR_UNSUPPORTED : LPHQ_ERR_FORMAT, err);
        }
    } catch (const std::exception& e) {
        delete idx;
        return fail(LPHQ_ERR_FORMAT, std::string("malformed file: ") + e.what());
    }
    memcpy(&idx->hdr, idx->image.data(), sizeof idx->hdr);
    {
        std::string why;
        if (!header_is_consistent(idx->hdr, idx->image.size(), why)) {  // self-check of the image builder
            delete idx;
            return fail(LPHQ_ERR_FORMAT, "internal: image builder produced an inconsistent header: " + why);
        }
    }
    idx->file_bytes = file.bytes_read;
    for (int i = 0; i < 4; ++i) idx->part_bits[i] = file.part_bytes[i] * 8;
    idx->ef_entries = file.kind == LPHQ_PARTITIONED ? file.sizes_and_positions.size() : file.sizes.size() + file.positions.size();
    idx->image_bytes = idx->image.size();
    *out = idx;
    return LPHQ_OK;
}

extern "C" int lphq_open_memory(const void* buf, size_t len, int kind, int kmer_bits, lphq_index** out) {
    if (kind == LPHQ_PTHASH) return fail(LPHQ_ERR_ARG, "kind must be LPHQ_PARTITIONED or LPHQ_UNPARTITIONED (bare PTHash files: lphq_open_pthash)");
    return open_memory(buf, len, kind, kmer_bits, 0, out);
}

static int open_file(const char* path, int kind, int kmer_bits, int k, lphq_index** out) {
    if (!path || !out) return fail(LPHQ_ERR_ARG, "null argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(LPHQ_ERR_IO, std::string("Unable to open the input file ") + path);
    std::vector<uint8_t> buf;
    if (fseek(f, 0, SEEK_END) == 0) {
        long n = ftell(f);
        if (n > 0) buf.resize((size_t)n);
        fseek(f, 0, SEEK_SET);
    }
    size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    if (got != buf.size()) return fail(LPHQ_ERR_IO, std::string("short read on ") + path);
    return open_memory(buf.data(), buf.size(), kind, kmer_bits, k, out);
}

extern "C" int lphq_open(const char* path, int kind, int kmer_bits, lphq_index** out) {
    if (kind == LPHQ_PTHASH) return fail(LPHQ_ERR_ARG, "kind must be LPHQ_PARTITIONED or LPHQ_UNPARTITIONED (bare PTHash files: lphq_open_pthash)");
    return open_file(path, kind, kmer_bits, 0, out);
}

extern "C" int lphq_open_pthash(const char* path, int k, int kmer_bits, lphq_index** out) { return open_file(path, LPHQ_PTHASH, kmer_bits, k, out); }

extern "C" int lphq_open_bbhash(const char* path, int k, int kmer_bits, lphq_index** out) {
    if (!path || !out) return fail(LPHQ_ERR_ARG, "null argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(LPHQ_ERR_IO, std::string("Unable to open the input file ") + path);
    std::vector<uint8_t> buf;
    if (fseek(f, 0, SEEK_END) == 0) {
        long n = ftell(f);
        if (n > 0) buf.resize((size_t)n);
        fseek(f, 0, SEEK_SET);
    }
    size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    if (got != buf.size()) return fail(LPHQ_ERR_IO, std::string("short read on ") + path);
    lphq_index* idx = nullptr;
    try {
        BbhFile file;
        std::string err;
        if (!parse_bbhash(buf.data(), buf.size(), k, kmer_bits, file, err)) return fail(LPHQ_ERR_FORMAT, err);
        idx = new (std::nothrow) lphq_index();
        if (!idx) return fail(LPHQ_ERR_ARG, "out of memory");
        if (!build_bbhash_image(file, idx->image, err)) {
            delete idx;
            return fail(LPHQ_ERR_FORMAT, err);
        }
        idx->file_bytes = file.bytes_read;
    } catch (const std::exception& e) {
        delete idx;
        return fail(LPHQ_ERR_FORMAT, std::string("malformed file: ") + e.what());
    }
    memcpy(&idx->hdr, idx->image.data(), sizeof idx->hdr);
    idx->image_bytes = idx->image.size();
    *out = idx;
    return LPHQ_OK;
}

extern "C" int lphq_build_bbhash(const char* const* records, const size_t* lengths, size_t n_records, int k, int kmer_bits, double gamma, int threads,
                                 const char* out_path, lphq_build_stats* stats) {
    if ((n_records && (!records || !lengths)) || !out_path) return fail(LPHQ_ERR_ARG, "null argument");
    std::vector<uint8_t> bytes;
    uint64_t nkeys = 0;
    std::string err;
    try {
        if (!build_bbhash_file(records, lengths, n_records, k, kmer_bits, gamma, threads, bytes, nkeys, err)) return fail(LPHQ_ERR_ARG, err);
    } catch (const std::exception& e) {
        return fail(LPHQ_ERR_ARG, std::string("bbhash build: ") + e.what());
    }
    FILE* fo = fopen(out_path, "wb");
    if (!fo) return fail(LPHQ_ERR_IO, std::string("cannot write ") + out_path);
    const size_t put = fwrite(bytes.data(), 1, bytes.size(), fo);
    fclose(fo);
    if (put != bytes.size()) return fail(LPHQ_ERR_IO, std::string("short write on ") + out_path);
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->num_kmers = nkeys;
        stats->num_records = n_records;
        stats->file_bytes = bytes.size();
    }
    return LPHQ_OK;
}

static void scratch_destroy(Scratch* s) {
    DeviceGuard g(s->device);
    if (s->stream) {
        (void)hipStreamSynchronize(s->stream);
        (void)hipStreamDestroy(s->stream);
    }
    if (s->d_seq) (void)hipFree(s->d_seq);
    if (s->d_out) (void)hipFree(s->d_out);
    if (s->d_offs) (void)hipFree(s->d_offs);
    if (s->h_pin) (void)hipHostFree(s->h_pin);
    delete s;
}

static void destroy_index(lphq_index* idx) {
    for (Scratch* s : idx->pool) scratch_destroy(s);
    for (const Replica& r : idx->reps) {
        if (r.owns && r.d_image) {
            DeviceGuard g(r.device);
            (void)hipFree(r.d_image);
        }
    }
    delete idx;
}

// A session keeps using the replica's device memory until it is closed: closing the index first only marks it; the free happens when the last
// session goes (include/lphq.h).  The handle itself is dead for the caller from here on in either case.
static void index_unref(lphq_index* idx) {
    if (idx->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) destroy_index(idx);
}
extern "C" void lphq_close(lphq_index* idx) {
    if (!idx) return;
    idx->closing.store(true);
    index_unref(idx);  // the handle's own reference
}

// the replica queries of the calling thread run on: its current HIP device if the index is attached there, else the primary
// LPHQ_TEST_SELF_PEER (a test hook for boxes with one GPU): a device may be attached twice -- the second replica is made by the peer-copy path of
// lphq_attach_devices, from the first, on the same device -- and queries use the LAST replica of the current device, i.e. the copy.
static bool self_peer_hook() {
    static const bool on = getenv("LPHQ_TEST_SELF_PEER") != nullptr;
    return on;
}
// LPHQ_TEST_FAIL_PEER_COPY=N (test hook): the first N peer copies of lphq_attach_devices are reported as failed after their wait
static bool test_fail_peer_copy() {
    static std::atomic<int> left{getenv("LPHQ_TEST_FAIL_PEER_COPY") ? atoi(getenv("LPHQ_TEST_FAIL_PEER_COPY")) : 0};
    return left.load() > 0 && left.fetch_sub(1) > 0;
}
static const Replica* pick_replica(const lphq_index* idx) {
    if (idx->reps.empty()) return nullptr;
    if (idx->reps.size() > 1) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess) {
            if (self_peer_hook()) {
                for (size_t i = idx->reps.size(); i-- > 0;)
                    if (idx->reps[i].device == cur) return &idx->reps[i];
            }
            for (const Replica& r : idx->reps)
                if (r.device == cur) return &r;
        }
    }
    return &idx->reps[0];
}

// a scratch slot for `device` (recycled or new); returns nullptr with the error text set on failure
static Scratch* scratch_acquire(lphq_index* idx, int device) {
    {
        std::lock_guard<std::mutex> lk(idx->pool_mu);
        for (size_t i = 0; i < idx->pool.size(); ++i)
            if (idx->pool[i]->device == device) {
                Scratch* s = idx->pool[i];
                idx->pool[i] = idx->pool.back();
                idx->pool.pop_back();
                return s;
            }
    }
    Scratch* s = new (std::nothrow) Scratch();
    if (!s) {
        g_err = "out of memory";
        return nullptr;
    }
    s->device = device;
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_offs, 8 * sizeof(uint64_t));
    if (e == hipSuccess) e = hipHostMalloc((void**)&s->h_pin, 8 * sizeof(uint64_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)fail_hip(e, "scratch slot");
        scratch_destroy(s);
        return nullptr;
    }
    return s;
}
static void scratch_release(lphq_index* idx, Scratch* s) {
    std::lock_guard<std::mutex> lk(idx->pool_mu);
    idx->pool.push_back(s);
}
struct ScratchLease {  // returns the slot on every exit path
    lphq_index* idx;
    Scratch* s;
    ~ScratchLease() {
        if (s) scratch_release(idx, s);
    }
};

extern "C" int lphq_get_info(const lphq_index* idx, lphq_info* info) {
    if (!idx || !info) return fail(LPHQ_ERR_ARG, "null argument");
    memset(info, 0, sizeof *info);
    info->kind = idx->hdr.kind;
    info->kmer_bits = idx->hdr.kmer_bits;
    info->k = idx->hdr.k;
    info->m = idx->hdr.m;
    info->mm_seed = idx->hdr.mm_seed;
    info->num_kmers = idx->hdr.nkmers;
    info->distinct_minimizers = idx->hdr.distinct_minimizers;
    info->fallback_kmers = idx->hdr.fallback_kmers;
    info->file_bytes = idx->file_bytes;
    info->image_bytes = idx->image_bytes;
    info->device = idx->reps.empty() ? -1 : idx->reps[0].device;
    for (int i = 0; i < 4; ++i) info->kmers_by_type[i] = idx->hdr.kmers_by_type[i];
    info->minimizer_table_size = idx->hdr.mm.table_size;
    for (int i = 0; i < 4; ++i) info->part_bits[i] = idx->part_bits[i];
    info->ef_entries = idx->ef_entries;
    return LPHQ_OK;
}

extern "C" int lphq_image_bytes(const lphq_index* idx, size_t* nbytes) {
    if (!idx || !nbytes) return fail(LPHQ_ERR_ARG, "null argument");
    *nbytes = idx->image_bytes;
    return LPHQ_OK;
}

extern "C" int lphq_image_copy(const lphq_index* idx, void* host_dst, size_t cap) {
    if (!idx || !host_dst) return fail(LPHQ_ERR_ARG, "null argument");
    if (idx->image.empty()) return fail(LPHQ_ERR_STATE, "handle has no host copy of the image");
    if (cap < idx->image.size()) return fail(LPHQ_ERR_ARG, "destination too small");
    memcpy(host_dst, idx->image.data(), idx->image.size());
    return LPHQ_OK;
}

// Every section a kernel may read must lie inside the image: guards lphq_open_device_image against a foreign / stale buffer.
// the reciprocal fields of a divisor are the ones make_divisor (image.cpp) derives from d64: the kernels divide by dn / inv / shift alone, and an
// adopted image (lphq_open_device_image) whose fields disagreed would yield positions beyond the table, i.e. reads outside the image
static bool divisor_is_consistent(const lphq_divisor& v) {
    if (v.d64 == 0 || v.d64 > 0xFFFFFFFFull) return false;
    const uint32_t shift = (uint32_t)__builtin_clz((uint32_t)v.d64), dn = (uint32_t)v.d64 << shift;
    return v.d == (uint32_t)v.d64 && v.shift == shift && v.dn == dn && v.inv == (uint32_t)(~(uint64_t)0 / dn - (1ull << 32));
}

static bool header_is_consistent(const lphq_image_header& h, size_t nbytes, std::string& why) {
    auto inside = [&](uint64_t off, uint64_t bytes) { return off >= sizeof(lphq_image_header) && off <= nbytes && bytes <= nbytes - off; };
    auto phf_ok = [&](const lphq_phf_desc& f, const char* name) {
        if (f.num_keys == 0) return true;
        const uint64_t nb = f.num_dense + f.num_sparse;
        bool ok = f.table_size >= f.num_keys && f.table_size <= 0xFFFFFFFFull && nb <= 0xFFFFFFFFull && f.num_dense <= nb && f.front_size <= nb && f.front_width >= 1 && f.front_width <= 32 &&
                  f.back_width >= 1 && f.back_width <= 32 && f.div_table.d64 == f.table_size && f.div_dense.d64 == f.num_dense &&
                  f.div_sparse.d64 == f.num_sparse && f.num_dense > 0 && f.num_sparse > 0 && divisor_is_consistent(f.div_table) &&
                  divisor_is_consistent(f.div_dense) && divisor_is_consistent(f.div_sparse) &&
                  inside(f.off_front_ranks, (f.front_size * f.front_width + 7) / 8 + 8) &&
                  inside(f.off_back_ranks, ((nb - f.front_size) * f.back_width + 7) / 8 + 8) && inside(f.off_front_hp, (uint64_t)f.n_front_dict * 8) &&
                  inside(f.off_back_hp, (uint64_t)f.n_back_dict * 8) && inside(f.off_free, (f.table_size - f.num_keys) * 4);
        if (!ok) why = std::string(name) + " descriptor points outside the image";
        return ok;
    };
    if (h.kind > 3 || (h.kmer_bits != 64 && h.kmer_bits != 128) || h.k < 1 || h.k > h.kmer_bits / 2 - 1 || h.m < 1 || h.m > h.k ||
        (h.m > 32 && h.kind < LPHQ_PTHASH) /* no m-mers in the competitor functions: m = k */ || h.W != h.k - h.m + 1 || h.P != (h.W <= 15 ? 4u : 6u) || h.distinct_minimizers > 0xFFFFFFFFull || h.mm.num_keys != h.distinct_minimizers) {
        why = "inconsistent scalar fields";
        return false;
    }
    if (!phf_ok(h.mm, "minimizer_order") || !phf_ok(h.fb, "fallback_kmer_order")) return false;
    if (h.kind == LPHQ_PTHASH) return h.m == h.k;
    if (h.kind == LPHQ_BBHASH) {  // every level's arrays and the exact map lie inside the image
        const uint64_t nl = h.n_maximal;
        if (h.m != h.k || nl < 2 || nl > 1024 || !inside(h.off_uni, nl * sizeof(lphq_bbh_level))) {
            why = "bbhash level table lies outside the image";
            return false;
        }
        return true;  // (the level table itself is checked by bbh_levels_inside, on the host copy or after reading it back from the device)
    }
    if (h.mm.num_keys != 0 && !inside(h.off_tail, (h.mm.table_size - h.mm.num_keys) * 8)) {
        why = "tail table lies outside the image";
        return false;
    }
    if (h.kind == LPHQ_PARTITIONED) {
        const uint64_t E = LPHQ_UNI_PER_BLOCK(h.P);
        if (!inside(h.off_uni, ((h.distinct_minimizers + E - 1) / E + 1) * sizeof(lphq_uni_block))) {
            why = "fused blocks lie outside the image";
            return false;
        }
    } else if (!inside(h.off_upair, ((h.distinct_minimizers + LPHQ_PAIR_PER_BLOCK - 1) / LPHQ_PAIR_PER_BLOCK + 1) * sizeof(lphq_plane_block))) {
        why = "pair blocks lie outside the image";
        return false;
    }
    return true;
}

// kind 3: every level's bit array and rank samples, and the exact map, lie inside the image
static bool bbh_levels_inside(const lphq_image_header& h, const lphq_bbh_level* lv, size_t nbytes) {
    auto inside = [&](uint64_t off, uint64_t bytes) { return off >= sizeof(lphq_image_header) && off <= nbytes && bytes <= nbytes - off; };
    for (uint64_t i = 0; i < h.n_maximal; ++i)
        if (lv[i].hash_domain == 0 || lv[i].nwords != 1 + lv[i].hash_domain / 64 || !inside(lv[i].off_bits, lv[i].nwords * 8) ||
            !inside(lv[i].off_ranks, (lv[i].nwords + 7) / 8 * 8))
            return false;
    if (h.reserved0 < 1 || h.reserved0 > 40 || ((uint64_t)1 << h.reserved0) < 2 * h.fallback_kmers) return false;
    return inside(h.off_upair, ((uint64_t)1 << h.reserved0) * (h.kmer_bits / 64 + 1) * 8);
}

static int check_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(LPHQ_ERR_NO_DEVICE, std::string("no usable HIP device: ") + hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(LPHQ_ERR_ARG, "device ordinal out of range");
    return LPHQ_OK;
}

// Add a replica on `device`: from the host image when there is one and no replica exists yet, otherwise by a peer copy of
// the primary replica (xGMI between GPUs of one node; the runtime stages through the host when peer access is unavailable).
// With `stream` the peer copy is only queued there (lphq_attach_devices runs the copies to all devices concurrently and waits once).
static int add_replica(lphq_index* idx, int device, hipStream_t* queued_on = nullptr) {
    if (!(self_peer_hook() && queued_on && idx->reps.size() == 1))
        for (const Replica& r : idx->reps)
            if (r.device == device) return LPHQ_OK;
    if (idx->reps.empty() && idx->image.empty()) return fail(LPHQ_ERR_STATE, "no host image to upload");
    int rc = check_device(device);
    if (rc) return rc;
    DeviceGuard g(device);
    if (!g.ok) return fail(LPHQ_ERR_HIP, "hipSetDevice failed");
    uint8_t* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, idx->image_bytes));
    hipError_t e;
    int how = 0;
    hipStream_t st = nullptr;
    if (idx->reps.empty()) {
        e = hipMemcpy(d, idx->image.data(), idx->image_bytes, hipMemcpyHostToDevice);
    } else {
        // peer access (destination device current): enabled once per pair; "already enabled" is fine, "not possible" leaves the staged path
        const int src = idx->reps[0].device;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device, src) == hipSuccess && can) {
            const hipError_t pe = hipDeviceEnablePeerAccess(src, 0);
            if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) how = 1;
            (void)hipGetLastError();  // (clear the sticky "already enabled")
        }
        if (how == 0) how = 2;
        if (queued_on) {
            e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipMemcpyPeerAsync(d, device, idx->reps[0].d_image, src, idx->image_bytes, st);
        } else {
            e = hipMemcpyPeer(d, device, idx->reps[0].d_image, src, idx->image_bytes);
        }
    }
    if (e != hipSuccess) {
        if (st) (void)hipStreamDestroy(st);
        (void)hipFree(d);
        return fail_hip(e, idx->reps.empty() ? "hipMemcpy(image)" : "hipMemcpyPeer(image)");
    }
    if (queued_on) *queued_on = st;
    Replica r;
    r.device = device;
    r.d_image = d;
    r.owns = true;
    r.how = how;
    idx->reps.push_back(r);
    return LPHQ_OK;
}

extern "C" int lphq_attach(lphq_index* idx, int device) {
    if (!idx) return fail(LPHQ_ERR_ARG, "null argument");
    return add_replica(idx, device);
}

// Upload to the first device, then one peer copy per further device, all in flight at once (each on a stream of its destination device: on a node
// whose GPUs are fully connected over xGMI the N - 1 copies leave the primary over N - 1 different links), one wait per copy at the end.
extern "C" int lphq_attach_devices(lphq_index* idx, const int* devices, int n) {
    if (!idx || (n > 0 && !devices)) return fail(LPHQ_ERR_ARG, "null argument");
    if (n <= 0) return fail(LPHQ_ERR_ARG, "no device given");
    struct Pending {
        int device;
        hipStream_t stream;
        size_t rep;  // index into idx->reps of the replica the copy fills
    };
    std::vector<Pending> pending;
    int rc = LPHQ_OK;
    for (int i = 0; i < n && rc == LPHQ_OK; ++i) {
        hipStream_t st = nullptr;
        rc = add_replica(idx, devices[i], &st);
        if (st) pending.push_back({devices[i], st, idx->reps.size() - 1});
    }
    std::string first_error = rc == LPHQ_OK ? std::string() : g_err;
    std::vector<size_t> failed;
    for (Pending& p : pending) {  // (also after a failure: nothing may still be writing into a replica when the caller closes the index)
        DeviceGuard g(p.device);
        hipError_t e = hipStreamSynchronize(p.stream);
        (void)hipStreamDestroy(p.stream);
        if (e == hipSuccess && test_fail_peer_copy()) e = hipErrorUnknown;
        if (e != hipSuccess) {
            failed.push_back(p.rep);
            if (rc == LPHQ_OK) {
                rc = fail_hip(e, "hipMemcpyPeerAsync(image)");
                first_error = g_err;
            }
        }
    }
    // A replica whose copy did not complete must not stay attached: a retry would find its device "already attached" and queries there would read
    // a partial image.  (Replicas whose copies completed stay: the call is not all-or-nothing, lphq_attached_devices says what is there.)
    for (size_t j = failed.size(); j-- > 0;) {
        Replica& r = idx->reps[failed[j]];
        DeviceGuard g(r.device);
        (void)hipFree(r.d_image);
        idx->reps.erase(idx->reps.begin() + (std::ptrdiff_t)failed[j]);
    }
    if (rc != LPHQ_OK) g_err = first_error;
    return rc;
}

extern "C" int lphq_replica_info(const lphq_index* idx, int i, int* device, int* how) {
    if (!idx || i < 0 || (size_t)i >= idx->reps.size()) return fail(LPHQ_ERR_ARG, "no such replica");
    if (device) *device = idx->reps[(size_t)i].device;
    if (how) *how = idx->reps[(size_t)i].how;
    return LPHQ_OK;
}

extern "C" int lphq_attached_devices(const lphq_index* idx, int* devices, int cap, int* n) {
    if (!idx || !n) return fail(LPHQ_ERR_ARG, "null argument");
    *n = (int)idx->reps.size();
    for (int i = 0; i < *n && i < cap && devices; ++i) devices[i] = idx->reps[i].device;
    return LPHQ_OK;
}

extern "C" int lphq_open_device_image(const void* d_image, size_t nbytes, int device, lphq_index** out) {
    if (!d_image || !out) return fail(LPHQ_ERR_ARG, "null argument");
    *out = nullptr;
    if (nbytes < sizeof(lphq_image_header)) return fail(LPHQ_ERR_FORMAT, "image smaller than its header");
    int rc = check_device(device);
    if (rc) return rc;
    DeviceGuard g(device);
    if (!g.ok) return fail(LPHQ_ERR_HIP, "hipSetDevice failed");
    lphq_index* idx = new (std::nothrow) lphq_index();
    if (!idx) return fail(LPHQ_ERR_ARG, "out of memory");
    hipError_t e = hipMemcpy(&idx->hdr, d_image, sizeof idx->hdr, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        delete idx;
        return fail_hip(e, "hipMemcpy(header)");
    }
    if (idx->hdr.magic != LPHQ_IMAGE_MAGIC || idx->hdr.total_bytes != nbytes) {
        delete idx;
        return fail(LPHQ_ERR_FORMAT, "not a flat LPHQ image (magic/size mismatch)");
    }
    {
        std::string why;
        if (!header_is_consistent(idx->hdr, nbytes, why)) {
            delete idx;
            return fail(LPHQ_ERR_FORMAT, "flat LPHQ image rejected: " + why);
        }
    }
    if (idx->hdr.kind == LPHQ_BBHASH) {  // the level table lives in the image: read it back before trusting its offsets
        std::vector<lphq_bbh_level> lv(idx->hdr.n_maximal);
        e = hipMemcpy(lv.data(), (const uint8_t*)d_image + idx->hdr.off_uni, lv.size() * sizeof(lphq_bbh_level), hipMemcpyDeviceToHost);
        if (e != hipSuccess || !bbh_levels_inside(idx->hdr, lv.data(), nbytes)) {
            delete idx;
            return fail(LPHQ_ERR_FORMAT, "flat LPHQ image rejected: bbhash level table points outside the image");
        }
    }
    Replica r;
    r.device = device;
    r.d_image = (uint8_t*)d_image;
    r.owns = false;
    r.how = 3;
    idx->reps.push_back(r);
    idx->image_bytes = nbytes;
    *out = idx;
    return LPHQ_OK;
}

extern "C" int lphq_device_image(const lphq_index* idx, const void** d_image, size_t* nbytes) {
    if (!idx || !d_image || !nbytes) return fail(LPHQ_ERR_ARG, "null argument");
    const Replica* r = pick_replica(idx);
    if (!r) return fail(LPHQ_ERR_STATE, "index is not attached to a device");
    *d_image = r->d_image;
    *nbytes = idx->image_bytes;
    return LPHQ_OK;
}

#define REQUIRE_ATTACHED(idx)                                                                        \
    do {                                                                                             \
        if (!(idx)) return fail(LPHQ_ERR_ARG, "null index");                                         \
        if ((idx)->reps.empty()) return fail(LPHQ_ERR_STATE, "index is not attached to a device (no CPU fallback exists)"); \
    } while (0)

extern "C" int lphq_query_packed(const lphq_index* idx, const void* d_kmers, size_t n, uint64_t* d_out, void* stream) {
    REQUIRE_ATTACHED(idx);
    if (n && (!d_kmers || !d_out)) return fail(LPHQ_ERR_ARG, "null buffer");
    const Replica* r = pick_replica(idx);
    DeviceGuard g(r->device);
    HIP_TRY(launch_query_packed(idx->hdr, r->d_image, d_kmers, n, d_out, (hipStream_t)stream));
    return LPHQ_OK;
}

extern "C" int lphq_query_records(const lphq_index* idx, const char* d_seq, size_t total_bases, const uint64_t* d_seq_offsets, size_t n_records,
                                  uint64_t* d_out, const uint64_t* d_out_offsets, int streaming, void* stream) {
    REQUIRE_ATTACHED(idx);
    if (n_records && (!d_seq_offsets || !d_out_offsets)) return fail(LPHQ_ERR_ARG, "null buffer");
    if (n_records > 0xFFFFFFFFull) return fail(LPHQ_ERR_ARG, "more than 2^32-1 records in one batch");
    if (total_bases && !d_seq) return fail(LPHQ_ERR_ARG, "null buffer");
    const Replica* r = pick_replica(idx);
    DeviceGuard g(r->device);
    HIP_TRY(launch_query_records(idx->hdr, r->d_image, d_seq, total_bases, d_seq_offsets, d_out_offsets, n_records, d_out, streaming, (hipStream_t)stream));
    return LPHQ_OK;
}

// the capacity arguments of the run-encoded result: whole arenas
static int check_run_caps(size_t cap_runs, size_t cap_codes) {
    if (cap_runs == 0 || cap_runs % LPHQ_RUN_ARENAS || cap_codes % LPHQ_RUN_ARENAS) return fail(LPHQ_ERR_ARG, "cap_runs / cap_codes must be multiples of LPHQ_RUN_ARENAS (cap_runs > 0)");
    return LPHQ_OK;
}

extern "C" int lphq_query_records_runs(const lphq_index* idx, const char* d_seq, size_t total_bases, const uint64_t* d_seq_offsets, size_t n_records,
                                       const uint64_t* d_out_offsets, lphq_run* d_runs, size_t cap_runs, uint64_t* d_codes, size_t cap_codes,
                                       lphq_run_counts* d_counts, void* stream) {
    REQUIRE_ATTACHED(idx);
    if (idx->hdr.kind >= LPHQ_PTHASH) return fail(LPHQ_ERR_ARG, "run-encoded results exist for LP-MPHF handles only (a competitor's codes are not affine)");
    if (!d_runs || !d_counts || (cap_codes && !d_codes)) return fail(LPHQ_ERR_ARG, "null buffer");
    if (n_records && (!d_seq_offsets || !d_out_offsets)) return fail(LPHQ_ERR_ARG, "null buffer");
    if (n_records > 0xFFFFFFFFull) return fail(LPHQ_ERR_ARG, "more than 2^32-1 records in one batch");
    if (total_bases && !d_seq) return fail(LPHQ_ERR_ARG, "null buffer");
    if ((uint64_t)total_bases >= (1ull << 40)) return fail(LPHQ_ERR_ARG, "a run addresses its first window in 40 bits: batches of up to 2^40 - 1 bases");
    const int rc = check_run_caps(cap_runs, cap_codes);
    if (rc) return rc;
    const Replica* r = pick_replica(idx);
    DeviceGuard g(r->device);
    const S8RunsArgs sink{d_runs, d_codes, d_counts, (uint64_t)cap_runs / LPHQ_RUN_ARENAS, (uint64_t)cap_codes / LPHQ_RUN_ARENAS};
    HIP_TRY(launch_query_records_runs(idx->hdr, r->d_image, d_seq, total_bases, d_seq_offsets, d_out_offsets, n_records, sink, (hipStream_t)stream));
    return LPHQ_OK;
}

extern "C" int lphq_expand_runs(const lphq_index* idx, const lphq_run* d_runs, size_t cap_runs, const uint64_t* d_codes, size_t cap_codes,
                                const lphq_run_counts* d_counts, uint64_t* d_out, size_t n_out, uint64_t* d_status, void* stream) {
    REQUIRE_ATTACHED(idx);
    if (!d_runs || !d_counts || (cap_codes && !d_codes) || (n_out && !d_out)) return fail(LPHQ_ERR_ARG, "null buffer");
    const int rc = check_run_caps(cap_runs, cap_codes);
    if (rc) return rc;
    const Replica* r = pick_replica(idx);
    DeviceGuard g(r->device);
    HIP_TRY(launch_expand_runs(d_runs, (uint64_t)cap_runs / LPHQ_RUN_ARENAS, d_codes, (uint64_t)cap_codes / LPHQ_RUN_ARENAS, d_counts, d_out, n_out,
                               (unsigned long long*)d_status, (hipStream_t)stream));
    return LPHQ_OK;
}

extern "C" int lphq_query_records_gather(const lphq_index* idx, const char* d_seq, size_t total_bases, const uint64_t* d_seq_offsets, size_t n_records,
                                         const uint64_t* d_out_offsets, const void* d_table, uint32_t elem_bytes, uint64_t missing, void* d_payload_out,
                                         void* stream) {
    REQUIRE_ATTACHED(idx);
    if (idx->hdr.kind >= LPHQ_PTHASH) return fail(LPHQ_ERR_ARG, "the fused lookup -> payload pass exists for LP-MPHF handles only (competitors: lphq_query_records + lphq_gather_payload)");
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) return fail(LPHQ_ERR_ARG, "elem_bytes must be 1, 2, 4 or 8");
    if (n_records && (!d_seq_offsets || !d_out_offsets || !d_table || !d_payload_out)) return fail(LPHQ_ERR_ARG, "null buffer");
    if (n_records > 0xFFFFFFFFull) return fail(LPHQ_ERR_ARG, "more than 2^32-1 records in one batch");
    if (total_bases && !d_seq) return fail(LPHQ_ERR_ARG, "null buffer");
    const Replica* r = pick_replica(idx);
    DeviceGuard g(r->device);
    const uint32_t el = elem_bytes == 1 ? 0u : elem_bytes == 2 ? 1u : elem_bytes == 4 ? 2u : 3u;
    const uint64_t miss = el == 3u ? missing : (missing & ((1ull << (8u * elem_bytes)) - 1ull));
    const S8GatherArgs sink{(const uint8_t*)d_table, (uint8_t*)d_payload_out, idx->hdr.nkmers, miss, el};
    HIP_TRY(launch_query_records_gather(idx->hdr, r->d_image, d_seq, total_bases, d_seq_offsets, d_out_offsets, n_records, sink, (hipStream_t)stream));
    return LPHQ_OK;
}

extern "C" int lphq_pack_kmers(const lphq_index* idx, const char* d_seq, const uint64_t* d_pos, size_t n, void* d_kmers, void* stream) {
    REQUIRE_ATTACHED(idx);
    if (n && (!d_seq || !d_pos || !d_kmers)) return fail(LPHQ_ERR_ARG, "null buffer");
    const Replica* r = pick_replica(idx);
    DeviceGuard g(r->device);
    HIP_TRY(launch_pack_kmers(idx->hdr, d_seq, d_pos, n, d_kmers, (hipStream_t)stream));
    return LPHQ_OK;
}

// count on `st`; on a failure the stream is drained before the scratch slot (which holds d_count) goes back to the pool, so that work
// still queued on the caller's stream cannot write into a slot another thread has leased meanwhile
static int count_valid_on(lphq_index* idx, const uint64_t* d_codes, size_t n, hipStream_t st, uint64_t* count) {
    REQUIRE_ATTACHED(idx);
    if (!count || (n && !d_codes)) return fail(LPHQ_ERR_ARG, "null argument");
    *count = 0;
    if (n == 0) return LPHQ_OK;
    const Replica* r = pick_replica(idx);
    DeviceGuard g(r->device);
    ScratchLease sl{idx, scratch_acquire(idx, r->device)};
    if (!sl.s) return LPHQ_ERR_HIP;
    unsigned long long* d_count = (unsigned long long*)(sl.s->d_offs + 4);
    hipError_t e = hipMemsetAsync(d_count, 0, 8, st);
    if (e == hipSuccess) e = launch_count_valid(d_codes, n, d_count, st);
    if (e == hipSuccess) e = hipMemcpyAsync(count, d_count, 8, hipMemcpyDeviceToHost, st);
    const hipError_t es = hipStreamSynchronize(st);  // on every path: nothing of this call is in flight when the lease ends
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) return fail_hip(e, "lphq_count_valid");
    return LPHQ_OK;
}
// the entry point of ABI 1 (four arguments, unchanged): counts on the default stream, i.e. after everything queued on blocking streams
extern "C" int lphq_count_valid(const lphq_index* idx, const uint64_t* d_codes, size_t n, uint64_t* count) {
    return count_valid_on(const_cast<lphq_index*>(idx), d_codes, n, nullptr, count);
}
// everything on the caller's stream: ordered after whatever produced d_codes there
extern "C" int lphq_count_valid_on_stream(const lphq_index* idx, const uint64_t* d_codes, size_t n, void* stream, uint64_t* count) {
    return count_valid_on(const_cast<lphq_index*>(idx), d_codes, n, (hipStream_t)stream, count);
}
extern "C" int lphq_abi_version(void) { return LPHQ_ABI_VERSION; }

extern "C" int lphq_gather_payload(const lphq_index* idx, const uint64_t* d_codes, size_t n, const void* d_table, uint32_t elem_bytes, uint64_t missing,
                                   void* d_out, void* stream) {
    REQUIRE_ATTACHED(idx);
    if (n && (!d_codes || !d_table || !d_out)) return fail(LPHQ_ERR_ARG, "null buffer");
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) return fail(LPHQ_ERR_ARG, "elem_bytes must be 1, 2, 4 or 8");
    const Replica* r = pick_replica(idx);
    DeviceGuard g(r->device);
    HIP_TRY(launch_gather_payload(d_codes, n, d_table, idx->hdr.nkmers, elem_bytes, missing, d_out, (hipStream_t)stream));
    return LPHQ_OK;
}

extern "C" int lphq_count_valid_async(const lphq_index* idx, const uint64_t* d_codes, size_t n, uint64_t* d_count, void* stream) {
    REQUIRE_ATTACHED(idx);
    if (!d_count || (n && !d_codes)) return fail(LPHQ_ERR_ARG, "null argument");
    const Replica* r = pick_replica(idx);
    DeviceGuard g(r->device);
    HIP_TRY(hipMemsetAsync(d_count, 0, 8, (hipStream_t)stream));
    HIP_TRY(launch_count_valid(d_codes, n, (unsigned long long*)d_count, (hipStream_t)stream));
    return LPHQ_OK;
}

extern "C" int lphq_query_contig(const lphq_index* cidx, const char* contig, size_t length, int streaming, uint64_t* out, size_t cap, size_t* n_out) {
    lphq_index* idx = const_cast<lphq_index*>(cidx);
    REQUIRE_ATTACHED(idx);
    if (!n_out || (length && !contig)) return fail(LPHQ_ERR_ARG, "null argument");
    *n_out = 0;
    const uint32_t k = idx->hdr.k;
    if (idx->hdr.kind >= LPHQ_PTHASH) streaming = 1;  // ptbb_file_itr yields nothing for windows holding a non-ACGT base
    if (length < k) return LPHQ_OK;  // partitioned_mphf.hpp:79 (streaming); the random branch is undefined there
    const size_t nwin = length - k + 1;
    if (!out || cap < nwin) return fail(LPHQ_ERR_ARG, "output buffer too small");
    const Replica* r = pick_replica(idx);
    DeviceGuard g(r->device);
    ScratchLease sl{idx, scratch_acquire(idx, r->device)};
    Scratch* sc = sl.s;
    if (!sc) return LPHQ_ERR_HIP;
    if (sc->d_seq_cap < length + 64) {
        if (sc->d_seq) (void)hipFree(sc->d_seq);
        sc->d_seq = nullptr;
        sc->d_seq_cap = 0;
        size_t want = (length + 64) * 5 / 4;
        HIP_TRY(hipMalloc((void**)&sc->d_seq, want));
        sc->d_seq_cap = want;
    }
    if (sc->d_out_cap < nwin) {
        if (sc->d_out) (void)hipFree(sc->d_out);
        sc->d_out = nullptr;
        sc->d_out_cap = 0;
        size_t want = nwin * 5 / 4 + 16;
        HIP_TRY(hipMalloc((void**)&sc->d_out, want * 8));
        sc->d_out_cap = want;
    }
    // The slot's own stream: callers on other threads run concurrently (the copies from / to pageable memory are staged by the
    // runtime; pinned caller buffers make them truly asynchronous).  One wait per call: the offsets travel from the slot's pinned
    // words, and whether the record holds a non-ACGT base at all is counted on the device behind the lookup kernel.
    uint64_t* hp = sc->h_pin;
    hp[0] = 0, hp[1] = (uint64_t)length, hp[2] = 0, hp[3] = (uint64_t)nwin, hp[4] = 0, hp[5] = 0;
    HIP_TRY(hipMemcpyAsync(sc->d_seq, contig, length, hipMemcpyHostToDevice, sc->stream));
    HIP_TRY_DRAIN(hipMemcpyAsync(sc->d_offs, hp, 5 * sizeof(uint64_t), hipMemcpyHostToDevice, sc->stream), sc->stream);
    HIP_TRY_DRAIN(launch_query_records(idx->hdr, r->d_image, sc->d_seq, length, sc->d_offs, sc->d_offs + 2, 1, sc->d_out, streaming, sc->stream), sc->stream);
    if (streaming) {
        HIP_TRY_DRAIN(launch_count_bad_bases(sc->d_seq, length, (unsigned long long*)(sc->d_offs + 4), sc->stream), sc->stream);
        HIP_TRY_DRAIN(hipMemcpyAsync(hp + 5, sc->d_offs + 4, sizeof(uint64_t), hipMemcpyDeviceToHost, sc->stream), sc->stream);
    }
    HIP_TRY_DRAIN(hipMemcpyAsync(out, sc->d_out, nwin * 8, hipMemcpyDeviceToHost, sc->stream), sc->stream);
    HIP_TRY_DRAIN(hipStreamSynchronize(sc->stream), sc->stream);
    size_t n = nwin;
    if (streaming && hp[5] != 0) {
        // Drop windows holding a non-ACGT base (the reference emits nothing for them).  Validity is taken from the input, not
        // from the LPHQ_INVALID sentinel: a non-member k-mer's wrapped arithmetic may legitimately produce 2^64-1.
        auto valid_base = [](unsigned char ch) {
            switch (ch) {
                case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': case 'U': case 'u': return true;
                default: return false;
            }
        };
        size_t last_bad = (size_t)-1;  // position of the last invalid base seen
        n = 0;
        for (size_t i = 0; i < length; ++i) {
            if (!valid_base((unsigned char)contig[i])) last_bad = i;
            if (i + 1 >= k) {  // window ending at base i starts at i + 1 - k
                const size_t w = i + 1 - k;
                if (last_bad == (size_t)-1 || last_bad < w) out[n++] = out[w];
            }
        }
    }
    *n_out = n;
    return LPHQ_OK;
}

// ------------------------------------------------------------------------------------------
// sessions: batches of host records, `depth` of them in flight (include/lphq.h, ABI 3)
// ------------------------------------------------------------------------------------------
struct SessionLane {
    int device = -1;
    const uint8_t* d_image = nullptr;
    hipStream_t stream = nullptr;
};
struct SessionSlot {
    const SessionLane* lane = nullptr;
    char* d_seq = nullptr;
    size_t d_seq_cap = 0;
    uint64_t* d_out = nullptr;
    size_t d_out_cap = 0;
    uint64_t* d_offs = nullptr;  // n + 1 record offsets, n + 1 code offsets, 1 counter
    size_t d_offs_cap = 0;
    uint64_t* h_offs = nullptr;  // pinned: the same two offset arrays, then the counter's copy
    size_t h_offs_cap = 0;
    uint64_t* h_out = nullptr;  // pinned: the codes
    size_t h_out_cap = 0;
    hipEvent_t done = nullptr;
    size_t n_records = 0, n_out = 0, total_bases = 0;
    bool streaming = false, in_flight = false;
    std::vector<const char*> recs;  // the caller's records (valid until collect): needed again only when the batch holds a non-ACGT byte
    std::vector<size_t> lens;
    std::vector<uint64_t> compact_offs;
    // run-encoded transport (streaming batches of LP-MPHF handles): the kernel appends to arenas, k_compact_runs makes them one dense array, and that
    // array -- 16 B per run of ~(W + 1) / 2 windows instead of 8 B per window -- is what crosses PCIe; collect() expands it on the host
    lphq_run* d_runs = nullptr;        // LPHQ_RUN_ARENAS x per_arena
    size_t d_runs_cap = 0;
    uint64_t* d_xc = nullptr;          // explicit codes, LPHQ_RUN_ARENAS x xper_arena
    size_t d_xc_cap = 0;
    lphq_run_counts* d_counts = nullptr;  // + 4 totals behind the counters
    size_t d_counts_cap = 0;
    lphq_run* d_dense = nullptr;
    size_t d_dense_cap = 0;
    uint64_t* d_xdense = nullptr;
    size_t d_xdense_cap = 0;
    lphq_run* h_runs = nullptr;        // pinned
    size_t h_runs_cap = 0;
    uint64_t* h_xc = nullptr;          // pinned
    size_t h_xc_cap = 0;
    unsigned long long* h_totals = nullptr;  // pinned: runs, explicit codes, overflow flag, windows
    size_t h_totals_cap = 0;
    size_t per_arena = 0, xper_arena = 0, copied_runs = 0, copied_xc = 0;
    bool runs_mode = false;   // this batch travels as runs
    bool expanded = false;    // ... and h_out already holds its expansion
};
struct lphq_session {
    lphq_index* idx = nullptr;
    // The replicas the session works on (lphq_session_open: one; lphq_session_open_devices: several): batches go to them in turn -- batch b to replica
    // b % replicas, `depth` in flight on each -- and are collected in submission order, so a caller's submit / collect loop spreads over the GPUs of a
    // node without knowing it.  The replica's device and image by value: lphq_attach* may grow (and move) idx->reps afterwards.
    std::vector<SessionLane> lanes;      // one per replica: its device, its image and a stream
    int depth = 2;                       // batches in flight per replica
    uint64_t submitted = 0, collected = 0;
    std::vector<SessionSlot> slot;       // depth x replicas; slot i works on replica i % replicas
    size_t xc_hint = 0;                  // explicit codes of the last runs batch (how much of the code array the next batch copies back unasked)
    std::unique_ptr<WorkPool> pool;      // host threads that expand runs into codes (created by the first collect that needs them)
};

extern "C" void* lphq_host_alloc(size_t bytes) {
    void* p = nullptr;
    // (portable: a caller's record buffers may feed sessions on several devices)
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
extern "C" void lphq_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

// the replica a session lane on `device` uses: the n-th replica there (LPHQ_TEST_SELF_PEER boxes hold several on one device), nullptr if there is none
static const Replica* nth_replica_on(const lphq_index* idx, int device, int nth) {
    const Replica* last = nullptr;
    for (const Replica& r : idx->reps)
        if (r.device == device) {
            last = &r;
            if (nth-- == 0) return &r;
        }
    return last;
}

static int session_open(lphq_index* idx, int depth, const std::vector<const Replica*>& reps, lphq_session** out) {
    if (depth < 1 || depth > 4) return fail(LPHQ_ERR_ARG, "depth must be 1..4");
    // the reference first, then the check: a lphq_close that slips in between finds the count >= 2 and cannot free the index under this call
    idx->refs.fetch_add(1, std::memory_order_acq_rel);
    if (idx->closing.load()) {
        index_unref(idx);
        return fail(LPHQ_ERR_STATE, "the index was closed");
    }
    lphq_session* s = new (std::nothrow) lphq_session();
    if (!s) {
        index_unref(idx);
        return fail(LPHQ_ERR_ARG, "out of memory");
    }
    s->idx = idx;
    s->depth = depth;
    // one lane (replica + stream) per replica; slot i works on lane i % replicas.  (A stream per SLOT was measured again in round 6, with the run-encoded
    // transport -- 2.6 B per k-mer out against 1 B in, so that overlapping the two directions could pay: the binding's streaming pass went from 0.18 to 0.26 ns
    // per k-mer with codes and from 0.12 to 0.22 with runs, profiles/r06_binding_rates.txt.)
    s->lanes.resize(reps.size());
    s->slot.resize(reps.size() * (size_t)depth);
    hipError_t e = hipSuccess;
    for (size_t l = 0; l < s->lanes.size() && e == hipSuccess; ++l) {
        s->lanes[l].device = reps[l]->device;
        s->lanes[l].d_image = reps[l]->d_image;
        DeviceGuard g(s->lanes[l].device);
        e = hipStreamCreateWithFlags(&s->lanes[l].stream, hipStreamNonBlocking);
    }
    for (size_t i = 0; i < s->slot.size() && e == hipSuccess; ++i) {
        s->slot[i].lane = &s->lanes[i % s->lanes.size()];
        DeviceGuard g(s->slot[i].lane->device);
        e = hipEventCreateWithFlags(&s->slot[i].done, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        lphq_session_close(s);
        return fail_hip(e, "lphq_session_open");
    }
    *out = s;
    return LPHQ_OK;
}

extern "C" int lphq_session_open(const lphq_index* cidx, int depth, lphq_session** out) {
    lphq_index* idx = const_cast<lphq_index*>(cidx);
    REQUIRE_ATTACHED(idx);
    if (!out) return fail(LPHQ_ERR_ARG, "null argument");
    *out = nullptr;
    return session_open(idx, depth, {pick_replica(idx)}, out);
}

extern "C" int lphq_session_open_devices(const lphq_index* cidx, int depth, const int* devices, int n_devices, lphq_session** out) {
    lphq_index* idx = const_cast<lphq_index*>(cidx);
    REQUIRE_ATTACHED(idx);
    if (!out || (n_devices > 0 && !devices)) return fail(LPHQ_ERR_ARG, "null argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return fail(LPHQ_ERR_ARG, "1..64 devices");
    std::vector<const Replica*> reps;
    for (int i = 0; i < n_devices; ++i) {
        int nth = 0;
        for (int j = 0; j < i; ++j) nth += devices[j] == devices[i];
        const Replica* r = nth_replica_on(idx, devices[i], nth);
        if (!r) return fail(LPHQ_ERR_STATE, "the index has no replica on device " + std::to_string(devices[i]) + " (lphq_attach_devices first)");
        reps.push_back(r);
    }
    return session_open(idx, depth, reps, out);
}

extern "C" void lphq_session_close(lphq_session* s) {
    if (!s) return;
    for (SessionLane& ln : s->lanes) {
        DeviceGuard g(ln.device);
        if (ln.stream) (void)hipStreamSynchronize(ln.stream);
    }
    for (SessionSlot& sl : s->slot) {
        DeviceGuard g(sl.lane ? sl.lane->device : 0);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.d_seq) (void)hipFree(sl.d_seq);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.d_offs) (void)hipFree(sl.d_offs);
        if (sl.h_offs) (void)hipHostFree(sl.h_offs);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.d_runs) (void)hipFree(sl.d_runs);
        if (sl.d_xc) (void)hipFree(sl.d_xc);
        if (sl.d_counts) (void)hipFree(sl.d_counts);
        if (sl.d_dense) (void)hipFree(sl.d_dense);
        if (sl.d_xdense) (void)hipFree(sl.d_xdense);
        if (sl.h_runs) (void)hipHostFree(sl.h_runs);
        if (sl.h_xc) (void)hipHostFree(sl.h_xc);
        if (sl.h_totals) (void)hipHostFree(sl.h_totals);
    }
    for (SessionLane& ln : s->lanes) {
        DeviceGuard g(ln.device);
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
    }
    lphq_index* idx = s->idx;
    delete s;
    if (idx) index_unref(idx);  // (lphq_close came first and this was the last session: the index goes now)
}

template <typename T>
static hipError_t grow_device(T*& p, size_t& cap, size_t want) {
    if (want <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t n = want + want / 4 + 64;
    const hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
}
template <typename T>
static hipError_t grow_pinned(T*& p, size_t& cap, size_t want) {
    if (want <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t n = want + want / 4 + 64;
    const hipError_t e = hipHostMalloc((void**)&p, n * sizeof(T), hipHostMallocDefault);
    if (e == hipSuccess) cap = n;
    return e;
}

static hipError_t grow_runs(SessionSlot& sl, const RunPlan& p) {
    hipError_t e = grow_device(sl.d_runs, sl.d_runs_cap, p.per_arena * LPHQ_RUN_ARENAS);
    if (e == hipSuccess) e = grow_device(sl.d_xc, sl.d_xc_cap, p.xper_arena * LPHQ_RUN_ARENAS);
    if (e == hipSuccess) e = grow_device(sl.d_counts, sl.d_counts_cap, (size_t)LPHQ_RUN_ARENAS + 1);  // the last entry holds the four totals
    if (e == hipSuccess) e = grow_device(sl.d_dense, sl.d_dense_cap, p.per_arena * LPHQ_RUN_ARENAS);
    if (e == hipSuccess) e = grow_device(sl.d_xdense, sl.d_xdense_cap, p.xper_arena * LPHQ_RUN_ARENAS);
    if (e == hipSuccess) e = grow_pinned(sl.h_runs, sl.h_runs_cap, p.per_arena * LPHQ_RUN_ARENAS);
    if (e == hipSuccess) e = grow_pinned(sl.h_xc, sl.h_xc_cap, p.copy_xc);
    if (e == hipSuccess) e = grow_pinned(sl.h_totals, sl.h_totals_cap, 4);
    return e;
}
// Which transport lphq_session_submit(streaming = 1) + lphq_session_collect use for LP-MPHF handles.  Codes (8 B per k-mer over PCIe) is the default: a
// caller that wants every code spelled out in host memory gets it fastest from the copy engine -- measured on the GPU box (16 cores' worth of CPU time,
// profiles/r06_binding_rates.txt): codes 0.18 ns per k-mer, runs expanded by 16 host threads 0.20.  LPHQ_SESSION_TRANSPORT=runs moves the batch as runs and
// expands on the host (hosts with more cores than PCIe).  A caller that can USE runs asks for them (lphq_session_submit_runs / _collect_runs: 0.10).
static bool default_transport_is_runs() {
    static const bool runs = [] {
        const char* e = getenv("LPHQ_SESSION_TRANSPORT");
        return e && std::string(e) == "runs";
    }();
    return runs;
}

// grow, keeping the first `keep` entries (a batch's unasked copy is already there when the rest is fetched)
template <typename T>
static hipError_t grow_pinned_keep(T*& p, size_t& cap, size_t want, size_t keep) {
    if (want <= cap) return hipSuccess;
    T* q = nullptr;
    const size_t n = want + want / 4 + 64;
    const hipError_t e = hipHostMalloc((void**)&q, n * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) return e;
    if (p && keep) memcpy(q, p, keep * sizeof(T));
    if (p) (void)hipHostFree(p);
    p = q;
    cap = n;
    return hipSuccess;
}

extern "C" int lphq_session_reserve(lphq_session* s, size_t total_bases, size_t n_records) {
    if (!s) return fail(LPHQ_ERR_ARG, "null argument");
    if (s->submitted != s->collected) return fail(LPHQ_ERR_STATE, "batches in flight: collect them first");
    const bool runs = s->idx->hdr.kind < LPHQ_PTHASH;
    for (SessionSlot& sl : s->slot) {
        DeviceGuard g(sl.lane->device);
        HIP_TRY(grow_device(sl.d_seq, sl.d_seq_cap, total_bases + 64));
        HIP_TRY(grow_device(sl.d_out, sl.d_out_cap, total_bases + 16));  // (windows <= bases)
        HIP_TRY(grow_device(sl.d_offs, sl.d_offs_cap, 2 * (n_records + 1) + 1));
        HIP_TRY(grow_pinned(sl.h_offs, sl.h_offs_cap, 2 * (n_records + 1) + 1));
        HIP_TRY(grow_pinned(sl.h_out, sl.h_out_cap, total_bases + 16));
        if (runs) HIP_TRY(grow_runs(sl, plan_runs(s->idx->hdr, total_bases, total_bases, n_records, 0, false)));
    }
    if (runs && !s->pool) s->pool.reset(new (std::nothrow) WorkPool(std::min(usable_cpus(), 16)));  // (thread creation before the caller's timer, like the buffers)
    return LPHQ_OK;
}

// the runs of a batch into its slot: kernel -> arenas -> one dense array -> the first p.copy_* entries to pinned memory (the rest, if any, when the batch is collected)
static int queue_runs(lphq_session* s, SessionSlot& sl, const RunPlan& p) {
    const lphq_index* idx = s->idx;
    const SessionLane& ln = *sl.lane;
    const size_t n = sl.n_records;
    const S8RunsArgs sink{sl.d_runs, sl.d_xc, sl.d_counts, (uint64_t)p.per_arena, (uint64_t)p.xper_arena};
    unsigned long long* d_totals = (unsigned long long*)(sl.d_counts + LPHQ_RUN_ARENAS);
    HIP_TRY_DRAIN(launch_query_records_runs(idx->hdr, ln.d_image, sl.d_seq, sl.total_bases, sl.d_offs, sl.d_offs + n + 1, n, sink, ln.stream), ln.stream);
    HIP_TRY_DRAIN(launch_compact_runs(sl.d_runs, p.per_arena, sl.d_xc, p.xper_arena, sl.d_counts, sl.d_dense, p.per_arena * LPHQ_RUN_ARENAS, sl.d_xdense,
                                      p.xper_arena * LPHQ_RUN_ARENAS, d_totals, ln.stream), ln.stream);
    HIP_TRY_DRAIN(hipMemcpyAsync(sl.h_totals, d_totals, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ln.stream), ln.stream);
    sl.copied_runs = std::min(p.copy_runs, sl.h_runs_cap);
    sl.copied_xc = std::min(p.copy_xc, sl.h_xc_cap);
    HIP_TRY_DRAIN(hipMemcpyAsync(sl.h_runs, sl.d_dense, sl.copied_runs * sizeof(lphq_run), hipMemcpyDeviceToHost, ln.stream), ln.stream);
    HIP_TRY_DRAIN(hipMemcpyAsync(sl.h_xc, sl.d_xdense, sl.copied_xc * sizeof(uint64_t), hipMemcpyDeviceToHost, ln.stream), ln.stream);
    sl.per_arena = p.per_arena;
    sl.xper_arena = p.xper_arena;
    return LPHQ_OK;
}

static int session_submit(lphq_session* s, const char* const* records, const size_t* lengths, size_t n, int streaming, bool want_runs) {
    if (!s || (n && (!records || !lengths))) return fail(LPHQ_ERR_ARG, "null argument");
    if (n > 0xFFFFFFFFull) return fail(LPHQ_ERR_ARG, "more than 2^32-1 records in one batch");
    if (s->submitted - s->collected >= (uint64_t)s->slot.size()) return fail(LPHQ_ERR_STATE, "all batches of the session are in flight: collect one first");
    lphq_index* idx = s->idx;
    const uint32_t k = idx->hdr.k;
    if (idx->hdr.kind >= LPHQ_PTHASH) streaming = 1;  // (as lphq_query_contig: the competitor functions yield nothing for windows holding a non-ACGT base)
    SessionSlot& sl = s->slot[s->submitted % (uint64_t)s->slot.size()];
    const SessionLane& ln = *sl.lane;
    DeviceGuard g(ln.device);
    uint64_t so = 0, oo = 0;
    for (size_t i = 0; i < n; ++i) {
        if (lengths[i] && !records[i]) return fail(LPHQ_ERR_ARG, "null record");
        so += lengths[i];
        oo += lengths[i] >= k ? lengths[i] - k + 1 : 0;
    }
    const bool runs = want_runs && streaming && idx->hdr.kind < LPHQ_PTHASH && so < (1ull << 40);
    RunPlan plan{};
    HIP_TRY(grow_device(sl.d_seq, sl.d_seq_cap, (size_t)so + 64));
    HIP_TRY(grow_device(sl.d_offs, sl.d_offs_cap, 2 * (n + 1) + 1));
    HIP_TRY(grow_pinned(sl.h_offs, sl.h_offs_cap, 2 * (n + 1) + 1));
    if (runs) {
        plan = plan_runs(idx->hdr, oo, so, n, s->xc_hint, false);
        HIP_TRY(grow_runs(sl, plan));
    } else {
        HIP_TRY(grow_device(sl.d_out, sl.d_out_cap, (size_t)oo + 16));
        HIP_TRY(grow_pinned(sl.h_out, sl.h_out_cap, (size_t)oo + 16));
    }
    uint64_t a = 0, o = 0;
    for (size_t i = 0; i < n; ++i) {
        sl.h_offs[i] = a;
        sl.h_offs[n + 1 + i] = o;
        a += lengths[i];
        o += lengths[i] >= k ? lengths[i] - k + 1 : 0;
    }
    sl.h_offs[n] = a;
    sl.h_offs[2 * n + 1] = o;
    sl.h_offs[2 * n + 2] = 0;
    sl.recs.assign(records, records + n);
    sl.lens.assign(lengths, lengths + n);
    sl.n_records = n;
    sl.n_out = (size_t)oo;
    sl.total_bases = (size_t)so;
    sl.streaming = streaming != 0;
    sl.runs_mode = runs;
    sl.expanded = false;
    // one copy per run of records that follow each other in the caller's memory
    size_t at = 0;
    for (size_t i = 0; i < n;) {
        const char* run = records[i];
        size_t len = lengths[i], j = i + 1;
        while (j < n && (lengths[j] == 0 || records[j] == run + len)) len += lengths[j++];
        if (len) HIP_TRY_DRAIN(hipMemcpyAsync(sl.d_seq + at, run, len, hipMemcpyHostToDevice, ln.stream), ln.stream);
        at += len;
        i = j;
    }
    HIP_TRY_DRAIN(hipMemcpyAsync(sl.d_offs, sl.h_offs, (2 * (n + 1) + 1) * 8, hipMemcpyHostToDevice, ln.stream), ln.stream);
    if (runs) {
        if (sl.h_totals) sl.h_totals[0] = sl.h_totals[1] = sl.h_totals[2] = sl.h_totals[3] = 0;
        if (oo) {
            const int rc = queue_runs(s, sl, plan);
            if (rc) return rc;
            HIP_TRY_DRAIN(launch_count_bad_bases(sl.d_seq, (size_t)so, (unsigned long long*)(sl.d_offs + 2 * n + 2), ln.stream), ln.stream);
            HIP_TRY_DRAIN(hipMemcpyAsync(sl.h_offs + 2 * n + 2, sl.d_offs + 2 * n + 2, 8, hipMemcpyDeviceToHost, ln.stream), ln.stream);
        }
    } else if (oo) {
        HIP_TRY_DRAIN(launch_query_records(idx->hdr, ln.d_image, sl.d_seq, so, sl.d_offs, sl.d_offs + n + 1, n, sl.d_out, streaming, ln.stream), ln.stream);
        if (streaming) {
            HIP_TRY_DRAIN(launch_count_bad_bases(sl.d_seq, (size_t)so, (unsigned long long*)(sl.d_offs + 2 * n + 2), ln.stream), ln.stream);
            HIP_TRY_DRAIN(hipMemcpyAsync(sl.h_offs + 2 * n + 2, sl.d_offs + 2 * n + 2, 8, hipMemcpyDeviceToHost, ln.stream), ln.stream);
        }
        HIP_TRY_DRAIN(hipMemcpyAsync(sl.h_out, sl.d_out, (size_t)oo * 8, hipMemcpyDeviceToHost, ln.stream), ln.stream);
    }
    HIP_TRY_DRAIN(hipEventRecord(sl.done, ln.stream), ln.stream);
    sl.in_flight = true;
    ++s->submitted;
    return LPHQ_OK;
}

extern "C" int lphq_session_submit(lphq_session* s, const char* const* records, const size_t* lengths, size_t n, int streaming) {
    return session_submit(s, records, lengths, n, streaming, default_transport_is_runs());
}
extern "C" int lphq_session_submit_runs(lphq_session* s, const char* const* records, const size_t* lengths, size_t n) {
    if (s && s->idx->hdr.kind >= LPHQ_PTHASH) return fail(LPHQ_ERR_ARG, "run-encoded results exist for LP-MPHF handles only (a competitor's codes are not affine)");
    return session_submit(s, records, lengths, n, 1, true);
}

// Wait for the oldest batch; a runs batch is completed here: whatever the unasked copies left behind is fetched, and a batch whose arenas overflowed
// (the estimate is for random sequence; a homopolymer run has one super-k-mer per window) is answered again with worst-case arenas.  Synchronous, rare.
static int session_wait(lphq_session* s, SessionSlot& sl) {
    const SessionLane& ln = *sl.lane;
    HIP_TRY(hipEventSynchronize(sl.done));
    sl.in_flight = false;
    ++s->collected;
    if (!sl.runs_mode || sl.n_out == 0) return LPHQ_OK;
    if (sl.h_totals[2]) {
        const RunPlan p = plan_runs(s->idx->hdr, sl.n_out, sl.total_bases, sl.n_records, 0, true);
        HIP_TRY(grow_runs(sl, p));
        const int rc = queue_runs(s, sl, p);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ln.stream));
        if (sl.h_totals[2]) return fail(LPHQ_ERR_HIP, "internal: the run-encoded result overflowed worst-case arenas");
    }
    const size_t nr = (size_t)sl.h_totals[0], nx = (size_t)sl.h_totals[1];
    if (nr > sl.copied_runs) {
        HIP_TRY(grow_pinned_keep(sl.h_runs, sl.h_runs_cap, nr, sl.copied_runs));
        HIP_TRY(hipMemcpyAsync(sl.h_runs + sl.copied_runs, sl.d_dense + sl.copied_runs, (nr - sl.copied_runs) * sizeof(lphq_run), hipMemcpyDeviceToHost, ln.stream));
    }
    if (nx > sl.copied_xc) {
        HIP_TRY(grow_pinned_keep(sl.h_xc, sl.h_xc_cap, nx, sl.copied_xc));
        HIP_TRY(hipMemcpyAsync(sl.h_xc + sl.copied_xc, sl.d_xdense + sl.copied_xc, (nx - sl.copied_xc) * sizeof(uint64_t), hipMemcpyDeviceToHost, ln.stream));
    }
    if (nr > sl.copied_runs || nx > sl.copied_xc) HIP_TRY(hipStreamSynchronize(ln.stream));
    s->xc_hint = nx;
    return LPHQ_OK;
}

extern "C" int lphq_session_collect_runs(lphq_session* s, const lphq_run** runs, size_t* n_runs, const uint64_t** explicit_codes, size_t* n_codes,
                                         uint64_t* n_windows, const uint64_t** offsets, size_t* n_records) {
    if (!s || !runs || !n_runs || !explicit_codes || !n_codes || !offsets || !n_records) return fail(LPHQ_ERR_ARG, "null argument");
    if (s->submitted == s->collected) return fail(LPHQ_ERR_STATE, "no batch in flight");
    SessionSlot& sl = s->slot[s->collected % (uint64_t)s->slot.size()];
    if (!sl.runs_mode) return fail(LPHQ_ERR_STATE, "the oldest batch was not submitted with lphq_session_submit_runs: lphq_session_collect hands out its codes");
    DeviceGuard g(sl.lane->device);
    const int rc = session_wait(s, sl);
    if (rc) return rc;
    *runs = sl.h_runs;
    *n_runs = sl.n_out ? (size_t)sl.h_totals[0] : 0;
    *explicit_codes = sl.h_xc;
    *n_codes = sl.n_out ? (size_t)sl.h_totals[1] : 0;
    if (n_windows) *n_windows = sl.n_out ? (uint64_t)sl.h_totals[3] : 0;
    *offsets = sl.h_offs + sl.n_records + 1;
    *n_records = sl.n_records;
    return LPHQ_OK;
}

extern "C" int lphq_session_collect(lphq_session* s, const uint64_t** codes, const uint64_t** offsets, size_t* n_records) {
    if (!s || !codes || !offsets || !n_records) return fail(LPHQ_ERR_ARG, "null argument");
    if (s->submitted == s->collected) return fail(LPHQ_ERR_STATE, "no batch in flight");
    SessionSlot& sl = s->slot[s->collected % (uint64_t)s->slot.size()];
    DeviceGuard g(sl.lane->device);
    {
        const int rc = session_wait(s, sl);
        if (rc) return rc;
    }
    if (sl.runs_mode && sl.n_out && !sl.expanded) {
        // the batch crossed PCIe as runs; the caller wants the reference's vector: expanded here, by the pool's threads, into the slot's code buffer
        // (window i of the batch at h_out[i]; windows holding a non-ACGT base are in no run and are dropped below, from the input bases as ever)
        HIP_TRY(grow_pinned(sl.h_out, sl.h_out_cap, sl.n_out + 16));
        if (!s->pool) s->pool.reset(new (std::nothrow) WorkPool(std::min(usable_cpus(), 16)));
        expand_runs_host(sl.h_runs, (size_t)sl.h_totals[0], sl.h_xc, sl.h_out, s->pool.get());
        sl.expanded = true;
    }
    const size_t n = sl.n_records;
    const uint32_t k = s->idx->hdr.k;
    *codes = sl.h_out;
    *offsets = sl.h_offs + n + 1;
    *n_records = n;
    if (sl.streaming && sl.n_out && sl.h_offs[2 * n + 2] != 0) {
        // some record holds a byte that is no base: its windows over that byte were not emitted by the reference (partitioned_mphf.hpp:179-183).
        // Validity comes from the input bases (as in lphq_query_contig), the codes move down in place.
        auto valid_base = [](unsigned char ch) {
            switch (ch) {
                case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': case 'U': case 'u': return true;
                default: return false;
            }
        };
        sl.compact_offs.assign(n + 1, 0);
        size_t w = 0;
        for (size_t r = 0; r < n; ++r) {
            sl.compact_offs[r] = w;
            const size_t len = sl.lens[r];
            if (len < k) continue;
            const uint64_t* src = sl.h_out + sl.h_offs[n + 1 + r];
            const unsigned char* b = (const unsigned char*)sl.recs[r];
            size_t last_bad = (size_t)-1;
            for (size_t i = 0; i < len; ++i) {
                if (!valid_base(b[i])) last_bad = i;
                if (i + 1 >= k) {
                    const size_t win = i + 1 - k;
                    if (last_bad == (size_t)-1 || last_bad < win) sl.h_out[w++] = src[win];
                }
            }
        }
        sl.compact_offs[n] = w;
        *offsets = sl.compact_offs.data();
    }
    return LPHQ_OK;
}
