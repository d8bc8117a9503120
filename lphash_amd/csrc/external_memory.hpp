// external_memory.hpp -- what construction with a memory limit stands on (the reference's include/external_memory_vector.hpp:43-109 and
// pthash's build_in_external_memory): records sorted in pieces, spilled to a temporary directory and merged by a heap; sequential record
// files; the small task pool and the partition-then-sort both builders share.  Host code only.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

namespace lphq {

// ---------------------------------------------------------------------------------------------
// tiny task pool
// ---------------------------------------------------------------------------------------------
inline void parallel_for(size_t n_tasks, int threads, const std::function<void(size_t)>& fn) {
    if (threads <= 1 || n_tasks <= 1) {
        for (size_t i = 0; i < n_tasks; ++i) fn(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    int nt = (int)std::min<size_t>((size_t)threads, n_tasks);
    for (int t = 0; t < nt; ++t)
        pool.emplace_back([&] {
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= n_tasks) break;
                fn(i);
            }
        });
    for (auto& th : pool) th.join();
}

template <typename Vec, typename KeyFn>
void parallel_sort_by_key(Vec& v, int threads, uint32_t key_bits, KeyFn key) {
    typedef typename Vec::value_type T;
    // partition by the top 10 key bits, then std::sort each partition
    const size_t n = v.size();
    if (n < (1u << 16) || threads <= 1) {
        std::sort(v.begin(), v.end(), [&](const T& a, const T& b) { return key(a) < key(b); });
        return;
    }
    const uint32_t PB = 10, NP = 1u << PB;
    const uint32_t shift = key_bits > PB ? key_bits - PB : 0;
    const size_t chunk = (n + threads - 1) / threads;
    const size_t nch = (n + chunk - 1) / chunk;
    std::vector<std::vector<uint64_t>> hist(nch, std::vector<uint64_t>(NP, 0));
    auto part = [&](const T& x) { return (uint32_t)std::min<uint64_t>(key(x) >> shift, NP - 1); };
    parallel_for(nch, threads, [&](size_t c) {
        for (size_t i = c * chunk, e = std::min(n, i + chunk); i < e; ++i) hist[c][part(v[i])]++;
    });
    std::vector<uint64_t> pstart(NP + 1, 0);
    for (uint32_t p = 0; p < NP; ++p) {
        uint64_t s = 0;
        for (size_t c = 0; c < nch; ++c) {
            uint64_t t = hist[c][p];
            hist[c][p] = pstart[p] + s;
            s += t;
        }
        pstart[p + 1] = pstart[p] + s;
    }
    Vec tmp(n);
    parallel_for(nch, threads, [&](size_t c) {
        for (size_t i = c * chunk, e = std::min(n, i + chunk); i < e; ++i) tmp[hist[c][part(v[i])]++] = v[i];
    });
    parallel_for(NP, threads, [&](size_t p) {
        std::sort(tmp.begin() + pstart[p], tmp.begin() + pstart[p + 1], [&](const T& a, const T& b) { return key(a) < key(b); });
    });
    v.swap(tmp);
}

// ---------------------------------------------------------------------------------------------
// records beyond the memory limit: sorted pieces on disk, merged by a heap
// (include/external_memory_vector.hpp:43-109: push_back -> sort_and_flush, const_iterator = heap over the files)
// T = a trivially copyable record, Vec = the vector type the caller fills, Key()(record) = the 64-bit sort key
// ---------------------------------------------------------------------------------------------
inline std::string tmp_name(const std::string& dir, const std::string& kind) {  // unique within the process (several builds may share a directory)
    static std::atomic<uint64_t> serial{0};
    return dir + "/lphash.tmp." + std::to_string((long)getpid()) + "." + std::to_string(serial.fetch_add(1)) + "." + kind;
}

template <typename T, typename Vec, typename Key>
struct Spill {
    std::string dir;
    const char* kind = "runs";  // file-name suffix (several instances live side by side in one directory)
    std::vector<std::string> files;
    static constexpr size_t kMaxPieces = 64;  // more sorted pieces than this are first merged into one (open files and read buffers stay bounded)
    ~Spill() { drop(); }
    void drop() {
        for (const std::string& f : files) ::remove(f.c_str());
        files.clear();
    }
    std::string fresh_name() { return tmp_name(dir, kind); }
    struct Reader {
        FILE* f = nullptr;
        Vec buf;
        size_t at = 0, n = 0;
        bool failed = false;
        bool open(const std::string& path, size_t buf_records) {
            f = fopen(path.c_str(), "rb");
            buf.resize(std::max<size_t>(1, buf_records));
            return f != nullptr;
        }
        bool next(T& r) {
            if (at == n) {
                if (!f) return false;
                n = fread((void*)buf.data(), sizeof(T), buf.size(), f);
                at = 0;
                if (n == 0) {
                    failed = ferror(f) != 0;
                    fclose(f);
                    f = nullptr;
                    return false;
                }
            }
            r = buf[at++];
            return true;
        }
        ~Reader() {
            if (f) fclose(f);
        }
    };
    // every record of `paths`, ascending by key (ties in no particular order), one next() at a time
    struct Merger {
        typedef std::pair<uint64_t, uint32_t> Head;  // (key, piece)
        std::vector<Reader> rd;
        std::priority_queue<Head, std::vector<Head>, std::greater<Head>> heap;
        Vec cur;
        const std::vector<std::string>* paths = nullptr;
        bool failed = false;
        std::string err;
        bool open(const std::vector<std::string>& p, size_t buf_records) {
            paths = &p;
            rd = std::vector<Reader>(p.size());
            cur.resize(p.size());
            for (size_t i = 0; i < p.size(); ++i) {
                if (!rd[i].open(p[i], buf_records)) {
                    err = "cannot reopen the temporary file " + p[i];
                    return !(failed = true);
                }
                if (rd[i].next(cur[i])) heap.emplace(Key()(cur[i]), (uint32_t)i);
                else if (rd[i].failed) {
                    err = "read error on the temporary file " + p[i];
                    return !(failed = true);
                }
            }
            return true;
        }
        bool next(T& out) {  // false: the end, or (failed) an error
            if (failed || heap.empty()) return false;
            const uint32_t i = heap.top().second;
            heap.pop();
            out = cur[i];
            if (rd[i].next(cur[i])) heap.emplace(Key()(cur[i]), i);
            else if (rd[i].failed) {
                err = "read error on the temporary file " + (*paths)[i];
                failed = true;  // (the record just taken is good; the next call reports the failure)
            }
            return true;
        }
    };
    template <typename Fn>
    static bool merge(const std::vector<std::string>& paths, size_t buf_records, std::string& err, Fn fn) {
        Merger mg;
        if (!mg.open(paths, buf_records)) {
            err = mg.err;
            return false;
        }
        T r;
        while (mg.next(r))
            if (!fn(r)) return false;
        if (mg.failed) {
            err = mg.err;
            return false;
        }
        return true;
    }
    bool write_piece(const Vec& recs, std::string& err) {
        const std::string name = fresh_name();
        FILE* f = fopen(name.c_str(), "wb");
        if (!f) {
            err = "cannot create the temporary file " + name;
            return false;
        }
        files.push_back(name);
        const size_t put = fwrite((const void*)recs.data(), sizeof(T), recs.size(), f);
        if (fclose(f) != 0 || put != recs.size()) {
            err = "short write on the temporary file " + name;
            return false;
        }
        return true;
    }
    // sort the buffered records and write them as one more piece; the buffer comes back empty
    bool flush(Vec& recs, int threads, uint32_t key_bits, size_t buf_records, std::string& err, bool sorted_already = false) {
        if (recs.empty()) return true;
        if (!sorted_already) parallel_sort_by_key(recs, threads, key_bits, [](const T& r) { return Key()(r); });
        if (!write_piece(recs, err)) return false;
        recs.clear();
        if (files.size() < kMaxPieces) return true;
        // too many pieces: merge them all into one
        const std::string name = fresh_name();
        FILE* f = fopen(name.c_str(), "wb");
        if (!f) {
            err = "cannot create the temporary file " + name;
            return false;
        }
        Vec out;
        out.reserve(buf_records);
        bool wrote = true;
        const bool ok = merge(files, buf_records, err, [&](const T& r) {
            out.push_back(r);
            if (out.size() == buf_records) {
                wrote = fwrite((const void*)out.data(), sizeof(T), out.size(), f) == out.size();
                out.clear();
            }
            return wrote;
        });
        if (ok && !out.empty()) wrote = fwrite((const void*)out.data(), sizeof(T), out.size(), f) == out.size();
        const bool closed = fclose(f) == 0;
        for (const std::string& p : files) ::remove(p.c_str());
        files.assign(1, name);
        if (!ok || !wrote || !closed) {
            if (err.empty()) err = "short write on the temporary file " + name;
            return false;
        }
        return true;
    }
};
// A sequential file of records (written once, read front to back any number of times): the per-minimizer vectors of a build with a memory limit.
template <typename T>
struct RecordFile {
    std::string path;
    FILE* f = nullptr;
    std::vector<T> buf;
    size_t at = 0, n = 0;
    uint64_t count = 0;
    bool failed = false, writing = false;
    ~RecordFile() {
        if (f) fclose(f);
        if (!path.empty()) ::remove(path.c_str());
    }
    bool create(const std::string& p, size_t buf_records) {
        path = p;
        buf.reserve(std::max<size_t>(1, buf_records));
        writing = true;
        return (f = fopen(p.c_str(), "wb")) != nullptr;
    }
    bool put(const T& r) {
        buf.push_back(r);
        ++count;
        if (buf.size() == buf.capacity()) return spill();
        return true;
    }
    bool spill() {
        const bool ok = fwrite((const void*)buf.data(), sizeof(T), buf.size(), f) == buf.size();
        buf.clear();
        return ok;
    }
    bool rewind_for_reading() {  // (also after a read pass)
        bool ok = true;
        if (writing) {
            ok = spill();
            writing = false;
        }
        if (f) ok = (fclose(f) == 0) && ok;
        f = fopen(path.c_str(), "rb");
        buf.resize(buf.capacity());
        at = n = 0;
        return ok && f;
    }
    bool next(T& r) {
        if (at == n) {
            n = fread((void*)buf.data(), sizeof(T), buf.size(), f);
            at = 0;
            if (n == 0) {
                failed = ferror(f) != 0;
                return false;
            }
        }
        r = buf[at++];
        return true;
    }
};

}  // namespace lphq
