"""ctypes binding of the CPU oracle (oracle/liblphoracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.environ.get("LPHQ_ORACLE_LIB") or os.path.join(_ROOT, "oracle", "liblphoracle.so")  # override: e.g. a sanitizer build

PARTITIONED, UNPARTITIONED = 0, 1


def _lib():
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")])
    lib = C.CDLL(_SO)
    lib.orc_load.restype = C.c_void_p
    lib.orc_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
    lib.orc_free.argtypes = [C.c_void_p]
    lib.orc_last_error.restype = C.c_char_p
    for f in ("orc_k", "orc_m", "orc_mm_seed", "orc_kmer_count", "orc_minimizer_L0", "orc_fallback_keys"):
        getattr(lib, f).restype = C.c_uint64
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.orc_query.restype = C.c_int64
    lib.orc_query.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
    lib.orc_query_packed.restype = None
    lib.orc_query_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.orc_build.restype = C.c_int
    lib.orc_build.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int,
                              C.c_uint64, C.c_double, C.c_int, C.c_int, C.c_char_p]
    lib.orc_query_file.restype = C.c_int
    lib.orc_query_file.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.orc_file_digest.restype = C.c_int
    lib.orc_file_digest.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.orc_ptbb_build.restype = C.c_int
    lib.orc_ptbb_build.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_char_p]
    lib.orc_ptbb_load.restype = C.c_void_p
    lib.orc_ptbb_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
    lib.orc_ptbb_free.argtypes = [C.c_void_p]
    lib.orc_ptbb_num_keys.restype = C.c_uint64
    lib.orc_ptbb_num_keys.argtypes = [C.c_void_p]
    lib.orc_ptbb_query_packed.restype = None
    lib.orc_ptbb_query_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.orc_ptbb_query_contig.restype = C.c_int64
    lib.orc_ptbb_query_contig.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.orc_murmur64a.restype = C.c_uint64
    lib.orc_murmur64a.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
    lib.orc_minimizer_order.restype = C.c_uint64
    lib.orc_minimizer_order.argtypes = [C.c_void_p, C.c_uint64]
    return lib


lib = _lib()


def build(contigs, k, m, out_path, kind=PARTITIONED, kmer_bits=64, mm_seed=42, c=3.0):
    """contigs: list of bytes (one record each, what kseq_read yields)."""
    n = len(contigs)
    arr = (C.c_char_p * n)(*contigs)
    lens = (C.c_size_t * n)(*[len(s) for s in contigs])
    rc = lib.orc_build(arr, lens, n, k, m, mm_seed, c, kind, kmer_bits, out_path.encode())
    if rc != 0:
        raise RuntimeError(lib.orc_last_error().decode())


def file_digest(path):
    """(records, bases, FNV-1a digest) of a FASTA/FASTQ(.gz) file as the oracle driver's reader parses it."""
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    if lib.orc_file_digest(path.encode(), C.byref(a), C.byref(b), C.byref(c)) != 0:
        raise RuntimeError("cannot open " + path)
    return a.value, b.value, c.value


def ptbb_build(contigs, k, out_path, kmer_bits=64, c=3.0):
    """ptbb_build -i ... -k k -p out_path (PTHash half of the competitor harness, src/ptbb/ptbb_build.cpp:83-99)."""
    n = len(contigs)
    arr = (C.c_char_p * n)(*contigs)
    lens = (C.c_size_t * n)(*[len(s) for s in contigs])
    if lib.orc_ptbb_build(arr, lens, n, k, kmer_bits, c, out_path.encode()) != 0:
        raise RuntimeError(lib.orc_last_error().decode())


class PtHash:
    """pthash_mphf_t of src/ptbb/ptbb.hpp:33 loaded from a file (essentials::load)."""

    def __init__(self, path, k, kmer_bits=64):
        self.h = lib.orc_ptbb_load(path.encode(), k, kmer_bits)
        if not self.h:
            raise RuntimeError(lib.orc_last_error().decode())
        self.k, self.kmer_bits = k, kmer_bits
        self.num_keys = lib.orc_ptbb_num_keys(self.h)

    def __call__(self, contig: bytes):
        cap = max(len(contig), 1)
        out = np.empty(cap, dtype=np.uint64)
        n = lib.orc_ptbb_query_contig(self.h, contig, len(contig), out.ctypes.data, cap)
        return out[:n].copy()

    def query_packed(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        n = kmers.size // (1 if self.kmer_bits == 64 else 2)
        out = np.empty(n, dtype=np.uint64)
        lib.orc_ptbb_query_packed(self.h, kmers.ctypes.data, n, out.ctypes.data)
        return out

    def __del__(self):
        try:
            if self.h:
                lib.orc_ptbb_free(self.h)
                self.h = None
        except Exception:
            pass


class Index:
    def __init__(self, path, kind=PARTITIONED, kmer_bits=64):
        self.h = lib.orc_load(path.encode(), kind, kmer_bits)
        if not self.h:
            raise RuntimeError(lib.orc_last_error().decode())
        self.kind, self.kmer_bits = kind, kmer_bits
        self.k, self.m = lib.orc_k(self.h), lib.orc_m(self.h)
        self.mm_seed = lib.orc_mm_seed(self.h)
        self.nkmers = lib.orc_kmer_count(self.h)
        self.distinct_minimizers = lib.orc_minimizer_L0(self.h)
        self.fallback_keys = lib.orc_fallback_keys(self.h)

    def __call__(self, contig: bytes, streaming=True):
        cap = max(len(contig), 1)
        out = np.empty(cap, dtype=np.uint64)
        n = lib.orc_query(self.h, contig, len(contig), 1 if streaming else 0, out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError(lib.orc_last_error().decode())
        assert n <= cap
        return out[:n].copy()

    def query_packed(self, kmers: np.ndarray):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        wpk = 1 if self.kmer_bits == 64 else 2
        n = kmers.size // wpk
        out = np.empty(n, dtype=np.uint64)
        lib.orc_query_packed(self.h, kmers.ctypes.data, n, out.ctypes.data)
        return out

    def query_packed_threads(self, kmers: np.ndarray, threads: int, out=None):
        """query_packed over contiguous slices on POSIX threads (no interpreter in the loop); out: a uint64 array to fill (a timed caller passes
        one whose pages exist already: first-touch faults of a fresh GB-sized array from hundreds of threads would be what is measured)"""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        n = kmers.size // (1 if self.kmer_bits == 64 else 2)
        if out is None:
            out = np.empty(n, dtype=np.uint64)
        assert out.dtype == np.uint64 and out.size >= n and out.flags["C_CONTIGUOUS"]
        lib.orc_query_packed_threads.restype = C.c_int
        lib.orc_query_packed_threads.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        if lib.orc_query_packed_threads(self.h, kmers.ctypes.data, n, out.ctypes.data, int(threads)) != 0:
            raise RuntimeError("orc_query_packed_threads failed")
        return out[:n]

    def query_file(self, path):
        """The reference's driver (src/query.cpp:24-88): returns (k-mers streaming, k-mers random, us streaming, us random)."""
        ns, nr, us, ur = C.c_uint64(), C.c_uint64(), C.c_double(), C.c_double()
        rc = lib.orc_query_file(self.h, path.encode(), C.byref(ns), C.byref(nr), C.byref(us), C.byref(ur))
        if rc != 0:
            raise RuntimeError(f"Unable to open the input file {path}")
        return ns.value, nr.value, us.value, ur.value

    def minimizer_order(self, mm):
        return lib.orc_minimizer_order(self.h, mm)

    def close(self):
        if self.h:
            lib.orc_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
