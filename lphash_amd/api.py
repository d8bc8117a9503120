"""Host-side mirror of the reference's MPHF interface for the query path.

Reference (paths relative to the reference tree):
  class mphf      include/partitioned_mphf.hpp:10-71   -> Mphf      (`lphash query-p`)
  class mphf_alt  include/unpartitioned_mphf.hpp:10-69 -> MphfAlt   (`lphash query-u`)
  essentials::load(hf, path)  src/query.cpp:35-37      -> Mphf.load(path)
  hf(contig, length, streaming)                        -> hf(contig, streaming=True) -> np.ndarray[uint64]
  hf.get_kmer_count(), hf.get_minimizer_L0()           same names

Everything that computes goes through the C ABI (include/lphq.h) to the gfx950 kernels; PyTorch is
only used by callers for device memory, streams and torch.distributed.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BBHASH, PARTITIONED, PTHASH, UNPARTITIONED, LphqError, check, lib


def build(records, k, m, out_path, kind=PARTITIONED, kmer_bits=None, mm_seed=42, c=3.0, threads=0, device=None):
    """`lphash build-p|build-u -i ... -k k -m m -s seed -c c -t threads -o out_path` over in-memory records
    (src/build.cpp:10-51).  records: iterable of bytes, one per FASTA/FASTQ record (what kseq_read yields).
    device: HIP ordinal for the PTHash pilot search (None = CPU search); the file written is the same."""
    records = [bytes(r) for r in records]
    if kmer_bits is None:
        kmer_bits = 64 if k <= 31 else 128
    n = len(records)
    arr = (C.c_char_p * n)(*records)
    lens = (C.c_size_t * n)(*[len(r) for r in records])
    stats = _lib.BuildStats()
    check(lib.lphq_build_device(arr, lens, n, int(k), int(m), int(mm_seed), float(c), int(kind), int(kmer_bits), int(threads),
                                -1 if device is None else int(device), str(out_path).encode(), C.byref(stats)))
    return {name: getattr(stats, name) for name, _ in stats._fields_}


def build_stream(batches, k, m, out_path, kind=PARTITIONED, kmer_bits=None, mm_seed=42, c=3.0, threads=0, device=None, tmp_dir=None,
                 max_memory_bytes=0):
    """`lphash build-p|build-u ... -d tmp_dir --max-memory G` (src/partitioned_mphf.cpp:58-130, include/external_memory_vector.hpp):
    construction from a stream of record batches with bounded memory.  batches: a callable returning a fresh iterator over batches
    (lists of bytes, whole records) -- it is called twice, once per pass over the input.  Same file as build() on the same records."""
    if kmer_bits is None:
        kmer_bits = 64 if k <= 31 else 128
    state = {"pass": -1, "it": None, "keep": None, "error": None}

    def source(_ctx, pas, p_recs, p_lens, p_n):
        try:
            if state["pass"] != pas:
                state["pass"], state["it"] = pas, iter(batches())
            batch = next(state["it"], None)
            if batch is None:
                return 0
            recs = [bytes(r) for r in batch]
            n = len(recs)
            arr = (C.c_char_p * max(n, 1))(*recs)
            lens = (C.c_size_t * max(n, 1))(*[len(r) for r in recs])
            state["keep"] = (recs, arr, lens)  # valid until the next call
            p_recs[0] = C.cast(arr, C.POINTER(C.c_char_p))
            p_lens[0] = C.cast(lens, C.POINTER(C.c_size_t))
            p_n[0] = n
            return 1
        except Exception as e:  # never unwind through the C frames
            state["error"] = e
            return -1

    cb = _lib.RECORD_SOURCE(source)
    stats = _lib.BuildStats()
    rc = lib.lphq_build_stream(C.cast(cb, C.c_void_p), None, int(k), int(m), int(mm_seed), float(c), int(kind), int(kmer_bits), int(threads),
                               -1 if device is None else int(device), None if tmp_dir is None else str(tmp_dir).encode(), int(max_memory_bytes),
                               str(out_path).encode(), C.byref(stats))
    if state["error"] is not None:
        raise state["error"]
    check(rc)
    return {name: getattr(stats, name) for name, _ in stats._fields_}


def build_pthash(records, k, out_path, kmer_bits=None, c=3.0, threads=0, device=None):
    """`ptbb_build -i ... -k k -p out_path [-c c] [-t threads]` (PTHash half of the reference's competitor harness,
    src/ptbb/ptbb_build.cpp:83-99) over in-memory records: a bare pthash::single_phf over every window of k valid bases."""
    records = [bytes(r) for r in records]
    if kmer_bits is None:
        kmer_bits = 64 if k <= 31 else 128
    n = len(records)
    arr = (C.c_char_p * n)(*records)
    lens = (C.c_size_t * n)(*[len(r) for r in records])
    stats = _lib.BuildStats()
    check(lib.lphq_build_pthash(arr, lens, n, int(k), int(kmer_bits), float(c), int(threads), -1 if device is None else int(device),
                                str(out_path).encode(), C.byref(stats)))
    return {name: getattr(stats, name) for name, _ in stats._fields_}


def build_bbhash(records, k, out_path, kmer_bits=None, gamma=1.0, threads=0):
    """`ptbb_build -i ... -k k -b out_path [-g gamma] [-t threads]` (BBHash half of the reference's competitor harness,
    src/ptbb/ptbb_build.cpp:140-199) over in-memory records: a file boomphf::mphf::load accepts."""
    records = [bytes(r) for r in records]
    if kmer_bits is None:
        kmer_bits = 64 if k <= 31 else 128
    n = len(records)
    arr = (C.c_char_p * n)(*records)
    lens = (C.c_size_t * n)(*[len(r) for r in records])
    stats = _lib.BuildStats()
    check(lib.lphq_build_bbhash(arr, lens, n, int(k), int(kmer_bits), float(gamma), int(threads), str(out_path).encode(), C.byref(stats)))
    return {name: getattr(stats, name) for name, _ in stats._fields_}


class _MphfBase:
    KIND = None

    def __init__(self):
        self._h = C.c_void_p()
        self._info = None

    # -- essentials::load ---------------------------------------------------------------------
    @classmethod
    def load(cls, path, kmer_bits=128, device=None):
        """kmer_bits: the reference's compile-time kmer_t width (include/compile_constants.tpd:1-2, default
        __uint128_t).  device: HIP ordinal to attach to (None = do not attach yet)."""
        self = cls()
        check(lib.lphq_open(str(path).encode(), cls.KIND, int(kmer_bits), C.byref(self._h)))
        self._refresh()
        if device is not None:
            self.attach(device)
        return self

    @classmethod
    def from_bytes(cls, data, kmer_bits=128, device=None):
        self = cls()
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        check(lib.lphq_open_memory(buf, len(data), cls.KIND, int(kmer_bits), C.byref(self._h)))
        self._refresh()
        if device is not None:
            self.attach(device)
        return self

    @classmethod
    def from_device_image(cls, d_image_ptr, nbytes, device):
        """Wrap a flat image already resident on `device` (e.g. received by an RCCL broadcast)."""
        self = cls()
        check(lib.lphq_open_device_image(C.c_void_p(d_image_ptr), nbytes, int(device), C.byref(self._h)))
        self._refresh()
        if self._info.kind != cls.KIND:
            self.close()
            raise LphqError(_lib.ERR_ARG, "image kind does not match the class")
        return self

    def _refresh(self):
        info = _lib.Info()
        check(lib.lphq_get_info(self._h, C.byref(info)))
        self._info = info

    def attach(self, device=0):
        check(lib.lphq_attach(self._h, int(device)))
        self._refresh()
        return self

    def attach_devices(self, devices):
        """Replicas on several GPUs of one process (lphq_attach_devices: upload to the first, peer copies to the others)."""
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        check(lib.lphq_attach_devices(self._h, devs, len(devices)))
        self._refresh()
        return self

    def attached_devices(self):
        n = C.c_int()
        devs = (C.c_int * 64)()
        check(lib.lphq_attached_devices(self._h, devs, 64, C.byref(n)))
        return [devs[i] for i in range(min(n.value, 64))]

    def count_valid(self, d_codes, n, stream=0):
        """Codes != LPHQ_INVALID among d_codes[0..n) (device pointer), counted on `stream`."""
        c = C.c_uint64()
        if stream:
            check(lib.lphq_count_valid_on_stream(self._h, C.c_void_p(d_codes), n, C.c_void_p(stream), C.byref(c)))
        else:
            check(lib.lphq_count_valid(self._h, C.c_void_p(d_codes), n, C.byref(c)))
        return c.value

    def gather_payload(self, d_codes, n, d_table, elem_bytes, d_out, missing=0, stream=0):
        """d_out[i] = d_table[d_codes[i]] on the device (codes that are not slots of the function -> `missing`)."""
        check(lib.lphq_gather_payload(self._h, C.c_void_p(d_codes), n, C.c_void_p(d_table), int(elem_bytes), int(missing) & 0xFFFFFFFFFFFFFFFF,
                                      C.c_void_p(d_out), C.c_void_p(stream)))

    def close(self):
        if self._h:
            lib.lphq_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- accessors ------------------------------------------------------------------------------
    @property
    def info(self):
        return self._info

    @property
    def k(self):
        return self._info.k

    @property
    def m(self):
        return self._info.m

    @property
    def kmer_bits(self):
        return self._info.kmer_bits

    def get_kmer_count(self):
        return self._info.num_kmers

    def get_minimizer_L0(self):
        return self._info.distinct_minimizers

    # -- flat image -----------------------------------------------------------------------------
    def image(self):
        n = C.c_size_t()
        check(lib.lphq_image_bytes(self._h, C.byref(n)))
        out = np.empty(n.value, dtype=np.uint8)
        check(lib.lphq_image_copy(self._h, out.ctypes.data, n.value))
        return out

    def device_image(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(lib.lphq_device_image(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    # -- MPHF::operator()(contig, length, streaming) ------------------------------------------------
    def __call__(self, contig, streaming=True):
        if isinstance(contig, str):
            contig = contig.encode()
        contig = bytes(contig)
        cap = max(len(contig), 1)
        out = np.empty(cap, dtype=np.uint64)
        n = C.c_size_t()
        check(lib.lphq_query_contig(self._h, contig, len(contig), 1 if streaming else 0, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].copy()

    # -- MPHF::operator() over many records per call, batches in flight (lphq_session_*) -----------
    def session(self, depth=2, devices=None):
        """devices: HIP ordinals of attached replicas the session spreads its batches over (None: one replica, as lphq_session_open picks it)."""
        return Session(self, depth, devices)

    def replica_info(self):
        """[(device, how)] in attach order; how: 0 uploaded, 1 peer copy with peer access, 2 peer copy staged by the runtime, 3 adopted."""
        out = []
        for i in range(len(self.attached_devices())):
            d, h = C.c_int(), C.c_int()
            check(lib.lphq_replica_info(self._h, i, C.byref(d), C.byref(h)))
            out.append((d.value, h.value))
        return out

    # -- device-resident batches (pointers are raw device addresses, e.g. tensor.data_ptr()) -------
    def query_packed(self, d_kmers, n, d_out, stream=0):
        check(lib.lphq_query_packed(self._h, C.c_void_p(d_kmers), n, C.c_void_p(d_out), C.c_void_p(stream)))

    def query_records(self, d_seq, total_bases, d_seq_offsets, n_records, d_out, d_out_offsets, streaming=True, stream=0):
        check(lib.lphq_query_records(self._h, C.c_void_p(d_seq), total_bases, C.c_void_p(d_seq_offsets), n_records,
                                     C.c_void_p(d_out), C.c_void_p(d_out_offsets), 1 if streaming else 0, C.c_void_p(stream)))

    def query_records_runs(self, d_seq, total_bases, d_seq_offsets, n_records, d_out_offsets, d_runs, cap_runs, d_codes, cap_codes, d_counts, stream=0):
        """The streaming pass with the run-encoded result (lphq_query_records_runs): 16-byte lphq_run entries in RUN_ARENAS arenas of cap_runs / RUN_ARENAS
        entries, explicit codes of colliding windows in arenas of cap_codes / RUN_ARENAS, RUN_ARENAS x 128-byte counters at d_counts."""
        check(lib.lphq_query_records_runs(self._h, C.c_void_p(d_seq), total_bases, C.c_void_p(d_seq_offsets), n_records, C.c_void_p(d_out_offsets),
                                          C.c_void_p(d_runs), cap_runs, C.c_void_p(d_codes), cap_codes, C.c_void_p(d_counts), C.c_void_p(stream)))

    def expand_runs(self, d_runs, cap_runs, d_codes, cap_codes, d_counts, d_out, n_out, d_status=0, stream=0):
        """d_out[0..n_out) from the runs: what query_records(streaming=True) writes for the same batch (lphq_expand_runs)."""
        check(lib.lphq_expand_runs(self._h, C.c_void_p(d_runs), cap_runs, C.c_void_p(d_codes), cap_codes, C.c_void_p(d_counts), C.c_void_p(d_out), n_out,
                                   C.c_void_p(d_status), C.c_void_p(stream)))

    def query_records_gather(self, d_seq, total_bases, d_seq_offsets, n_records, d_out_offsets, d_table, elem_bytes, d_payload_out, missing=0, stream=0):
        """d_payload_out[window] = d_table[code of the window], streaming lookup and gather in one kernel (lphq_query_records_gather)."""
        check(lib.lphq_query_records_gather(self._h, C.c_void_p(d_seq), total_bases, C.c_void_p(d_seq_offsets), n_records, C.c_void_p(d_out_offsets),
                                            C.c_void_p(d_table), int(elem_bytes), int(missing) & 0xFFFFFFFFFFFFFFFF, C.c_void_p(d_payload_out), C.c_void_p(stream)))

    def pack_kmers(self, d_seq, d_pos, n, d_kmers, stream=0):
        check(lib.lphq_pack_kmers(self._h, C.c_void_p(d_seq), C.c_void_p(d_pos), n, C.c_void_p(d_kmers), C.c_void_p(stream)))


class Session:
    """Batches of host records through one index, `depth` of them in flight (include/lphq.h: lphq_session_*).  submit() queues a batch (copy in,
    kernel, copy out, nothing waited for); collect() waits for the oldest batch and returns one uint64 array per record -- what a loop over
    MPHF::operator() would have returned."""

    def __init__(self, mphf, depth=2, devices=None):
        self._s = C.c_void_p()
        self._mphf = mphf  # the session uses the index's device image: keep the index alive as long as the session (lphq.h: close sessions first)
        self._keep = []  # the records of the batches in flight (their memory must outlive the copies)
        if devices is None:
            check(lib.lphq_session_open(mphf._h, depth, C.byref(self._s)))
        else:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            check(lib.lphq_session_open_devices(mphf._h, depth, devs, len(devices), C.byref(self._s)))

    def reserve(self, total_bases, n_records):
        check(lib.lphq_session_reserve(self._s, total_bases, n_records))

    def submit(self, records, streaming=True):
        recs = [bytes(r) for r in records]
        n = len(recs)
        ptrs = (C.c_char_p * max(n, 1))(*recs)
        lens = (C.c_size_t * max(n, 1))(*[len(r) for r in recs])
        check(lib.lphq_session_submit(self._s, ptrs, lens, n, 1 if streaming else 0))
        self._keep.append((recs, ptrs, lens))

    def submit_runs(self, records):
        """submit(streaming=True) whose result is collected as runs (lphq_session_submit_runs)."""
        recs = [bytes(r) for r in records]
        n = len(recs)
        ptrs = (C.c_char_p * max(n, 1))(*recs)
        lens = (C.c_size_t * max(n, 1))(*[len(r) for r in recs])
        check(lib.lphq_session_submit_runs(self._s, ptrs, lens, n))
        self._keep.append((recs, ptrs, lens))

    def collect_runs(self):
        """The oldest batch as the reference computes it: (runs, explicit_codes, n_windows, offsets) -- runs: structured array (code, where) of lphq_run
        entries in one dense array, offsets: where each record's windows begin in the space LPHQ_RUN_FIRST counts in (lphq_session_collect_runs)."""
        runs, nr, xc, nx = C.c_void_p(), C.c_size_t(), C.POINTER(C.c_uint64)(), C.c_size_t()
        nw, offs, n = C.c_uint64(), C.POINTER(C.c_uint64)(), C.c_size_t()
        check(lib.lphq_session_collect_runs(self._s, C.byref(runs), C.byref(nr), C.byref(xc), C.byref(nx), C.byref(nw), C.byref(offs), C.byref(n)))
        self._keep.pop(0)
        dt = np.dtype([("code", "<u8"), ("where", "<u8")])
        r = np.frombuffer((C.c_char * (nr.value * 16)).from_address(runs.value), dtype=dt).copy() if nr.value else np.zeros(0, dtype=dt)
        x = np.ctypeslib.as_array(xc, shape=(nx.value,)).copy() if nx.value else np.zeros(0, dtype=np.uint64)
        o = np.ctypeslib.as_array(offs, shape=(n.value + 1,)).copy() if n.value else np.zeros(1, dtype=np.uint64)
        return r, x, int(nw.value), o

    def collect(self):
        codes, offs, n = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.c_size_t()
        check(lib.lphq_session_collect(self._s, C.byref(codes), C.byref(offs), C.byref(n)))
        self._keep.pop(0)
        o = np.ctypeslib.as_array(offs, shape=(n.value + 1,)).copy() if n.value else np.zeros(1, dtype=np.uint64)
        total = int(o[-1])
        flat = np.ctypeslib.as_array(codes, shape=(total,)).copy() if total else np.zeros(0, dtype=np.uint64)
        return [flat[int(o[i]):int(o[i + 1])] for i in range(n.value)]

    def close(self):
        if self._s:
            lib.lphq_session_close(self._s)
            self._s = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mphf(_MphfBase):
    """Partitioned LP-MPHF (reference class `mphf`, tool `query-p`)."""
    KIND = PARTITIONED


class MphfAlt(_MphfBase):
    """Unpartitioned LP-MPHF (reference class `mphf_alt`, tool `query-u`)."""
    KIND = UNPARTITIONED


class PtHash(_MphfBase):
    """Plain PTHash MPHF over whole k-mers (`pthash_mphf_t`, src/ptbb/ptbb.hpp:33; tool `ptbb_query`): the competitor the reference
    measures LPHash against.  hf(contig) yields one code per window of k valid bases (ptbb_file_itr, src/ptbb/ptbb.hpp:108-131)."""
    KIND = PTHASH

    @classmethod
    def load(cls, path, k, kmer_bits=128, device=None):
        self = cls()
        check(lib.lphq_open_pthash(str(path).encode(), int(k), int(kmer_bits), C.byref(self._h)))
        self._refresh()
        if device is not None:
            self.attach(device)
        return self


class BbHash(PtHash):
    """BBHash MPHF over whole k-mers (`bbhash_mphf_t`, src/ptbb/ptbb.hpp:34): the reference's second competitor."""
    KIND = BBHASH

    @classmethod
    def load(cls, path, k, kmer_bits=128, device=None):
        self = cls()
        check(lib.lphq_open_bbhash(str(path).encode(), int(k), int(kmer_bits), C.byref(self._h)))
        self._refresh()
        if device is not None:
            self.attach(device)
        return self
