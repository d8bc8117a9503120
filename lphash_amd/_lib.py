"""ctypes binding of liblphq.so (the C ABI declared in include/lphq.h).

The shared library holds the loader, the flat-image builder, the index builder and the gfx950 kernels.
There is no Python or CPU implementation of the query path: if the library is missing, importing this
module raises, and every query call needs a HIP device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LPHQ_LIB_PATH") or os.path.join(_HERE, "liblphq.so")  # override: A/B of two builds (tools/ab_lib.sh), sanitizer builds

OK, ERR_ARG, ERR_IO, ERR_FORMAT, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED = range(8)
PARTITIONED, UNPARTITIONED, PTHASH, BBHASH = 0, 1, 2, 3
INVALID = 0xFFFFFFFFFFFFFFFF
RUN_ARENAS = 32  # LPHQ_RUN_ARENAS; an lphq_run is 16 bytes (code, where), an lphq_run_counts 128 (runs, codes, 14 reserved words)
RUN_UP, RUN_DOWN, RUN_EXPLICIT = 0, 1, 2


class LphqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lphq error {code}: {msg}")
        self.code = code


class Info(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32), ("kmer_bits", C.c_uint32), ("k", C.c_uint32), ("m", C.c_uint32),
        ("mm_seed", C.c_uint64), ("num_kmers", C.c_uint64), ("distinct_minimizers", C.c_uint64),
        ("fallback_kmers", C.c_uint64), ("file_bytes", C.c_uint64), ("image_bytes", C.c_uint64),
        ("device", C.c_int32), ("reserved", C.c_uint32),
        ("kmers_by_type", C.c_uint64 * 4), ("minimizer_table_size", C.c_uint64), ("part_bits", C.c_uint64 * 4), ("ef_entries", C.c_uint64),
    ]


class BuildStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("num_kmers", "num_records", "num_super_kmers", "distinct_minimizers",
                                          "colliding_super_kmers", "fallback_kmers", "file_bytes")]


# every symbol include/lphq.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "lphq_last_error": (C.c_char_p, []),
    "lphq_open": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "lphq_open_memory": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "lphq_close": (None, [C.c_void_p]),
    "lphq_get_info": (C.c_int, [C.c_void_p, C.POINTER(Info)]),
    "lphq_image_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "lphq_image_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "lphq_open_device_image": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "lphq_attach": (C.c_int, [C.c_void_p, C.c_int]),
    "lphq_attach_devices": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "lphq_attached_devices": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    "lphq_device_image": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "lphq_query_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "lphq_query_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p]),
    "lphq_pack_kmers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "lphq_count_valid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "lphq_count_valid_on_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64)]),
    "lphq_abi_version": (C.c_int, []),
    "lphq_gather_payload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "lphq_count_valid_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "lphq_query_records_runs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_void_p]),
    "lphq_expand_runs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "lphq_query_records_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p,
                                            C.c_void_p]),
    "lphq_query_contig": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                    C.POINTER(C.c_size_t)]),
    "lphq_build": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int, C.c_uint64,
                             C.c_double, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(BuildStats)]),
    "lphq_build_device": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int, C.c_uint64,
                                    C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(BuildStats)]),
    "lphq_build_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                    C.c_uint64, C.c_char_p, C.POINTER(BuildStats)]),
    "lphq_open_bbhash": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "lphq_build_bbhash": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_int, C.c_char_p,
                                    C.POINTER(BuildStats)]),
    "lphq_session_open": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "lphq_session_open_devices": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "lphq_session_close": (None, [C.c_void_p]),
    "lphq_session_reserve": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t]),
    "lphq_session_submit": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int]),
    "lphq_session_collect": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t)]),
    "lphq_session_submit_runs": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t]),
    "lphq_session_collect_runs": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t)]),
    "lphq_host_alloc": (C.c_void_p, [C.c_size_t]),
    "lphq_host_free": (None, [C.c_void_p]),
    "lphq_replica_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lphq_open_pthash": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "lphq_build_pthash": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                    C.c_char_p, C.POINTER(BuildStats)]),
}


# lphq_record_source: int (*)(void* ctx, int pass, const char* const** records, const size_t** lengths, size_t* n_records)
RECORD_SOURCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_size_t)), C.POINTER(C.c_size_t))


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7) and ask for it by file name, so if liblphq.so pulled in /opt/rocm's copy first a
    later `import torch` would load a second runtime that sees no device.  Loading torch's copy first
    (without importing torch) makes both resolve to the same object, whatever the import order."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C lphash_amd/csrc`). There is no fallback implementation.")
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


def check(rc):
    if rc != OK:
        raise LphqError(rc, (lib.lphq_last_error() or b"").decode(errors="replace"))
