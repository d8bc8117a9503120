"""The run-encoded streaming result and the fused lookup -> payload pass (include/lphq.h, ABI 4) on the CPU: the device sources
(stream8_device.cuh sinks kSinkRuns / kSinkGather, k_expand_runs, k_compact_runs) compiled for the host and run by the wave emulator
(tests/emu, test infrastructure), against the oracle's streaming codes -- the reference's case-0 updates, include/partitioned_mphf.hpp:131-145.
The same checks run on the GPU through the C ABI in tests/test_gpu_runs.py."""
import ctypes as C
import os

import numpy as np
import pytest

import lphash_amd as L
import orc
from runs_util import ARENAS, RUN_DTYPE, COUNTS_DTYPE, batch_layout, check_runs_against_codes, expand_runs_numpy
from test_emu_cpu import emu, emuk  # noqa: F401  (fixtures)
from util import dedup_records, random_records

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emur(emuk):
    emuk.emu_kernel_query_records_runs.restype = C.c_int
    emuk.emu_kernel_query_records_runs.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                                                   C.c_uint64, C.c_void_p]
    emuk.emu_kernel_query_records_gather.restype = C.c_int
    emuk.emu_kernel_query_records_gather.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32,
                                                     C.c_uint64, C.c_void_p]
    emuk.emu_kernel_expand_runs.restype = C.c_int
    emuk.emu_kernel_expand_runs.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_ulonglong)]
    emuk.emu_kernel_compact_runs.restype = C.c_int
    emuk.emu_kernel_compact_runs.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    return emuk


def emu_runs(lib, h, recs, k, misalign=0, cap_runs=None, cap_codes=None):
    """lphq_query_records_runs on the emulator -> (runs[ARENAS, arena_runs], codes[ARENAS, arena_codes], counts, out_off)."""
    img = h.image()
    seq_off, out_off, raw, base, nflat = batch_layout(recs, k, misalign)
    n_out = int(out_off[-1])
    # worst-case arenas (every window a run of its own, all tiles of a small batch in different arenas): min(n, 1024 x tiles per arena) entries each
    per_arena = min(n_out, 1024 * (n_out // (ARENAS * 900) + 1)) + 8
    cap_runs = cap_runs if cap_runs is not None else ARENAS * per_arena
    cap_codes = cap_codes if cap_codes is not None else ARENAS * per_arena
    runs = np.zeros(cap_runs + 1, dtype=RUN_DTYPE)
    runs[-1]["code"] = 0xDEAD
    codes = np.full(cap_codes + 1, 0xDEAD, dtype=np.uint64)
    counts = np.zeros(ARENAS, dtype=COUNTS_DTYPE)
    rc = lib.emu_kernel_query_records_runs(img.ctypes.data, img.size, raw.ctypes.data + base, nflat, seq_off.ctypes.data, out_off.ctypes.data, len(recs), runs.ctypes.data,
                                           cap_runs, codes.ctypes.data, cap_codes, counts.ctypes.data)
    assert rc == 0 and runs[-1]["code"] == 0xDEAD and codes[-1] == 0xDEAD
    return runs[:-1].reshape(ARENAS, -1), codes[:-1].reshape(ARENAS, -1), counts, out_off


def emu_expand(lib, runs, codes, counts, n_out):
    out = np.full(n_out + 1, 0xBEEF, dtype=np.uint64)
    st = C.c_ulonglong(7)
    r, c = np.ascontiguousarray(runs.reshape(-1)), np.ascontiguousarray(codes.reshape(-1))
    assert lib.emu_kernel_expand_runs(r.ctypes.data, r.size, c.ctypes.data, c.size, counts.ctypes.data, out.ctypes.data, n_out, C.byref(st)) == 0
    assert out[-1] == 0xBEEF
    return out[:-1], st.value


def streaming_expectation(o, recs, k, with_valid=False):
    """What lphq_query_records(streaming = 1) writes for the batch: the oracle's codes, LPHQ_INVALID where a window holds a non-ACGT base (with_valid: also
    the bool per window that says so -- decided from the bases: a non-member's wrapped code may itself be 2^64 - 1)."""
    from real_fixture import valid_windows

    parts, vparts = [], []
    for q in recs:
        if len(q) < k:
            continue
        e = o(q, False).copy()  # random branch == streaming branch on clean windows (src/build.cpp:53-70); a non-ACGT window is INVALID in the device output
        v = valid_windows(q, k)
        e[~v] = np.uint64(L.INVALID)
        parts.append(e)
        vparts.append(v)
    exp = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)
    return (exp, np.concatenate(vparts) if vparts else np.zeros(0, dtype=bool)) if with_valid else exp


CONFIGS = [(31, 20, 64, 0), (63, 20, 128, 0), (31, 20, 64, 1), (21, 8, 64, 0), (47, 17, 128, 1), (31, 31, 64, 0), (15, 9, 64, 0), (9, 7, 64, 0),
           (15, 8, 64, 0), (35, 20, 128, 0), (31, 15, 64, 0), (37, 20, 128, 0), (38, 20, 128, 1), (41, 20, 128, 0), (33, 5, 128, 0)]


@pytest.mark.parametrize("k,m,bits,kind", CONFIGS)
def test_runs_and_gather_on_the_emulator(emur, tmp_path, k, m, bits, kind):
    """Ragged multi-record batches (empty and short records, > 64 records in a tile, tile-straddling records, non-members, non-ACGT bytes) through the
    runs sink and the gather sink, every (W - 8) % 8 instantiation and the narrow-window form down to W = 1."""
    recs = dedup_records(random_records(k * 19 + m, 14 if k > 12 else 60, k, 2600 if k > 12 else k + 12), k)
    path = str(tmp_path / "t.lph")
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=4)
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits)
    noisy = bytearray(recs[3])
    noisy[len(noisy) // 2] = ord("N")
    noisy[5:5 + k + 3] = b"n" * (k + 3)
    batch = [b"", recs[0], recs[1][: k - 1], b"", recs[2][:k], bytes(noisy)] + recs[3:] + random_records(3, 6, 1, 1500) + [b""]
    batch += random_records(11, 150, 0, 5) + [recs[0][:k + 2]] + random_records(12, 40, k, k + 20) + [recs[1]]
    exp, valid = streaming_expectation(o, batch, k, with_valid=True)
    for mis in (0, 7):
        runs, codes, counts, out_off = emu_runs(emur, h, batch, k, mis)
        n_out = int(out_off[-1])
        assert n_out == len(exp)
        check_runs_against_codes(runs, codes, counts, exp, L.INVALID, valid)  # every valid window in exactly one run, affine as claimed, never across records
        assert np.array_equal(expand_runs_numpy(runs, codes, counts, n_out, L.INVALID), exp)
        got, st = emu_expand(emur, runs, codes, counts, n_out)
        assert st == 0 and np.array_equal(got, exp)
    # the dense form a session ships over PCIe
    r, c = np.ascontiguousarray(runs.reshape(-1)), np.ascontiguousarray(codes.reshape(-1))
    nr, nx = int(counts["runs"].sum()), int(counts["codes"].sum())
    dense_r, dense_c, tot = np.zeros(nr + 1, dtype=RUN_DTYPE), np.zeros(nx + 1, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    assert emur.emu_kernel_compact_runs(r.ctypes.data, r.size, c.ctypes.data, c.size, counts.ctypes.data, dense_r.ctypes.data, nr, dense_c.ctypes.data, nx, tot.ctypes.data) == 0
    assert tot.tolist() == [nr, nx, 0, int(valid.sum())]
    one = np.zeros(ARENAS, dtype=COUNTS_DTYPE)
    one[0]["runs"], one[0]["codes"] = nr, nx  # a dense array = one arena
    pad_r = np.zeros((ARENAS, nr + 1), dtype=RUN_DTYPE)
    pad_c = np.zeros((ARENAS, nx + 1), dtype=np.uint64)
    pad_r[0], pad_c[0] = dense_r, dense_c
    assert np.array_equal(expand_runs_numpy(pad_r, pad_c, one, n_out, L.INVALID), exp)
    # the fused payload pass: table[code], `missing` for what is not a slot
    nk = h.get_kmer_count()
    img = h.image()
    seq_off, out_off, raw, base, nflat = batch_layout(batch, k, 3)
    rng = np.random.default_rng(k)
    for el, dt in enumerate((np.uint8, np.uint16, np.uint32, np.uint64)):
        table = rng.integers(0, np.iinfo(dt).max, nk, dtype=dt, endpoint=True)
        missing = int(np.iinfo(dt).max) - 2
        out = np.full(n_out + 8, 0x5A, dtype=dt)
        rc = emur.emu_kernel_query_records_gather(img.ctypes.data, img.size, raw.ctypes.data + base, nflat, seq_off.ctypes.data, out_off.ctypes.data, len(batch),
                                                  table.ctypes.data, nk, el, missing, out.ctypes.data)
        assert rc == 0 and np.all(out[n_out:] == dt(0x5A))
        want = np.where(exp < np.uint64(nk), table[np.minimum(exp, np.uint64(nk - 1)).astype(np.int64)], dt(missing))
        assert np.array_equal(out[:n_out], want), el


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (63, 20, 128, 0), (21, 8, 64, 1)])
def test_runs_on_repeat_rich_sets_on_the_emulator(emur, tmp_path, k, m, bits, kind):
    """Index sets with the shape of genomes: 5-70 % of the k-mers behind colliding minimizers -> explicit runs and the explicit-code arenas carry real work."""
    from util import genomic_like_records

    recs = genomic_like_records(1000 + k + m, 30, k, 5000)
    path = str(tmp_path / "t.lph")
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=4)
    o = orc.Index(path, kind, bits)
    assert o.fallback_keys > 0.03 * o.nkmers
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits)
    noisy = bytearray(recs[1])
    noisy[len(noisy) // 3:len(noisy) // 3 + k + 4] = b"N" * (k + 4)
    batch = recs[:12] + [bytes(noisy), recs[0][::-1], b"AC" * 400, recs[2][:k], b""]
    exp, valid = streaming_expectation(o, batch, k, with_valid=True)
    runs, codes, counts, out_off = emu_runs(emur, h, batch, k, 5)
    assert int(counts["codes"].sum()) > 0.03 * len(exp)
    kinds = check_runs_against_codes(runs, codes, counts, exp, L.INVALID, valid)
    assert kinds[2] > 0 and kinds[0] > 0 and (kind == 1 or kinds[1] > 0)  # (the unpartitioned function only counts up: unpartitioned_mphf.hpp:130-137)
    got, st = emu_expand(emur, runs, codes, counts, len(exp))
    assert st == 0 and np.array_equal(got, exp)


def test_runs_capacity_overflow_is_reported_on_the_emulator(emur, tmp_path):
    """Arenas too small for the batch: the counters end above the capacity, nothing is written beyond it, and the expander says so."""
    k, m = 31, 20
    recs = dedup_records(random_records(77, 6, 1500, 2500), k)
    path = str(tmp_path / "t.lph")
    L.build(recs, k, m, path, kmer_bits=64, threads=4)
    h = L.Mphf.load(path, kmer_bits=64)
    runs, codes, counts, out_off = emu_runs(emur, h, recs, k, 0, cap_runs=ARENAS * 2, cap_codes=ARENAS)
    assert int(counts["runs"].max()) > 2
    _, st = emu_expand(emur, runs, codes, counts, int(out_off[-1]))
    assert st == 1


def forged_runs():
    """Runs no kernel of the library writes (lphq_expand_runs takes its runs from the caller): an explicit run whose code index + length wraps 2^64, one whose index lies beyond the
    arena, a run beyond the output, a run of kind 3 -- and one good run.  -> (runs[ARENAS, 8], codes[ARENAS, 8], counts, n_out, expected output)."""
    runs = np.zeros((ARENAS, 8), dtype=RUN_DTYPE)
    codes = np.arange(ARENAS * 8, dtype=np.uint64).reshape(ARENAS, 8) + np.uint64(1000)
    counts = np.zeros(ARENAS, dtype=COUNTS_DTYPE)
    n_out = 64
    where = lambda first, length, kind: np.uint64(first | (length << 40) | (kind << 62))
    runs[0][0] = (np.uint64(0xFFFFFFFFFFFFFFFF), where(0, 2, 2))   # explicit, code index 2^64 - 1: index + length wraps to 1
    runs[0][1] = (np.uint64(9), where(4, 3, 2))                    # explicit, index beyond the arena's 8 codes
    runs[0][2] = (np.uint64(7), where(60, 10, 0))                  # up-run that leaves the output
    runs[0][3] = (np.uint64(7), where(8, 2, 3))                    # no such kind
    runs[0][4] = (np.uint64(500), where(20, 5, 0))                 # good: windows 20..24 = 500..504
    runs[0][5] = (np.uint64(6), where(30, 2, 2))                   # good: explicit codes 6, 7 of arena 0
    counts[0]["runs"], counts[0]["codes"] = 6, 8
    want = np.full(n_out, L.INVALID, dtype=np.uint64)
    want[20:25] = np.arange(500, 505, dtype=np.uint64)
    want[30:32] = codes[0][6:8]
    return runs, codes, counts, n_out, want


def test_forged_runs_are_skipped_and_reported_on_the_emulator(emur):
    runs, codes, counts, n_out, want = forged_runs()
    got, st = emu_expand(emur, runs, codes, counts, n_out)
    assert st == 1 and np.array_equal(got, want)


@pytest.mark.parametrize("mode,seed", [(1, 0), (2, 424242)])
def test_sinks_under_other_thread_schedules(emur, tmp_path, mode, seed):
    """A missing barrier in the new sinks becomes a mismatch when the threads of a workgroup run in reverse / reshuffled order between synchronisation points."""
    emur.emu_set_schedule.restype = None
    emur.emu_set_schedule.argtypes = [C.c_int, C.c_uint64]
    emur.emu_set_schedule(mode, seed)
    try:
        test_runs_and_gather_on_the_emulator(emur, tmp_path, 31, 20, 64, 0)
        test_runs_and_gather_on_the_emulator(emur, tmp_path, 9, 7, 64, 0)      # W = 3: > 256 leaders per tile, the slot-by-slot rounds
        test_runs_and_gather_on_the_emulator(emur, tmp_path, 63, 20, 128, 0)
        test_runs_on_repeat_rich_sets_on_the_emulator(emur, tmp_path, 21, 8, 64, 1)
    finally:
        emur.emu_set_schedule(0, 0)
