"""Test-side view of the run-encoded streaming result (include/lphq.h: lphq_run, lphq_run_counts): numpy dtypes, a plain expander and the
invariants a batch of runs must satisfy.  Shared by the emulator tests (test_runs_cpu.py) and the GPU tests (test_gpu_runs.py)."""
import numpy as np

ARENAS = 32  # LPHQ_RUN_ARENAS
RUN_DTYPE = np.dtype([("code", "<u8"), ("where", "<u8")])  # lphq_run
COUNTS_DTYPE = np.dtype([("runs", "<u8"), ("codes", "<u8"), ("reserved", "<u8", (14,))])  # lphq_run_counts: 128 B per arena
UP, DOWN, EXPLICIT = 0, 1, 2
assert RUN_DTYPE.itemsize == 16 and COUNTS_DTYPE.itemsize == 128


def run_fields(runs):
    w = runs["where"]
    return (w & np.uint64(0xFFFFFFFFFF)).astype(np.int64), ((w >> np.uint64(40)) & np.uint64(0x3FFFFF)).astype(np.int64), (w >> np.uint64(62)).astype(np.int64)


def batch_layout(recs, k, misalign=0):
    """Offsets and a padded, (mis)aligned copy of the concatenated records, as the emulator harness wants them -> (seq_off, out_off, raw, base, n_bases)."""
    lens = np.array([len(r) for r in recs], dtype=np.uint64)
    seq_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    nwin = np.where(lens >= k, lens - np.uint64(k) + np.uint64(1), np.uint64(0)).astype(np.uint64)
    out_off = np.concatenate([[0], np.cumsum(nwin)]).astype(np.uint64)
    flat = np.frombuffer(b"".join(recs), dtype=np.uint8)
    raw = np.full(len(flat) + 256, ord("#"), dtype=np.uint8)
    base = (-raw.ctypes.data) % 16 + 64 + misalign
    raw[base:base + len(flat)] = flat
    return seq_off, out_off, raw, base, len(flat)


def expand_runs_numpy(runs, codes, counts, n_out, invalid):
    """out[first + i] = code + i (UP) / code - i (DOWN) / codes[arena][code + i] (EXPLICIT), wrapping uint64; windows in no run = `invalid`."""
    out = np.full(n_out, invalid, dtype=np.uint64)
    for a in range(ARENAS):
        n = int(counts[a]["runs"])
        r = runs[a][:n]
        first, length, kind = run_fields(r)
        for f, ln, kd, c in zip(first.tolist(), length.tolist(), kind.tolist(), r["code"].tolist()):
            i = np.arange(ln, dtype=np.uint64)
            if kd == EXPLICIT:
                out[f:f + ln] = codes[a][c:c + ln]
            elif kd == UP:
                out[f:f + ln] = np.uint64(c) + i
            else:
                out[f:f + ln] = np.uint64(c) - i
    return out


def check_runs_against_codes(runs, codes, counts, exp, invalid, valid=None):
    """The contract of lphq_query_records_runs against the expanded codes `exp` of the same batch (LPHQ_INVALID at windows that hold a non-ACGT base):
    counters inside the capacities, every valid window in exactly one run, no run over an invalid window, lengths >= 1, kinds in {UP, DOWN, EXPLICIT},
    explicit runs inside their arena's code array.  valid: bool per window, from the BASES (a non-member k-mer's wrapped arithmetic may legitimately yield
    2^64 - 1, which a comparison with LPHQ_INVALID would take for a non-ACGT window); default: exp != invalid.  Returns the number of runs per kind."""
    n_out = len(exp)
    cover = np.zeros(n_out + 1, dtype=np.int64)
    kinds = [0, 0, 0]
    for a in range(ARENAS):
        n, nx = int(counts[a]["runs"]), int(counts[a]["codes"])
        assert n <= runs.shape[1] and nx <= codes.shape[1], "arena overflow"
        first, length, kind = run_fields(runs[a][:n])
        assert np.all(length >= 1) and np.all(kind <= 2) and np.all(first + length <= n_out)
        ex = kind == EXPLICIT
        assert np.all(runs[a][:n]["code"][ex].astype(np.int64) + length[ex] <= nx)
        assert int(length[ex].sum()) == nx, "explicit codes and explicit runs of an arena must add up"
        np.add.at(cover, first, 1)
        np.add.at(cover, first + length, -1)
        for kd in range(3):
            kinds[kd] += int((kind == kd).sum())
    cover = np.cumsum(cover)[:n_out]
    if valid is None:
        valid = exp != np.uint64(invalid)
    assert np.array_equal(cover, valid.astype(np.int64)), "every window without a non-ACGT base belongs to exactly one run, the others to none"
    return kinds
