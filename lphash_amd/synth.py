"""Deterministic synthetic inputs (SplitMix64 -> 2 bits per base), shared by bench.py and the tests.

The reference ships no generator; SURVEY.md section 8(d) fixes this one: forward strand only
(the reference is non-canonical, src/partitioned_mphf.cpp:34), equal-length records.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(x):
    """Vectorised SplitMix64 finaliser over a uint64 array of counters."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_bases(seed, n, chunk_words=1 << 22):
    """n ASCII bases as a uint8 array; base i is bits (2*(i%32)) of splitmix64(seed * 2^40 + i//32)."""
    nw = (n + 31) // 32
    out = np.empty(nw * 32, dtype=np.uint8)
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    for a in range(0, nw, chunk_words):
        b = min(nw, a + chunk_words)
        with np.errstate(over="ignore"):
            ctr = np.arange(a, b, dtype=np.uint64) + (np.uint64(seed) << np.uint64(40))
        w = splitmix64(ctr)
        codes = ((w[:, None] >> shifts) & np.uint64(3)).astype(np.uint8).reshape(-1)
        out[a * 32:b * 32] = _ACGT[codes]
    return out[:n]


def synthetic_records(seed, n_records, length, views=True):
    """n_records records of `length` bases each, as one contiguous uint8 array + the list of per-record bytes (views=False: None instead of the
    list -- a second copy of the set that a caller who cuts records out on demand does not need)."""
    flat = random_bases(seed, n_records * length)
    return flat, ([flat[i * length:(i + 1) * length].tobytes() for i in range(n_records)] if views else None)


def diverged_copy_records(seed, n_records, length, k, views=True):
    """A synthetic set with the repeat structure of a genome, at any size: the second half of the records are diverged copies of the first half -- record
    n/2 + i is record i with one substitution in every stretch of k - 1 bases (phase and new base drawn per record and per site).  No k-mer is repeated (every window
    of a copy holds a substituted base), but an m-mer survives in a copy whenever it fits between two substitutions, so a large share of the minimizers occurs in
    two super-k-mers: those are the reference's colliding minimizers (src/partitioned_mphf.cpp:163-268), and their k-mers -- tens of per cent of the set instead of
    the 0.01 % of uniform bases -- are answered by the fallback function."""
    assert n_records >= 2 and k >= 3
    half = n_records // 2
    flat = random_bases(seed, n_records * length).copy()
    recs2d = flat.reshape(n_records, length)
    step = k - 1
    for i in range(n_records - half):
        src = recs2d[i % half]
        dst = recs2d[half + i]
        dst[:] = src
        w = splitmix64(np.arange(length // step + 2, dtype=np.uint64) + (np.uint64(seed) << np.uint64(40)) + (np.uint64(half + i) << np.uint64(20)) + np.uint64(1 << 39))
        phase = int(w[0] % np.uint64(step))
        pos = np.arange(phase, length, step)
        bump = (1 + (w[1:1 + len(pos)] % np.uint64(3))).astype(np.uint8)  # 1..3: never the same base
        code = ((dst[pos] >> 1) ^ (dst[pos] >> 2)) & 3  # A C G T -> 0 1 2 3 (seq_nt4_table)
        dst[pos] = _ACGT[(code + bump) & 3]
    return flat, ([recs2d[i].tobytes() for i in range(n_records)] if views else None)
