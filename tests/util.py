"""Shared helpers for the tests: seeded synthetic records and k-mer packing (numpy, test side only)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
CODE = np.full(256, 0, dtype=np.uint8)  # string_to_integer_no_reverse: table value & 3 (non-ACGT -> 0)
for ch, v in ((b"A", 0), (b"a", 0), (b"C", 1), (b"c", 1), (b"G", 2), (b"g", 2), (b"T", 3), (b"t", 3), (b"U", 3), (b"u", 3)):
    CODE[ch[0]] = v


def random_records(seed, n, lmin, lmax):
    rng = np.random.default_rng(seed)
    return [bytes(ACGT[rng.integers(0, 4, int(rng.integers(lmin, lmax + 1)))]) for _ in range(n)]


def pack_kmers(seq: bytes, positions, k, kmer_bits):
    """kmer_t values (first base in the most significant used bits) of the windows at `positions`."""
    codes = CODE[np.frombuffer(seq, dtype=np.uint8)]
    out = np.zeros((len(positions), 2), dtype=np.uint64)
    for i, p in enumerate(positions):
        v = 0
        for c in codes[p:p + k]:
            v = (v << 2) | int(c)
        out[i, 0] = v & 0xFFFFFFFFFFFFFFFF
        out[i, 1] = v >> 64
    return out[:, 0].copy() if kmer_bits == 64 else out.reshape(-1).copy()


def dedup_records(recs, k):
    """Keep only records that do not repeat a k-mer (inside themselves or of an earlier kept record): the index
    format requires duplicate-free k-mer sets (README.md:112-116 of the reference)."""
    seen, keep = set(), []
    for r in recs:
        ks = [r[i:i + k] for i in range(len(r) - k + 1)]
        if len(set(ks)) == len(ks) and not (set(ks) & seen):
            seen |= set(ks)
            keep.append(r)
    return keep
