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


def low_complexity_index_records(k, seed):
    """Index records whose windows hold EQUAL m-mers (ties of the minimizer hash -> the leftmost-minimum rule decides): homopolymers, tandem repeats of
    period p as long as they may be without repeating a k-mer (k + p - 1 bases), short repeats inside random flanks, beside ordinary random records."""
    rng = np.random.default_rng(seed)
    recs = [bytes([c]) * k for c in b"ACGT"]  # one k-mer each: every m-mer of the window is the same
    for unit in (b"AC", b"AG", b"CT", b"ACG", b"AAT", b"ACGT", b"AACCG", b"AAAAAC", b"ACGTTGC", b"ACCGGTTA", b"AAAAAAAAAAG"):
        p = len(unit)
        recs.append((unit * (k // p + 3))[:k + p - 1])  # p distinct k-mers, every one a rotation of the other
    for unit in (b"A", b"T", b"CA", b"GT", b"TTG", b"ACAG"):
        for _ in range(3):
            p = len(unit)
            rep = (unit * k)[:int(rng.integers(k // 2, k + p - 1))]
            left, right = random_records(int(rng.integers(1 << 30)), 2, k, 3 * k)
            recs.append(left + rep + right)
    recs += random_records(seed + 1, 12, 300, 4000)
    return dedup_records(recs, k)


def low_complexity_queries(recs, k, seed):
    rng = np.random.default_rng(seed)
    big = b"".join(recs)
    qs = list(recs)
    qs += [b"A" * 700, b"C" * (k + 1), b"T" * 2048, (b"AC" * 600), (b"ACG" * 500)[:1400], (b"AAAAAAAAAAG" * 200), (b"ACGTTGC" * 300)]  # repeats far longer than a window (non-members)
    for _ in range(6):  # members interrupted by runs of N of k bases and more, and by single N
        a = int(rng.integers(0, len(big) - 3000))
        q = bytearray(big[a:a + 3000])
        s = int(rng.integers(100, 2000))
        run = k + int(rng.integers(0, 40))
        q[s:s + run] = b"N" * run
        q[int(rng.integers(0, 3000 - 1))] = ord("N")
        qs.append(bytes(q[:3000]))
    qs += [b"N" * (2 * k), b"A" * (k - 1) + b"N" + b"A" * k, (b"A" * k + b"N") * 20, b"G" * 300 + b"NN" + b"G" * 300]
    return qs


def genomic_like_records(seed, n, k, max_len):
    """Records with the SHAPE of genomic sequence instead of i.i.d. uniform bases: a GC content drawn per set, tandem repeats and homopolymer runs as long as they
    may be without repeating a k-mer, and diverged copies of earlier segments (one substitution in every window of k - 1 bases: no k-mer is repeated, but most
    minimizers are -- the colliding-minimizer / fallback path of src/partitioned_mphf.cpp:163-268 gets real work).  De-duplicated like every index set."""
    rng = np.random.default_rng(seed)
    gc = rng.uniform(0.3, 0.7)
    probs = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    recs = []
    for _ in range(n):
        ln = int(rng.integers(k, max(k + 1, max_len)))
        r = bytearray(ACGT[rng.choice(4, size=ln, p=probs)].tobytes())
        style = rng.integers(0, 4)
        if style == 1 and ln > 3 * k:  # a tandem repeat / homopolymer inside
            p = int(rng.integers(1, 7))
            unit = bytes(ACGT[rng.integers(0, 4, p)])
            rep = (unit * (k // p + 3))[:int(rng.integers(max(2, k // 2), k + p - 1))]
            at = int(rng.integers(0, ln - len(rep)))
            r[at:at + len(rep)] = rep
        elif style == 2 and recs:  # a diverged copy of a segment of an earlier record
            src = recs[int(rng.integers(0, len(recs)))]
            seg_len = min(len(src), ln)
            a = int(rng.integers(0, len(src) - seg_len + 1))
            seg = bytearray(src[a:a + seg_len])
            step = max(1, k - 1)
            for pos in range(int(rng.integers(0, step)), seg_len, step):
                seg[pos] = b"ACGT"[(b"ACGT".index(seg[pos]) + 1 + int(rng.integers(0, 3))) % 4]
            at = int(rng.integers(0, ln - seg_len + 1))
            r[at:at + seg_len] = seg
        recs.append(bytes(r))
    return dedup_records(recs, k)


def dirty_index_records(seed, k, m, n_records=6):
    """Index records full of non-ACGT breaks at the distances that matter: stretches of k - 1, k, k + 1, m - 1, m, m + 1, 2k, ... valid bases, closed by one to three
    N, at the start, in the middle and at the end of records.  A stretch of EXACTLY k bases closed by an N is the case the reference treats specially (its k-mer is
    counted but pushed into no super-k-mer, include/minimizer.hpp:60-66,139-152).  No k-mer is repeated among the valid windows (else None)."""
    rng = np.random.default_rng(seed)
    recs = []
    for _ in range(n_records):
        parts = []
        for _ in range(int(rng.integers(1, 7))):
            ln = int(rng.choice([k - 1, k, k, k + 1, k + 2, max(1, m - 1), m, m + 1, 2 * k, 5 * k, int(rng.integers(1, 300))]))
            parts.append(random_records(int(rng.integers(1 << 30)), 1, ln, ln)[0])
        rec = b"".join(p + b"N" * int(rng.integers(1, 4)) for p in parts)
        if rng.random() < 0.5:
            rec = rec.rstrip(b"N")
        if rng.random() < 0.3:
            rec = b"N" + rec
        recs.append(rec)
    seen = set()
    for r in recs:
        for i in range(len(r) - k + 1):
            w = r[i:i + k]
            if b"N" in w:
                continue
            if w in seen:
                return None
            seen.add(w)
    return recs if seen else None
