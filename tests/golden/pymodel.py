#!/usr/bin/env python3
"""A second, independently structured restatement of the LPHash build + query path -- TEST INFRASTRUCTURE ONLY.

Pure Python over plain lists and dictionaries: no ring buffers, no streaming state machines, no shared code with
oracle/lph_oracle.c.  LPHash-owned semantics are taken from the reference sources (paths relative to the reference tree):
  super-k-mers            include/minimizer.hpp:11-170   (a run of consecutive k-mers whose leftmost-minimum m-mer is the same occurrence)
  classify                src/minimizer.cpp:5-50         (minimizers seen in more than one super-k-mer -> size 0, their k-mers -> fallback)
  inverted index          src/partitioned_mphf.cpp:170-267, src/unpartitioned_mphf.cpp:85-170
  wavelet tree, rank9     src/quartet_wtree.cpp:12-54, include/rs_bit_vector.hpp:120-157
  Elias-Fano              include/ef_sequence.hpp:37-75
  query                   src/partitioned_mphf.cpp:292-339, src/unpartitioned_mphf.cpp:191-206, include/mphf_utils.hpp:118-137
  serialization           the visit() orders of include/partitioned_mphf.hpp:204-219, include/unpartitioned_mphf.hpp:199-211
PTHash v1 (github.com/jermp/pthash; absent from the reference tree) is restated from SURVEY.md Appendix A, like the C oracle -- so the
two agree on PTHash only if both transcribed the same recall correctly; what this model adds is a check of every transcription.

    python tests/golden/pymodel.py        regenerates every tests/golden/*.lph and compares bytes and query codes with the committed files
tests/test_oracle_cpu.py::test_python_model_regenerates_the_golden_files runs the same comparison.
"""
import math
import os
import sys

import numpy as np

M64 = (1 << 64) - 1
M128 = (1 << 128) - 1
MUR = 0xC6A4A7935BD1E995


def murmur8(v, seed):
    """MurmurHash2_64 (MurmurHash64A) of the 8 little-endian bytes of v."""
    h = (seed ^ ((8 * MUR) & M64)) & M64
    k = (v * MUR) & M64
    k ^= k >> 47
    k = (k * MUR) & M64
    h ^= k
    h = (h * MUR) & M64
    h ^= h >> 47
    h = (h * MUR) & M64
    h ^= h >> 47
    return h


NT = {ord(c): v for c, v in zip("AaCcGgTtUu", (0, 0, 1, 1, 2, 2, 3, 3, 3, 3))}


# ------------------------------------------------------------------------------------------- serialized containers
class Out:
    def __init__(self):
        self.b = bytearray()

    def u8(self, v):
        self.b += int(v).to_bytes(1, "little")

    def u64(self, v):
        self.b += int(v & M64).to_bytes(8, "little")

    def u128(self, v):
        self.b += int(v & M128).to_bytes(16, "little")

    def vec(self, values, width):
        self.u64(len(values))
        for v in values:
            self.b += int(v & ((1 << (8 * width)) - 1)).to_bytes(width, "little")


def words_of_bits(nbits, ones):
    w = [0] * ((nbits + 63) // 64)
    for p in ones:
        w[p >> 6] |= 1 << (p & 63)
    return w


def put_bit_vector(o, nbits, ones):
    o.u64(nbits)
    o.vec(words_of_bits(nbits, ones), 8)


def put_default_compact_vector(o):
    """a default-constructed pthash::compact_vector (never built): all-zero scalars, no words"""
    for _ in range(3):
        o.u64(0)
    o.vec([], 8)


def put_compact_vector(o, values, width):
    """pthash::compact_vector built from n values of `width` bits, little-endian bit order, one padding word."""
    n = len(values)
    big = 0
    if width:
        for i, v in enumerate(values):
            big |= v << (i * width)
    nwords = (n * width + 63) // 64 + 1
    o.u64(n)
    o.u64(width)
    o.u64(M64 if width == 64 else (1 << width) - 1)
    o.vec([(big >> (64 * i)) & M64 for i in range(nwords)], 8)


def put_compact_vector_auto(o, values):
    mx = max(values) if values else 0
    put_compact_vector(o, values, max(1, mx.bit_length()))


def put_darray1(o, ones):
    """pthash::darray1 over the positions of the set bits: blocks of 1024 ones, sub-blocks of 32."""
    blocks, subs, overflow = [], [], []
    for a in range(0, len(ones), 1024):
        cur = ones[a:a + 1024]
        if cur[-1] - cur[0] < (1 << 16):
            blocks.append(cur[0])
            subs += [cur[i] - cur[0] for i in range(0, len(cur), 32)]
        else:
            blocks.append(-len(overflow) - 1)
            overflow += cur
            subs += [0xFFFF] * len(range(0, len(cur), 32))
    o.u64(len(ones))
    o.vec([b & M64 for b in blocks], 8)
    o.vec(subs, 2)
    o.vec(overflow, 8)


def put_elias_fano(o, values, universe, leading_zero):
    """values non-decreasing.  leading_zero: lphash::ef_sequence::encode (a 0 is stored first); else pthash::ef_sequence<false>."""
    n = len(values)
    if n == 0:  # encode() returns at once (ef_sequence.hpp:39): default-constructed members
        put_bit_vector(o, 0, [])
        put_darray1(o, [])
        put_default_compact_vector(o)
        return
    stored = ([0] if leading_zero else []) + list(values)
    nn = len(stored)
    q = universe // nn
    low_bits = (q.bit_length() - 1) if q else 0
    ones = [(v >> low_bits) + i for i, v in enumerate(stored)]
    put_bit_vector(o, nn + (universe >> low_bits) + 1, ones)
    put_darray1(o, ones)
    put_compact_vector(o, [v & ((1 << low_bits) - 1) for v in stored], low_bits)


def put_rs_bit_vector(o, bits):
    """lphash::rs_bit_vector::build(bits, false): the bits, one (rank before, 7 x 9-bit in-block prefix counts) pair per 512-bit block,
    a closing (total, 0) pair, no select hints."""
    words = words_of_bits(len(bits), [i for i, b in enumerate(bits) if b])
    put_bit_vector(o, len(bits), [i for i, b in enumerate(bits) if b])
    pairs, before = [], 0
    for b in range(0, len(words), 8):
        blk = words[b:b + 8] + [0] * 8
        inside, packed = 0, 0
        for j in range(1, 8):
            inside += bin(blk[j - 1]).count("1")
            packed |= inside << ((7 - j) * 9)
        pairs += [before, packed]
        before += sum(bin(w).count("1") for w in words[b:b + 8])
    pairs += [before, 0]
    o.vec(pairs, 8)
    o.vec([], 8)


# ------------------------------------------------------------------------------------------- PTHash v1
class PtHash:
    T = 0x9999999999999800  # uint64_t(0.6 * UINT64_MAX)

    def __init__(self, hashes, seed, c=3.0, alpha=0.94):
        self.seed, self.n = seed, len(hashes)
        self.table_size = self.num_dense = self.num_sparse = 0
        self.pilots, self.free = [], []
        n = self.n
        if n == 0:
            return
        ts = int(n / alpha)
        if ts & (ts - 1) == 0:
            ts += 1
        nb = 6 if n < 2 else math.ceil(c * n / math.log2(n))
        self.table_size, self.num_buckets = ts, nb
        self.num_dense = int(0.3 * nb)
        self.num_sparse = nb - self.num_dense
        buckets = {}
        for h in hashes:
            buckets.setdefault(self.bucket(h), []).append(h)
        for keys in buckets.values():
            assert len(set(keys)) == len(keys), "duplicate hash in a bucket"
        taken = set()
        self.pilots = [0] * nb
        for b in sorted(buckets, key=lambda b: (-len(buckets[b]), b)):  # non-increasing size, then bucket id
            pilot = 0
            while True:
                hp = murmur8(pilot, seed)
                pos = [(h ^ hp) % ts for h in buckets[b]]
                if len(set(pos)) == len(pos) and not any(p in taken for p in pos):
                    break
                pilot += 1
                assert pilot < (1 << 28)
            taken.update(pos)
            self.pilots[b] = pilot
        # free slots (minimal output): table positions >= n are remapped to the unused positions < n
        self.free = self._fill_free_slots(taken)

    def _fill_free_slots(self, taken):
        """pthash fill_free_slots: entry p - n holds the unused slot < n that the key at table position p >= n moves to; entries of
        unoccupied positions repeat a neighbouring value so that the sequence is non-decreasing (it is Elias-Fano coded)."""
        n, ts = self.n, self.table_size
        if n >= ts:
            return []
        out, next_used, last_free, last_valid = [], n, 0, 0
        while True:
            while last_free < n and last_free in taken:
                last_free += 1
            if last_free == n:
                break
            while next_used not in taken:
                out.append(last_free)
                next_used += 1
            out.append(last_free)
            last_valid = last_free
            next_used += 1
            last_free += 1
        while next_used != ts:
            out.append(last_valid)
            next_used += 1
        return out

    def bucket(self, h):
        return h % self.num_dense if h < self.T else self.num_dense + h % self.num_sparse

    def position(self, h):
        p = (h ^ murmur8(self.pilots[self.bucket(h)], self.seed)) % self.table_size
        return p if p < self.n else self.free[p - self.n]

    @staticmethod
    def _put_dictionary(o, values):
        freq = {}
        for v in values:
            freq[v] = freq.get(v, 0) + 1
        dic = sorted(freq, key=lambda v: (-freq[v], v))  # by non-increasing frequency (ties: increasing value)
        code = {v: i for i, v in enumerate(dic)}
        put_compact_vector_auto(o, [code[v] for v in values])
        put_compact_vector_auto(o, dic)

    def put(self, o):
        def M(d):
            return (M128 // d + 1) & M128 if d else 0

        o.u64(self.seed)
        o.u64(self.n)
        o.u64(self.table_size)
        o.u128(M(self.table_size))
        o.u64(self.num_dense)
        o.u64(self.num_sparse)
        o.u128(M(self.num_dense))
        o.u128(M(self.num_sparse))
        if self.n == 0:  # an all-zero function (the reference never builds one): four empty compact vectors, empty Elias-Fano
            for _ in range(4):
                put_default_compact_vector(o)
            put_elias_fano(o, [], 0, False)
            return
        front = int(self.num_buckets * 0.3)
        self._put_dictionary(o, self.pilots[:front])
        self._put_dictionary(o, self.pilots[front:])
        put_elias_fano(o, self.free, self.free[-1] if self.free else 0, False)


# ------------------------------------------------------------------------------------------- LPHash
LEFT, RIGHT, MAXIMAL, NONE = 0, 1, 2, 3


def segments(record):
    """maximal runs of valid bases of a record, as (list of 2-bit codes, closed): closed = a non-ACGT byte ends the run (False: the record's end does)"""
    cur = []
    for ch in record:
        if ch in NT:
            cur.append(NT[ch])
        else:
            if cur:
                yield cur, True
            cur = []
    if cur:
        yield cur, False


def windows_of(codes, k, m, mm_seed):
    """for every k-mer of a run of valid bases: (k-mer value, minimizer value, offset of the minimizer in the k-mer, its absolute position)"""
    W = k - m + 1
    mmers = [0] * (len(codes) - m + 1)
    v = 0
    for i, c in enumerate(codes):
        v = ((v << 2) | c) & ((1 << (2 * m)) - 1)
        if i >= m - 1:
            mmers[i - m + 1] = v
    hs = [murmur8(x, mm_seed) for x in mmers]
    kv = 0
    for i, c in enumerate(codes):
        kv = ((kv << 2) | c) & ((1 << (2 * k)) - 1)
        if i >= k - 1:
            w = i - k + 1
            best = min(range(w, w + W), key=lambda j: (hs[j], j))  # leftmost minimum
            yield kv, mmers[best], best - w, best


class Model:
    def __init__(self, records, k, m, kind, kmer_bits, mm_seed=42, c=3.0):
        self.k, self.m, self.kind, self.kmer_bits, self.mm_seed, self.W = k, m, kind, kmer_bits, mm_seed, k - m + 1
        supers = []  # (minimizer, p1 of the first k-mer, [k-mers]) in input order
        self.nkmers = 0
        for rec in records:
            for codes, closed in segments(rec):
                if len(codes) < k:
                    continue
                if len(codes) == k and closed:
                    # A run of EXACTLY k valid bases that a non-ACGT byte closes: minimizer::from_string counts its k-mer (include/minimizer.hpp:60) but looks
                    # for the window's minimum only when base k + 1 arrives (:61-66); the break comes first and pushes nothing (:139-152), and
                    # get_colliding_kmers (:287-301) skips it the same way.  The k-mer is in nkmers and in no super-k-mer.  (A run of k bases that ends WITH the
                    # record is kept: :154-163.)
                    self.nkmers += 1
                    continue
                prev = None
                for kv, mm, off, where in windows_of(codes, k, m, mm_seed):
                    if where != prev:
                        supers.append([mm, off, []])
                        prev = where
                    supers[-1][2].append(kv)
                    self.nkmers += 1
        seen = {}
        for mm, _, _ in supers:
            seen[mm] = seen.get(mm, 0) + 1
        self.fallback_keys = [kv for mm, _, kms in supers if seen[mm] > 1 for kv in kms]
        distinct = sorted(seen)
        self.D = len(distinct)
        self.mphf = PtHash([murmur8(x, 1) for x in distinct], 1, c)
        info = {mm: ((p1, len(kms)) if seen[mm] == 1 else (0, 0)) for mm, p1, kms in supers}
        self.by_order = [None] * self.D  # (p1, size) of the minimizer with order i; (0, 0) = colliding
        for mm in distinct:
            self.by_order[self.mphf.position(murmur8(mm, 1))] = info[mm]
        assert all(x is not None for x in self.by_order)
        self.fb = PtHash([self.fallback_hash(kv) for kv in self.fallback_keys], 1, c)
        self._index()

    def fallback_hash(self, kv, seed=1):
        if self.kmer_bits == 64:
            return murmur8(kv & M64, seed)
        return murmur8(kv & M64, seed) ^ murmur8(kv >> 64, (~seed) & M64)

    def _index(self):
        k, m, W = self.k, self.m, self.W
        if self.kind == 1:  # mphf_alt: two prefix-sum sequences in minimizer order
            self.positions = [p for p, _ in self.by_order]
            self.sizes = [s for _, s in self.by_order]
            self.main = sum(self.sizes)
            return
        self.types, self.left, self.right, self.none_s, self.none_p = [], [], [], [], []
        self.n_max = 0
        for p1, size in self.by_order:
            if size == 0:
                self.types.append(RIGHT)
                self.right.append(0)
            elif p1 == k - m:
                if size == W:
                    self.types.append(MAXIMAL)
                    self.n_max += 1
                else:
                    self.types.append(RIGHT)
                    self.right.append(size)
            elif p1 == size - 1:
                self.types.append(LEFT)
                self.left.append(p1 + 1)
            else:
                self.types.append(NONE)
                self.none_p.append(p1)
                self.none_s.append(size)
        self.seq = self.left + self.right + self.none_s + self.none_p
        self.prefix = [0]
        for v in self.seq:
            self.prefix.append(self.prefix[-1] + v)
        self.rcs = len(self.left)
        self.nss = self.rcs + len(self.right)
        self.nps = self.nss + len(self.none_s)

    # ---- essentials::save
    def serialize(self):
        o = Out()
        o.u8(self.k)
        o.u8(self.m)
        o.u64(self.mm_seed)
        o.u64(self.nkmers)
        o.u64(self.D)
        if self.kind == 0:
            o.u64(self.n_max)
            o.u64(self.rcs)
            o.u64(self.nss)
            o.u64(self.nps)
            self.mphf.put(o)
            put_rs_bit_vector(o, [t >> 1 for t in self.types])
            put_rs_bit_vector(o, [t & 1 for t in self.types if t >> 1 == 0])
            put_rs_bit_vector(o, [t & 1 for t in self.types if t >> 1 == 1])
            put_elias_fano(o, self.prefix[1:], self.prefix[-1], True)
        else:
            o.u64(self.main)
            self.mphf.put(o)
            for seq in (self.positions, self.sizes):
                acc, cum = 0, []
                for v in seq:
                    acc += v
                    cum.append(acc)
                put_elias_fano(o, cum, acc, True)
        self.fb.put(o)
        return bytes(o.b)

    # ---- MPHF::operator()(contig, length, streaming = false) for a record of valid bases (streaming == random there)
    def query(self, record):
        out = []
        codes = [NT[ch] for ch in record]
        if len(codes) < self.k:
            return out
        for kv, mm, off, _ in windows_of(codes, self.k, self.m, self.mm_seed):
            out.append(self.lookup(kv, mm, off) & M64)  # `position` = offset of the minimizer from the left end of the k-mer (mphf_utils.hpp:136)
        return out

    def lookup(self, kv, mm, p):
        h = self.mphf.position(murmur8(mm, 1)) if self.D else 0
        k, m, W = self.k, self.m, self.W
        if self.kind == 1:
            before, size = sum(self.sizes[:h]), self.sizes[h]
            if size == 0:
                return self.main + self.fb.position(self.fallback_hash(kv))
            return before + self.positions[h] - p
        t = self.types[h]
        r = sum(1 for x in self.types[:h] if x == t)
        shift = W * self.n_max
        if t == MAXIMAL:
            return W * r + p
        if t == LEFT:
            return self.prefix[r] + shift + p
        if t == RIGHT:
            a, b = self.prefix[self.rcs + r], self.prefix[self.rcs + r + 1]
            if a == b:
                return self.prefix[self.nps] + shift + self.fb.position(self.fallback_hash(kv))
            return a + shift + (k - m - p)
        a = self.prefix[self.nss + r]
        p1 = self.prefix[self.nps + r + 1] - self.prefix[self.nps + r]
        return a + shift + (p1 - p)


def check_against_golden(golden_dir, verbose=False):
    """Rebuild every committed fixture from its records with this model: file bytes and query codes must equal the committed ones."""
    names = sorted(f[:-4] for f in os.listdir(golden_dir) if f.endswith(".npz"))
    for name in names:
        z = np.load(os.path.join(golden_dir, name + ".npz"))
        recs = [bytes(z["records"][z["rec_off"][i]:z["rec_off"][i + 1]]) for i in range(len(z["rec_off"]) - 1)]
        mdl = Model(recs, int(z["k"]), int(z["m"]), int(z["kind"]), int(z["kmer_bits"]))
        blob = mdl.serialize()
        want = open(os.path.join(golden_dir, name + ".lph"), "rb").read()
        assert blob == want, f"{name}: serialized model differs from the committed .lph ({len(blob)} vs {len(want)} bytes)"
        assert (mdl.nkmers, mdl.D, len(mdl.fallback_keys)) == (int(z["nkmers"]), int(z["distinct_minimizers"]), int(z["fallback_keys"]))
        for i in range(len(z["q_off"]) - 1):
            q = bytes(z["queries"][z["q_off"][i]:z["q_off"][i + 1]])
            got = np.array(mdl.query(q), dtype=np.uint64)
            for key, off in (("random", "r_off"), ("streaming", "s_off")):
                exp = z[key][z[off][i]:z[off][i + 1]]
                assert np.array_equal(got, exp), f"{name}: query {i} ({key}) differs"
        if verbose:
            print(name, "ok:", len(blob), "bytes,", mdl.nkmers, "k-mers,", mdl.D, "minimizers,", len(mdl.fallback_keys), "fallback k-mers")
    return len(names)


if __name__ == "__main__":
    n = check_against_golden(os.path.dirname(os.path.abspath(__file__)), verbose=True)
    print(n, "fixtures reproduced byte for byte")
    sys.exit(0)
