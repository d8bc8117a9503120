"""CPU tests of the oracle (oracle/lph_oracle.c): golden fixtures, the reference's own --check properties
(bijection, range, streaming == random; build.cpp:53-70, mphf_utils.hpp:51-100) and known-answer checks of the
pieces that do not depend on PTHash."""
import glob
import sys
import os

import numpy as np
import pytest

import orc
from util import random_records

M64 = (1 << 64) - 1


def murmur64a_py(data: bytes, seed: int) -> int:
    """Independent restatement of the public MurmurHash64A (Appleby), used to cross-check the C one."""
    m, r = 0xC6A4A7935BD1E995, 47
    h = (seed ^ (len(data) * m)) & M64
    nblocks = len(data) // 8
    for i in range(nblocks):
        k = int.from_bytes(data[8 * i:8 * i + 8], "little")
        k = (k * m) & M64
        k ^= k >> r
        k = (k * m) & M64
        h ^= k
        h = (h * m) & M64
    tail = data[8 * nblocks:]
    if tail:
        h ^= int.from_bytes(tail, "little")
        h = (h * m) & M64
    h ^= h >> r
    h = (h * m) & M64
    h ^= h >> r
    return h


def test_murmur64a_known_answers_and_cross_check():
    assert orc.lib.orc_murmur64a(b"", 0, 0) == 0  # h = 0 ^ (0*m) -> stays 0 through the finaliser
    rng = np.random.default_rng(5)
    for n in list(range(0, 24)) + [31, 64, 100]:
        for _ in range(5):
            data = bytes(rng.integers(0, 256, n, dtype=np.uint8))
            seed = int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2))
            assert orc.lib.orc_murmur64a(data, n, seed) == murmur64a_py(data, seed)


def golden_cases():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(here, "*.lph")))


def split(flat, off):
    return [flat[off[i]:off[i + 1]] for i in range(len(off) - 1)]


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_matches_golden(name, golden_dir):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    ix = orc.Index(os.path.join(golden_dir, name + ".lph"), int(z["kind"]), int(z["kmer_bits"]))
    assert (ix.k, ix.m, ix.nkmers, ix.distinct_minimizers, ix.fallback_keys) == (
        int(z["k"]), int(z["m"]), int(z["nkmers"]), int(z["distinct_minimizers"]), int(z["fallback_keys"]))
    queries = [bytes(q) for q in split(z["queries"], z["q_off"])]
    for q, s, r in zip(queries, split(z["streaming"], z["s_off"]), split(z["random"], z["r_off"])):
        assert np.array_equal(ix(q, True), s)
        assert np.array_equal(ix(q, False), r)


def test_golden_set_is_not_empty():
    assert len(golden_cases()) >= 6


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (31, 20, 128, 0), (63, 20, 128, 0), (31, 20, 64, 1), (15, 9, 64, 0),
                                           (31, 31, 64, 0), (21, 8, 64, 0), (21, 8, 64, 1), (33, 5, 128, 0), (9, 1, 64, 1)])
def test_check_properties(tmp_path, k, m, bits, kind):
    """The reference's --check: random-mode codes are a bijection onto [0, n) and streaming == random."""
    recs = random_records(k * 97 + m, 12, k, 2500)
    kmers = set()
    keep = []
    for r in recs:  # the format requires duplicate-free k-mer sets: drop records that would add a duplicate
        ks = [r[i:i + k] for i in range(len(r) - k + 1)]
        if len(set(ks)) == len(ks) and not (set(ks) & kmers):
            kmers |= set(ks)
            keep.append(r)
    path = str(tmp_path / "t.lph")
    orc.build(keep, k, m, path, kind=kind, kmer_bits=bits)
    ix = orc.Index(path, kind, bits)
    assert ix.nkmers == len(kmers)
    codes = []
    for r in keep:
        a, b = ix(r, True), ix(r, False)
        assert len(a) == len(b) == len(r) - k + 1 and np.array_equal(a, b)
        codes.append(b)
    codes = np.concatenate(codes)
    assert len(np.unique(codes)) == len(codes) and int(codes.max()) == ix.nkmers - 1


def test_short_and_empty_records(tmp_path):
    recs = random_records(3, 4, 100, 300)
    path = str(tmp_path / "t.lph")
    orc.build(recs, 31, 20, path)
    ix = orc.Index(path, 0, 64)
    for s in (True, False):
        assert len(ix(b"", s)) == 0
        assert len(ix(recs[0][:30], s)) == 0
        assert len(ix(recs[0][:31], s)) == 1


def test_query_packed_equals_contig_random(tmp_path):
    from util import pack_kmers

    for k, m, bits in ((31, 20, 64), (47, 17, 128)):
        recs = random_records(k, 3, 200, 600)
        path = str(tmp_path / f"t{k}.lph")
        orc.build(recs, k, m, path, kmer_bits=bits)
        ix = orc.Index(path, 0, bits)
        r = recs[1]
        pk = pack_kmers(r, range(len(r) - k + 1), k, bits)
        assert np.array_equal(ix.query_packed(pk), ix(r, False))


def test_load_rejects_garbage(tmp_path, golden_dir):
    data = open(os.path.join(golden_dir, "p_k31_m20_b64.lph"), "rb").read()
    p = tmp_path / "trunc.lph"
    p.write_bytes(data[:-7])
    with pytest.raises(RuntimeError):
        orc.Index(str(p), 0, 64)
    with pytest.raises(RuntimeError):
        orc.Index(os.path.join(golden_dir, "p_k31_m20_b64.lph"), 1, 64)  # wrong kind


def test_reference_driver_restatement(tmp_path):
    """orc_query_file = the reference's query driver (src/query.cpp:24-88): both passes over a .fa.gz / FASTQ, counts equal the sum of the
    per-record operator() sizes, timers are positive, a missing file gives the driver's exit code 2."""
    import gzip

    recs = random_records(77, 12, 20, 3000)
    path = str(tmp_path / "t.lph")
    orc.build(recs, 31, 20, path)
    ix = orc.Index(path, 0, 64)
    fa = tmp_path / "q.fa.gz"
    with gzip.open(fa, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">r%d some comment\n" % i)
            for j in range(0, len(r), 61):
                f.write(r[j:j + 61] + b"\n")
    fq = tmp_path / "q.fastq"
    with open(fq, "wb") as f:
        for i, r in enumerate(recs[:5]):
            f.write(b"@q%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))
    for p, rr in ((fa, recs), (fq, recs[:5])):
        n_s, n_r, us_s, us_r = ix.query_file(str(p))
        want = sum(max(0, len(r) - 30) for r in rr)
        assert n_s == n_r == want and us_s > 0 and us_r > 0
        nrec, nbases, _ = orc.file_digest(str(p))
        assert nrec == len(rr) and nbases == sum(len(r) for r in rr)
    with pytest.raises(RuntimeError):
        ix.query_file(str(tmp_path / "missing.fa"))


def test_python_model_regenerates_the_golden_files(golden_dir):
    """tests/golden/pymodel.py -- a second restatement of build + query, pure Python over lists and dictionaries, written from the
    reference sources independently of oracle/lph_oracle.c -- rebuilds every committed fixture from its records: the serialized bytes equal
    the committed .lph files (which the C oracle wrote) and its query codes equal the committed ones.  Catches transcription errors on
    either side; it cannot pin the PTHash recall itself, which both share."""
    sys.path.insert(0, golden_dir)
    import pymodel

    assert pymodel.check_against_golden(golden_dir) == 7


@pytest.mark.parametrize("k,m,bits,kind,seed", [(33, 5, 128, 0, 1), (21, 8, 64, 1, 2), (15, 9, 64, 0, 3), (31, 31, 64, 0, 4), (63, 32, 128, 1, 5), (9, 7, 64, 0, 6)])
def test_python_model_equals_the_oracle_on_fresh_inputs(tmp_path, golden_dir, k, m, bits, kind, seed):
    """The same comparison on inputs that are not fixtures, incl. 128-bit fallback hashing (k=33, m=5: many colliding minimizers), the
    unpartitioned function, W = 1, and non-member queries with wrapping arithmetic."""
    sys.path.insert(0, golden_dir)
    import pymodel
    from util import dedup_records

    recs = dedup_records(random_records(500 + seed, 5 if k > 12 else 40, k, 1200 if k > 12 else k + 10), k)
    path = str(tmp_path / "o.lph")
    orc.build(recs, k, m, path, kind=kind, kmer_bits=bits)
    mdl = pymodel.Model(recs, k, m, kind, bits)
    assert mdl.serialize() == open(path, "rb").read()
    ix = orc.Index(path, kind, bits)
    assert (mdl.nkmers, mdl.D, len(mdl.fallback_keys)) == (ix.nkmers, ix.distinct_minimizers, ix.fallback_keys)
    for q in recs[:2] + random_records(seed, 2, k, 300):
        assert np.array_equal(np.array(mdl.query(q), dtype=np.uint64), ix(q, False))


def test_all_cores_entry_equals_the_single_thread_one(tmp_path):
    """orc_query_packed_threads (the all-cores CPU figure of bench.py): same codes as orc_query_packed, for any thread count, both k-mer widths."""
    from util import pack_kmers, random_records

    for k, bits in ((31, 64), (41, 128)):
        recs = random_records(5 + k, 6, 300, 2000)
        path = str(tmp_path / f"t{k}.lph")
        orc.build(recs, k, 13, path, kind=0, kmer_bits=bits)
        o = orc.Index(path, 0, bits)
        flat = b"".join(recs)
        rng = np.random.default_rng(k)
        pk = pack_kmers(flat, rng.integers(0, len(flat) - k, 5000), k, bits)
        want = o.query_packed(pk)
        for threads in (1, 3, 8, 10_000):
            assert np.array_equal(o.query_packed_threads(pk, threads), want)
