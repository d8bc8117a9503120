"""CPU regression of the device lookup chain: lphash_amd/csrc/device_query.cuh compiled for the host (tests/emu/, test
infrastructure only) over the product's flat image, against the oracle.  Covers the image re-packing (fused blocks, pair
blocks, hashed-pilot tables, free-slot table, exact 32-bit modular reduction) and minimizer_of / phf_position / uni_lookup /
pair_lookup / finish_hval bit for bit -- without a GPU.  Kernels (tiling, LDS phases) are covered by the -m gpu tests."""
import ctypes as C
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

import lphash_amd as L
import orc
from util import dedup_records, pack_kmers, random_records

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-j4", "-C", d], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.environ.get("LPHQ_EMU_LIB") or os.path.join(d, "libemu.so"))  # LPHQ_EMU_LIB: e.g. a -fsanitize=undefined build of tests/emu
    lib.emu_query_packed.restype = C.c_int
    lib.emu_query_packed.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


def emu_query(lib, h, packed, n):
    img = h.image()
    out = np.empty(n, dtype=np.uint64)
    packed = np.ascontiguousarray(packed, dtype=np.uint64)
    assert lib.emu_query_packed(img.ctypes.data, img.size, packed.ctypes.data, n, out.ctypes.data) == 0
    return out


def check_index(lib, path, kind, bits, k, queries):
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits)
    for q in queries:
        if len(q) < k:
            continue
        pos = range(len(q) - k + 1)
        pk = pack_kmers(q, pos, k, bits)
        assert np.array_equal(emu_query(lib, h, pk, len(pos)), o(q, False))


def golden_cases():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.lph")))


@pytest.mark.parametrize("name", golden_cases())
def test_device_chain_on_golden(emu, name, golden_dir):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    queries = [bytes(z["queries"][z["q_off"][i]:z["q_off"][i + 1]]) for i in range(len(z["q_off"]) - 1)]
    check_index(emu, os.path.join(golden_dir, name + ".lph"), int(z["kind"]), int(z["kmer_bits"]), int(z["k"]), queries)


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (31, 20, 128, 0), (63, 20, 128, 0), (63, 17, 128, 0), (31, 20, 64, 1), (15, 9, 64, 0), (31, 31, 64, 0),
                                           (21, 8, 64, 0), (21, 8, 64, 1), (47, 17, 128, 1), (33, 5, 128, 0), (63, 32, 128, 0), (35, 32, 128, 1), (20, 4, 64, 0)])
def test_device_chain_random_indexes(emu, tmp_path, k, m, bits, kind):
    """Members and non-members, every type of minimizer, free slots, collisions/fallback, both block geometries (P = 4, 6)."""
    recs = dedup_records(random_records(k * 77 + m, 20, k, 5000), k)
    path = str(tmp_path / "t.lph")
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=4)
    check_index(emu, path, kind, bits, k, recs + random_records(5, 5, k, 3000))


def test_device_chain_larger_index(emu, tmp_path):
    """2 M k-mers: many blocks, multi-word planes, the 6 % free-slot path and a non-trivial pilot dictionary."""
    from lphash_amd.synth import synthetic_records

    flat, recs = synthetic_records(seed=5, n_records=20, length=100_030)
    path = str(tmp_path / "m.lph")
    L.build(recs, 31, 20, path, kmer_bits=64, threads=8)
    o = orc.Index(path, 0, 64)
    h = L.Mphf.load(path, kmer_bits=64)
    rng = np.random.default_rng(1)
    codes = np.frombuffer(b"".join(recs), dtype=np.uint8)
    lut = np.zeros(256, dtype=np.uint64)
    for ch, v in ((b"A", 0), (b"C", 1), (b"G", 2), (b"T", 3)):
        lut[ch[0]] = v
    two = lut[codes]
    starts = (rng.integers(0, 20, 200_000) * 100_030 + rng.integers(0, 100_000, 200_000)).astype(np.int64)
    pk = np.zeros(len(starts), dtype=np.uint64)
    for j in range(31):
        pk = (pk << np.uint64(2)) | two[starts + j]
    got = emu_query(emu, h, pk, len(pk))
    assert np.array_equal(got, o.query_packed(pk))
    assert got.max() < h.get_kmer_count()


# ------------------------------------------------------------------------------------------------------------------
# the kernels themselves (kernels_device.cuh) under the wave64 fiber emulator
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def emuk(emu):
    emu.emu_kernel_query_packed.restype = C.c_int
    emu.emu_kernel_query_packed.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    emu.emu_kernel_query_records.restype = C.c_int
    emu.emu_kernel_query_records.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int]
    emu.emu_kernel_pack_kmers.restype = C.c_int
    emu.emu_kernel_pack_kmers.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p]
    emu.emu_kernel_count_valid.restype = C.c_int
    emu.emu_kernel_count_valid.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_ulonglong)]
    return emu


def emu_records(lib, h, recs, k, streaming, misalign=0, block=256):
    """lphq_query_records as kernels.hip launches it, on the emulator; returns one code array per record."""
    img = h.image()
    lens = np.array([len(r) for r in recs], dtype=np.uint64)
    seq_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    nwin = np.where(lens >= k, lens - np.uint64(k) + np.uint64(1), np.uint64(0)).astype(np.uint64)
    out_off = np.concatenate([[0], np.cumsum(nwin)]).astype(np.uint64)
    flat = np.frombuffer(b"".join(recs), dtype=np.uint8)
    raw = np.full(len(flat) + 256, ord("#"), dtype=np.uint8)  # margins: the tile loader reads whole aligned 16-byte chunks
    base = (-raw.ctypes.data) % 16 + 64 + misalign
    raw[base:base + len(flat)] = flat
    out = np.full(int(out_off[-1]) + 1, 0xDEAD, dtype=np.uint64)
    rc = lib.emu_kernel_query_records(img.ctypes.data, img.size, raw.ctypes.data + base, len(flat), seq_off.ctypes.data, out_off.ctypes.data, len(recs),
                                      out.ctypes.data, 1 if streaming else 0, block)
    assert rc == 0 and out[-1] == 0xDEAD
    return [out[int(out_off[i]):int(out_off[i + 1])] for i in range(len(recs))]


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (63, 20, 128, 0), (31, 20, 64, 1), (21, 8, 64, 0), (47, 17, 128, 1), (31, 31, 64, 0), (15, 9, 64, 0), (9, 7, 64, 0),
                                           (63, 32, 128, 0), (33, 5, 128, 0), (15, 8, 64, 0),  # (15, 8): W = 8, > 200 leaders per tile: the eight-per-lane kernel's further queue rounds
                                           (35, 20, 128, 0), (31, 15, 64, 0),  # W = 16, 17: the block-form variant (P == 6) with ~120 leaders for its queue of 128 -- slot-by-slot path
                                           (37, 20, 128, 0), (38, 20, 128, 1), (41, 20, 128, 0)])  # W = 18, 19, 22: with the cases above every (W - 8) % 8 instantiation of the wide variant
def test_tile_kernels_on_the_emulator(emuk, tmp_path, k, m, bits, kind):
    """k_query_tiles_stream / k_query_tiles_random over ragged multi-record batches: members, non-members, records shorter than
    k, empty records, tile-straddling records, every pointer alignment class, both workgroup sizes."""
    recs = dedup_records(random_records(k * 19 + m, 14 if k > 12 else 60, k, 2600 if k > 12 else k + 12), k)
    assert len(recs) >= 4
    path = str(tmp_path / "t.lph")
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=4)
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits)
    batch = [b"", recs[0], recs[1][: k - 1], b"", recs[2][:k]] + recs[3:] + random_records(3, 6, 1, 1500) + [b""]
    # > 64 records inside one tile (beyond the LDS record table of the streaming kernels), then ~20 records per tile
    batch += random_records(11, 150, 0, 5) + [recs[0][:k + 2]] + random_records(12, 40, k, k + 20) + [recs[1]]
    exp = [o(r, False) for r in batch]
    BLK, RAW = 0x10000, 0x20000  # window minima by block decomposition / in the direct form, whatever W (the launcher picks by W)
    S8 = 0x40000                 # eight positions per lane (stream8_device.cuh), whatever W (the launcher picks by W)
    runs = [(True, 0, 256), (True, 5, 256 | BLK), (True, 15, 128 | RAW), (False, 0, 256), (False, 9, 256), (True, 7, 256 | RAW), (True, 2, 128 | BLK),
            (True, 0, 128 | S8), (True, 11, 128 | S8), (True, 4, 256 | S8)]
    for streaming, mis, block in runs:
        got = emu_records(emuk, h, batch, k, streaming, mis, block)
        for g, e in zip(got, exp):
            assert np.array_equal(g, e), (streaming, mis, block)


def test_tile_kernels_non_acgt_on_the_emulator(emuk, tmp_path):
    k, m = 31, 20
    recs = random_records(42, 5, 400, 1800)
    path = str(tmp_path / "t.lph")
    L.build(recs, k, m, path, kmer_bits=64, threads=4)
    o = orc.Index(path, 0, 64)
    h = L.Mphf.load(path, kmer_bits=64)
    rng = np.random.default_rng(0)
    batch = []
    for r in recs:
        q = bytearray(r)
        for p in rng.integers(0, len(q), 4):
            q[p] = ord("N")
        q[0], q[-1], q[len(q) // 2] = ord("R"), ord("-"), ord("u")
        batch.append(bytes(q))
    got_s = emu_records(emuk, h, batch, k, True)
    got_r = emu_records(emuk, h, batch, k, False)
    got_p = emu_records(emuk, h, batch, k, True, 3)  # another pointer alignment
    got_8 = emu_records(emuk, h, batch, k, True, 6, 128 | 0x40000)  # the eight-positions-per-lane kernel
    for q, gs, gr, gp, g8 in zip(batch, got_s, got_r, got_p, got_8):
        exp = o(q, False)  # random mode reads a non-ACGT base as 'A' (mphf_utils.hpp:108)
        assert np.array_equal(gr, exp)
        bad = np.array([c not in b"ACGTUacgtu" for c in q])
        csum = np.concatenate([[0], np.cumsum(bad)])
        valid = (csum[k:] - csum[:-k]) == 0
        assert np.array_equal(gs[valid], exp[valid]) and np.all(gs[~valid] == np.uint64(L.INVALID))
        assert np.array_equal(gp, gs) and np.array_equal(g8, gs)
        cnt = C.c_ulonglong()
        gs = np.ascontiguousarray(gs)
        assert emuk.emu_kernel_count_valid(gs.ctypes.data, len(gs), C.byref(cnt)) == 0 and cnt.value == int(valid.sum())


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (63, 20, 128, 0), (31, 20, 64, 1), (47, 20, 128, 1), (31, 15, 64, 0)])
def test_low_complexity_ties_on_the_emulator(emuk, tmp_path, k, m, bits, kind):
    """Equal m-mers inside a window at the BASELINE (k, m) pairs (on uniform bases a tie there has probability ~1e-10 per window): homopolymers, tandem
    repeats of period < k - m + 1, N runs of k bases and more, in the index and in the queries, through every tile-kernel variant == the oracle
    (leftmost minimum: mphf_utils.hpp:118-137, partitioned_mphf.hpp:119-160).  The same inputs run on the GPU in tests/test_gpu_real.py."""
    from real_fixture import valid_windows
    from util import low_complexity_index_records, low_complexity_queries

    recs = low_complexity_index_records(k, 900 + k + m)
    assert sum(1 for r in recs if len(set(r)) == 1) >= 3
    path, po = str(tmp_path / "t.lph"), str(tmp_path / "o.lph")
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=4)
    orc.build(recs, k, m, po, kind=kind, kmer_bits=bits)
    assert open(path, "rb").read() == open(po, "rb").read()  # the product's position-parallel builder on tied minimizers == the reference's state machine
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits)
    qs = low_complexity_queries(recs, k, 5 + k)
    exp = [o(q, False) if len(q) >= k else np.zeros(0, dtype=np.uint64) for q in qs]
    val = [valid_windows(q, k) for q in qs]
    for q, e, v in zip(qs, exp, val):
        if v.size and v.all():
            assert np.array_equal(o(q, True), e)  # clean records: the reference's streaming branch breaks ties like its random branch
    BLK, RAW, S8 = 0x10000, 0x20000, 0x40000
    for streaming, mis, block in [(True, 0, 256), (True, 5, 256 | BLK), (True, 3, 128 | RAW), (False, 0, 256), (True, 0, 128 | S8), (True, 9, 256 | S8)]:
        got = emu_records(emuk, h, qs, k, streaming, mis, block)
        for g, e, v in zip(got, exp, val):
            if streaming:
                assert np.array_equal(g[v], e[v]) and np.all(g[~v] == np.uint64(L.INVALID)), (streaming, mis, block)
            else:
                assert np.array_equal(g, e)


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (63, 20, 128, 0), (21, 8, 64, 1), (47, 17, 128, 0)])
def test_genomic_like_sets_on_the_emulator(emuk, tmp_path, k, m, bits, kind):
    """Index sets with the shape of genomes (tests/util.py: GC skew, tandem repeats, diverged copies of earlier segments -> 5-70 % of the k-mers sit behind
    colliding minimizers, where uniform bases put 0.01 %): the product builder's file == the oracle builder's, and the tile kernels == the oracle with the
    fallback function on the common path."""
    from real_fixture import valid_windows
    from util import genomic_like_records

    recs = genomic_like_records(1000 + k + m, 30, k, 5000)
    path, po = str(tmp_path / "t.lph"), str(tmp_path / "o.lph")
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=4)
    orc.build(recs, k, m, po, kind=kind, kmer_bits=bits)
    assert open(path, "rb").read() == open(po, "rb").read()
    o = orc.Index(path, kind, bits)
    assert o.fallback_keys > 0.03 * o.nkmers
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits)
    noisy = bytearray(recs[1])
    noisy[len(noisy) // 3:len(noisy) // 3 + k + 4] = b"N" * (k + 4)
    batch = recs[:12] + [bytes(noisy), recs[0][::-1], b"AC" * 400, recs[2][:k], b""]
    exp = [o(q, False) if len(q) >= k else np.zeros(0, dtype=np.uint64) for q in batch]
    val = [valid_windows(q, k) for q in batch]
    for q, e, v in zip(batch, exp, val):
        if v.size and v.all():
            assert np.array_equal(o(q, True), e)
    for streaming, mis, block in [(True, 0, 256), (False, 2, 256), (True, 5, 128 | 0x40000)]:
        got = emu_records(emuk, h, batch, k, streaming, mis, block)
        for g, e, v in zip(got, exp, val):
            assert np.array_equal(g[v], e[v]) if streaming else np.array_equal(g, e)


@pytest.mark.parametrize("name", ["se_k31_m20_b64_p", "se_k63_m20_b128_p", "se_k47_m20_b128_u"])
def test_real_data_on_the_emulator(emuk, tmp_path, name):
    """The device sources on real genomic input (tests/golden/real/: the reference's bundled unitigs and query files): slices of the unitigs (members),
    of the Salmonella genome, of the E. coli genome with its runs of N and reads with N, against the committed codes and the oracle."""
    import json

    from real_fixture import CASES, OUT, load_inputs, query_sets, unitig_records, valid_windows

    kind, k, m, bits, ust = CASES[name]
    z = load_inputs()
    recs = unitig_records(z, ust)
    path = os.path.join(OUT, name + ".lph")
    if not os.path.exists(path):
        path = str(tmp_path / "x.lph")
        L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8)
    e = json.load(open(os.path.join(OUT, "expected.json")))["cases"][name]
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits)
    assert h.get_kmer_count() == e["nkmers"]
    qs = query_sets(z)
    heads = np.load(os.path.join(OUT, f"codes_{name}.npz"))
    batch = [qs["salmonella"][0][:6000], qs["ecoli"][0][9000:15_000], qs["ecoli"][1][5000:9000]] + qs["fastq"][:30] + qs["fastq"][-30:]
    batch += sorted(recs, key=len)[:40] + [r[:5000] for r in sorted(recs, key=len)[-3:]]
    exp = [o(q, False) if len(q) >= k else np.zeros(0, dtype=np.uint64) for q in batch]
    assert np.array_equal(exp[0], heads["salmonella"][:len(exp[0])])  # what the build container's oracle said
    val = [valid_windows(q, k) for q in batch]
    assert any((~v).any() for v in val)
    S8 = 0x40000
    for streaming, mis, block in [(True, 0, 256), (False, 0, 256), (True, 7, 128 | S8)]:
        got = emu_records(emuk, h, batch, k, streaming, mis, block)
        for g, x, v in zip(got, exp, val):
            if streaming:
                assert np.array_equal(g[v], x[v]) and np.all(g[~v] == np.uint64(L.INVALID))
            else:
                assert np.array_equal(g, x)


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (63, 20, 128, 0), (21, 8, 64, 1)])
def test_packed_and_pack_kernels_on_the_emulator(emuk, tmp_path, k, m, bits, kind):
    recs = dedup_records(random_records(k + 3 * m, 8, 200, 1500), k)
    path = str(tmp_path / "t.lph")
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=4)
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits)
    seq = b"".join(recs) + b"NNACGTNacgtu" + random_records(8, 1, 700, 700)[0]
    pos = np.arange(0, len(seq) - k + 1, dtype=np.uint64)
    pk = pack_kmers(seq, [int(p) for p in pos], k, bits)
    raw = np.frombuffer(seq + b"\0" * 16, dtype=np.uint8).copy()
    got_pk = np.zeros_like(pk)
    assert emuk.emu_kernel_pack_kmers(raw.ctypes.data, pos.ctypes.data, len(pos), k, 1 if bits == 128 else 0, got_pk.ctypes.data) == 0
    assert np.array_equal(got_pk, pk)
    img = h.image()
    out = np.empty(len(pos), dtype=np.uint64)
    assert emuk.emu_kernel_query_packed(img.ctypes.data, img.size, pk.ctypes.data, len(pos), out.ctypes.data) == 0
    assert np.array_equal(out, o.query_packed(pk))


# ------------------------------------------------------------------------------------------------------------------
# GPU pilot search of the index builder (search_device.cuh kernels + search_rounds.hpp round loop) on the emulator
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_keys,look_ahead,steps,seed", [(3000, 16384, 32, 1), (2000, 16384, 4, 2), (2500, 8192, 1, 3), (800, 100, 1000000, 4), (4000, 16384, 32, 5)])
def test_gpu_pilot_search_on_the_emulator(emuk, n_keys, look_ahead, steps, seed):
    """Speculative window + claim/validate/commit must give exactly the sequential first-fit pilots and bitmap, for any step
    budget and for (clamped) out-of-range look-ahead values -- the configuration that once committed unsearched buckets."""
    lib = emuk
    lib.emu_pilot_search.restype = C.c_int
    lib.emu_pilot_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                     C.c_void_p, C.POINTER(C.c_uint64)]
    lib.ref_pilot_search.restype = C.c_int
    lib.ref_pilot_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(seed)
    hashes = rng.integers(0, 2**63, n_keys, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n_keys, dtype=np.uint64)
    hashes = np.unique(hashes)
    n = len(hashes)
    table_size = int(n / 0.94)
    if table_size & (table_size - 1) == 0:
        table_size += 1
    nb = int(np.ceil(3.0 * n / np.log2(n)))
    # skewed buckets like PTHash's: 60 % of the keys in the first 30 % of the buckets
    dense = max(1, int(0.3 * nb))
    r = rng.random(n)
    bucket = np.where(r < 0.6, rng.integers(0, dense, n), rng.integers(dense, nb, n)).astype(np.int64)
    perm = np.argsort(bucket, kind="stable")
    hs = np.ascontiguousarray(hashes[perm])
    counts = np.bincount(bucket, minlength=nb)
    bstart = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    order = np.array(sorted(range(nb), key=lambda b: (-counts[b], b)), dtype=np.uint32)
    n_nonempty = int((counts > 0).sum())
    words = (table_size + 63) // 64 * 2 + 2
    p_emu, p_ref = np.zeros(n_nonempty, dtype=np.uint64), np.zeros(n_nonempty, dtype=np.uint64)
    t_emu, t_ref = np.zeros(words, dtype=np.uint32), np.zeros(words, dtype=np.uint32)
    rounds = C.c_uint64()
    assert lib.ref_pilot_search(hs.ctypes.data, bstart.ctypes.data, order.ctypes.data, n_nonempty, table_size, 1, p_ref.ctypes.data, t_ref.ctypes.data) == 0
    # rounds enqueued per host look at the control words: 1 = the old lock-step, larger = rounds running on after the search is over
    per_sync = 1 + (seed * 7) % 9
    assert lib.emu_pilot_search(hs.ctypes.data, bstart.ctypes.data, order.ctypes.data, n_nonempty, table_size, 1, look_ahead, steps, per_sync,
                                p_emu.ctypes.data, t_emu.ctypes.data, C.byref(rounds)) == 0
    assert np.array_equal(p_emu, p_ref) and np.array_equal(t_emu, t_ref)
    assert int(np.unpackbits(t_emu.view(np.uint8)).sum()) == n and rounds.value >= 1


# ------------------------------------------------------------------------------------------------------------------
# LDS-race check: a kernel whose barriers are complete gives the same answer for every order in which the threads of a
# workgroup run between two synchronisation points (tests/emu/emu_runtime.cpp: set_schedule).  The default order runs lane 0
# first, so a consumer in a low lane reading what a high lane has not written yet only fails under the other orders.
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode,seed", [(1, 0), (2, 12345), (2, 987654321)])
def test_kernels_under_other_thread_schedules(emuk, tmp_path, mode, seed):
    emuk.emu_set_schedule.restype = None
    emuk.emu_set_schedule.argtypes = [C.c_int, C.c_uint64]
    emuk.emu_set_schedule(mode, seed)
    try:
        test_tile_kernels_on_the_emulator(emuk, tmp_path, 31, 20, 64, 0)
        test_tile_kernels_on_the_emulator(emuk, tmp_path, 9, 7, 64, 0)     # W = 3 path, > 256 leaders per tile
        test_tile_kernels_on_the_emulator(emuk, tmp_path, 47, 17, 128, 1)  # unpartitioned, 128-bit k-mers
        test_tile_kernels_non_acgt_on_the_emulator(emuk, tmp_path)
        test_packed_and_pack_kernels_on_the_emulator(emuk, tmp_path, 31, 20, 64, 0)
        test_gpu_pilot_search_on_the_emulator(emuk, 1500, 16384, 8, 3)
    finally:
        emuk.emu_set_schedule(0, 0)


REF_NT4 = "/root/reference/src/constants.cpp"


@pytest.mark.skipif(not os.path.exists(REF_NT4), reason="reads the literal of the reference's seq_nt4_table where /root/reference exists")
def test_base_encoding_against_the_reference_table(emuk, tmp_path):
    """Pin of SURVEY 8(a) a9: constants::seq_nt4_table (src/constants.cpp:5-13) is read from the reference tree as DATA (256 numbers) and
    every byte value is pushed through the oracle and through the product's streaming and random kernels (emulator): a byte the table
    maps to 0..3 must behave exactly like that base, a byte it maps to 4 must void the streaming windows that contain it and read
    as 'A' in random mode (mphf_utils.hpp:108)."""
    import re

    text = open(REF_NT4).read()
    body = text[text.index("seq_nt4_table"):]
    table = [int(x) for x in re.findall(r"\b\d+\b", body[body.index("{"):body.index("}")])]
    assert len(table) == 256 and set(table) <= {0, 1, 2, 3, 4}
    k, m = 31, 20
    recs = random_records(9001, 3, 300, 500)
    path = str(tmp_path / "t.lph")
    L.build(recs, k, m, path, kmer_bits=64, threads=2)
    o = orc.Index(path, 0, 64)
    h = L.Mphf.load(path, kmer_bits=64)
    base = recs[0][:90]
    pos = 45
    queries, clean = [], []
    for c in range(256):
        q = bytearray(base)
        q[pos] = c
        queries.append(bytes(q))
        q[pos] = b"ACGT"[table[c]] if table[c] < 4 else ord("A")
        clean.append(bytes(q))
    got_s = emu_records(emuk, h, queries, k, True)
    got_r = emu_records(emuk, h, queries, k, False)
    for c in range(256):
        exp = o(clean[c], False)
        assert np.array_equal(got_r[c], exp), c                      # random mode: the table's code, or 'A' for the 4s
        assert np.array_equal(o(queries[c], False), exp), c          # ... and the oracle's random branch agrees
        os_ = o(queries[c], True)
        if table[c] < 4:
            assert np.array_equal(got_s[c], exp) and np.array_equal(os_, exp), c
        else:
            bad = np.zeros(len(base) - k + 1, dtype=bool)
            bad[max(0, pos - k + 1):pos + 1] = True
            assert np.all(got_s[c][bad] == np.uint64(L.INVALID)) and np.array_equal(got_s[c][~bad], exp[~bad]), c
            # (the oracle's streaming branch restates the reference literally, including the spurious extra code its stale min_pos can
            # emit after a reset -- SURVEY 8(a) note 4(ii) -- so only its count is bounded here)
            assert int((~bad).sum()) <= len(os_) <= int((~bad).sum()) + 1, c


@pytest.mark.parametrize("k,bits", [(31, 64), (31, 128), (63, 128), (21, 64)])
def test_pthash_competitor_kernels_on_the_emulator(emuk, tmp_path, k, bits):
    """Competitor harness: a bare PTHash over k-mers (lphq_build_pthash == the oracle's ptbb_build bytes) through the
    packed kernel and the record-tile kernel (KIND 2): one code per window, LPHQ_INVALID where ptbb_file_itr yields nothing."""
    recs = dedup_records(random_records(900 + k, 10, k, 2500), k)
    path, ref = str(tmp_path / "p.pthash"), str(tmp_path / "o.pthash")
    L.build_pthash(recs, k, path, kmer_bits=bits, threads=4)
    orc.ptbb_build(recs, k, ref, kmer_bits=bits)
    assert open(path, "rb").read() == open(ref, "rb").read()
    o = orc.PtHash(path, k, bits)
    h = L.PtHash.load(path, k, kmer_bits=bits)
    assert h.get_kmer_count() == o.num_keys == sum(len(r) - k + 1 for r in recs)
    img = h.image()
    # packed: members and non-members
    seq = b"".join(recs) + random_records(3, 1, 900, 900)[0]
    pk = pack_kmers(seq, np.arange(0, len(seq) - k + 1), k, bits)
    out = np.zeros(len(seq) - k + 1, dtype=np.uint64)
    assert emuk.emu_kernel_query_packed(img.ctypes.data, img.size, pk.ctypes.data, len(out), out.ctypes.data) == 0
    assert np.array_equal(out, o.query_packed(pk))
    # records with invalid bases, short and empty records, two alignments
    q = bytearray(recs[0])
    q[7] = q[len(q) // 2] = ord("N")
    batch = [recs[1], bytes(q), b"", recs[2][: k - 1], recs[2][:k]] + recs[3:] + random_records(4, 5, 1, 700)
    for mis in (0, 11):
        got = emu_records(emuk, h, batch, k, False, mis)
        for r, g in zip(batch, got):
            bad = np.array([c not in b"ACGTUacgtu" for c in r])
            cs = np.concatenate([[0], np.cumsum(bad)])
            valid = (cs[k:] - cs[:-k]) == 0 if len(r) >= k else np.zeros(0, dtype=bool)
            assert np.array_equal(g[valid], o(r)) and (g[~valid] == np.uint64(2**64 - 1)).all()
    allc = np.concatenate([o(r) for r in recs])
    assert len(np.unique(allc)) == len(allc) == o.num_keys and int(allc.max()) == len(allc) - 1  # ptbb_build --check


@pytest.mark.parametrize("eb,dt", [(1, np.uint8), (2, np.uint16), (4, np.uint32), (8, np.uint64)])
def test_gather_payload_on_the_emulator(emuk, eb, dt):
    """k_gather_payload (result consumer): payload[code], `missing` for codes that are not slots."""
    rng = np.random.default_rng(eb)
    n_slots, n = 1000, 3001
    table = rng.integers(0, np.iinfo(dt).max, n_slots, dtype=dt)
    codes = rng.integers(0, n_slots, n).astype(np.uint64)
    codes[::7] = np.uint64(2**64 - 1)
    codes[3::11] = np.uint64(n_slots) + codes[3::11]
    out = np.zeros(n, dtype=dt)
    emuk.emu_kernel_gather_payload.restype = C.c_int
    emuk.emu_kernel_gather_payload.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_void_p]
    assert emuk.emu_kernel_gather_payload(codes.ctypes.data, n, table.ctypes.data, n_slots, eb, 0xABCDEF0123456789, out.ctypes.data) == 0
    miss = np.array([0xABCDEF0123456789 & (2**(8 * eb) - 1)], dtype=np.uint64).astype(dt)[0]
    want = np.where(codes < n_slots, table[np.minimum(codes, n_slots - 1).astype(np.int64)], miss)
    assert np.array_equal(out, want)


def bbhash_golden_cases():
    d = os.path.join(ROOT, "tests", "golden", "bbhash")
    return sorted(os.path.splitext(f)[0] for f in os.listdir(d) if f.endswith(".npz"))


def valid_windows(r, k):
    bad = np.array([c not in b"ACGTUacgtu" for c in r])
    cs = np.concatenate([[0], np.cumsum(bad)])
    return (cs[k:] - cs[:-k]) == 0 if len(r) >= k else np.zeros(0, dtype=bool)


@pytest.mark.parametrize("name", bbhash_golden_cases())
def test_bbhash_competitor_against_the_real_boophf(emuk, tmp_path, name):
    """PINNED against the reference's own code: tests/golden/bbhash/*.bbh were written by the REAL BooPHF (external/BooPHF.hpp compiled from
    /root/reference, tests/golden/make_bbhash_golden.py) and the expected codes are its lookup().  The loader accepts the files (every redundant
    field cross-checked), the device lookup (KIND 3; packed kernel and record-tile kernel on the emulator) returns the same code for every window of
    k valid bases -- members, non-members (incl. ULLONG_MAX from the exact map) -- and LPHQ_INVALID where ptbb_file_itr yields nothing.  The
    product's builder, given the same records, writes a function with the same bit arrays (byte-identical up to the order of the exact map)."""
    gd = os.path.join(ROOT, "tests", "golden", "bbhash")
    z = np.load(os.path.join(gd, name + ".npz"))
    k, bits = int(z["k"]), int(z["kmer_bits"])
    h = L.BbHash.load(os.path.join(gd, name + ".bbh"), k, kmer_bits=bits)
    recs = [bytes(z["records"][z["rec_off"][i]:z["rec_off"][i + 1]]) for i in range(len(z["rec_off"]) - 1)]
    queries = [bytes(z["queries"][z["q_off"][i]:z["q_off"][i + 1]]) for i in range(len(z["q_off"]) - 1)]
    assert h.info.kind == L.BBHASH and h.get_kmer_count() == sum(len(r) - k + 1 for r in recs)
    want = z["codes"]
    for mis in (0, 9):
        got = emu_records(emuk, h, queries, k, True, mis)
        at = 0
        for q, g in zip(queries, got):
            v = valid_windows(q, k)
            n = int(v.sum())
            assert np.array_equal(g[v], want[at:at + n]) and (g[~v] == np.uint64(2**64 - 1)).all()
            at += n
        assert at == len(want)
    # packed kernel on the windows of the member records: a bijection onto [0, n)
    img = h.image()
    seq = recs[0]
    pk = pack_kmers(seq, np.arange(0, len(seq) - k + 1), k, bits)
    out = np.zeros(len(seq) - k + 1, dtype=np.uint64)
    assert emuk.emu_kernel_query_packed(img.ctypes.data, img.size, pk.ctypes.data, len(out), out.ctypes.data) == 0
    assert np.array_equal(out, want[: len(out)])
    # the product's builder on the same records: same bit arrays as the reference's
    ours = tmp_path / "ours.bbh"
    L.build_bbhash(recs, k, str(ours), kmer_bits=bits, gamma=float(z["gamma"]), threads=3)
    a, b = ours.read_bytes(), open(os.path.join(gd, name + ".bbh"), "rb").read()
    nfinal = h.info.fallback_kmers
    tail = 8 + nfinal * (bits // 8 + 8)  # the exact map: count + (key, value) pairs, in an order the reference does not define
    assert len(a) == len(b) and a[: len(a) - tail] == b[: len(b) - tail]
    h2 = L.BbHash.load(str(ours), k, kmer_bits=bits)
    every = np.concatenate(emu_records(emuk, h2, recs, k, True, 0))
    assert len(np.unique(every)) == len(every) == h2.get_kmer_count() and int(every.max()) == len(every) - 1


# ------------------------------------------------------------------------------------------------------------------
# index construction, part 1 on the GPU: k_extract_runs (extract_device.cuh)
# ------------------------------------------------------------------------------------------------------------------
def expected_runs(recs, k, m, seed, first_base=0, lost=None):
    """Super-k-mers by the independent Python model (tests/golden/pymodel.py): maximal runs of consecutive k-mers of a stretch of valid
    bases whose leftmost-minimum m-mer sits at the same position -> (minimizer, first window (global), offset, size)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import pymodel

    runs, at = [], first_base
    lost = [0] if lost is None else lost  # lost[0] += the windows the reference leaves out of every run
    for rec in recs:
        pos = 0
        text = bytes(rec)  # (iterating bytes yields the ints pymodel.NT is keyed by)
        while pos < len(text):
            if text[pos] not in pymodel.NT:
                pos += 1
                continue
            end = pos
            while end < len(text) and text[end] in pymodel.NT:
                end += 1
            codes = [pymodel.NT[c] for c in text[pos:end]]
            if len(codes) == k and end < len(text):  # exactly k valid bases closed by a non-ACGT byte: the reference counts the k-mer and pushes no record
                lost[0] += 1                         # (include/minimizer.hpp:60-66,139-152)
            elif len(codes) >= k:
                prev = None
                for w, (_kv, mm, p, best) in enumerate(pymodel.windows_of(codes, k, m, seed)):
                    if best == prev:
                        runs[-1][3] += 1
                    else:
                        runs.append([mm, at + pos + w, p, 1])
                    prev = best
            pos = end
        at += len(rec)
    return sorted(tuple(r) for r in runs)


@pytest.mark.parametrize("k,m,form", [(31, 20, 0), (31, 20, 1), (63, 20, 0), (63, 20, 2), (21, 8, 0), (15, 15, 0), (33, 5, 0), (9, 7, 0), (63, 1, 1), (47, 32, 0), (2, 1, 0)])
def test_extract_runs_kernel_on_the_emulator(emu, k, m, form):
    """k_extract_runs over ragged multi-record batches (records shorter than k, empty, non-ACGT bases, many records per tile, records spanning
    several tiles): the set of super-k-mer records equals the independent Python model's; a too-small output buffer is reported through the counter."""
    emu.emu_kernel_extract_runs.restype = C.c_longlong
    emu.emu_kernel_extract_runs.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                            C.c_uint64, C.c_int, C.POINTER(C.c_ulonglong)]
    rng = np.random.default_rng(k * 100 + m)
    recs = random_records(k + m, 5, 900, 2600) + [b"", b"ACGT"[: min(4, k - 1)]] + random_records(k, 40, 0, 2 * k) + random_records(m, 3, 3000, 5000)
    noisy = bytearray(recs[1])
    for p in rng.integers(0, len(noisy), 6):
        noisy[p] = ord("N")
    recs[1] = bytes(noisy)
    recs[2] = recs[2].lower()
    # stretches of exactly k valid bases: closed by a non-ACGT byte (the reference leaves their window out of every run), closed by the record's end (kept), at a
    # record's start, two in a row, k + 1 bases (kept), beside tile borders (the long record)
    kk = [random_records(1000 + i, 1, k, k)[0] for i in range(8)]
    recs += [kk[0] + b"N" + random_records(77, 1, 300, 300)[0] + b"N" + kk[1] + b"NN" + kk[2] + b"N" + kk[3], kk[4] + b"N", b"N" + kk[5], kk[6] + b"A" + b"N" + kk[7][:k - 1] + b"N"]
    big = bytearray(recs[-5])
    for at in range(500, len(big) - 2 * k, 509):
        big[at] = ord("N")
        big[at + k + 1] = ord("N")
    recs[-5] = bytes(big)
    seed, first_base = 42, 1_000_000_007
    lost_want = [0]
    want = expected_runs(recs, k, m, seed, first_base, lost_want)
    assert lost_want[0] >= 8
    flat = np.frombuffer(b"".join(recs), dtype=np.uint8)
    raw = np.full(len(flat) + 256, ord("#"), dtype=np.uint8)
    base = (-raw.ctypes.data) % 16 + 64
    raw[base:base + len(flat)] = flat
    seq_off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    cap = len(want) + 7
    mm = np.zeros(cap, dtype=np.uint64)
    val = np.zeros(2 * cap, dtype=np.uint64)
    lost = C.c_ulonglong()
    n = emu.emu_kernel_extract_runs(raw.ctypes.data + base, len(flat), seq_off.ctypes.data, len(recs), k, m, seed, first_base, mm.ctypes.data, val.ctypes.data, cap, form,
                                    C.byref(lost))
    assert n == len(want) and lost.value == lost_want[0]
    got = sorted((int(mm[i]), int(val[2 * i]), int(val[2 * i + 1]) & 0xFFFFFFFF, int(val[2 * i + 1]) >> 32) for i in range(n))
    assert got == want
    # not enough room: the counter still says how many there are, nothing is written beyond the buffer
    small = max(1, len(want) // 3)
    mm2 = np.full(small + 1, 0xDEAD, dtype=np.uint64)
    val2 = np.full(2 * small + 2, 0xDEAD, dtype=np.uint64)
    n2 = emu.emu_kernel_extract_runs(raw.ctypes.data + base, len(flat), seq_off.ctypes.data, len(recs), k, m, seed, first_base, mm2.ctypes.data, val2.ctypes.data, small, form,
                                     C.byref(lost))
    assert n2 == len(want) and lost.value == lost_want[0] and mm2[small] == 0xDEAD and val2[2 * small] == 0xDEAD


def test_classify_kernels_on_the_emulator(emu):
    """k_group_flags / k_group_scatter (minimizer::classify on the device, src/minimizer.cpp:5-50) over sorted records with repeated minimizers at the
    ends, across workgroup borders and in long stretches: distinct minimizers ascending with the (offset, size) of singletons and 0 for the others, the
    colliding records as a set, the k-mer total."""
    emu.emu_kernel_classify.restype = C.c_longlong
    emu.emu_kernel_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_ulonglong)]
    rng = np.random.default_rng(5)
    for n, distinct in ((1, 1), (2, 1), (700, 650), (5000, 1200), (1024, 1024), (513, 3)):
        keys = np.sort(rng.choice(np.arange(1, 10 * distinct + 2, dtype=np.uint64), size=distinct, replace=False))
        mm = np.sort(np.concatenate([keys, rng.choice(keys, size=n - distinct)])) if n > distinct else keys.copy()
        first = rng.permutation(n).astype(np.uint64) * np.uint64(100)
        p1 = rng.integers(0, 62, n).astype(np.uint64)
        size = rng.integers(1, 64, n).astype(np.uint64)
        val = np.empty(2 * n, dtype=np.uint64)
        val[0::2] = first
        val[1::2] = p1 | (size << np.uint64(32))
        umm = np.zeros(n, dtype=np.uint64)
        ups = np.zeros(n, dtype=np.uint16)
        cf = np.zeros(n, dtype=np.uint64)
        cs = np.zeros(n, dtype=np.uint32)
        nc, km = C.c_uint64(), C.c_ulonglong()
        nu = emu.emu_kernel_classify(mm.ctypes.data, val.ctypes.data, n, umm.ctypes.data, ups.ctypes.data, cf.ctypes.data, cs.ctypes.data, C.byref(nc), C.byref(km))
        vals, counts = np.unique(mm, return_counts=True)
        assert nu == len(vals) and np.array_equal(umm[:nu], vals) and km.value == int(size.sum())
        single = counts == 1
        idx_of_single = np.searchsorted(mm, vals[single])
        assert np.array_equal(ups[:nu][single], (p1[idx_of_single] | (size[idx_of_single] << np.uint64(8))).astype(np.uint16))
        assert np.all(ups[:nu][~single] == 0)
        in_multi = np.isin(mm, vals[~single])
        assert nc.value == int(in_multi.sum())
        assert sorted(zip(cf[:nc.value].tolist(), cs[:nc.value].tolist())) == sorted(zip(first[in_multi].tolist(), size[in_multi].tolist()))
