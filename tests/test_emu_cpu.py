"""CPU regression of the device lookup chain: lphash_amd/csrc/device_query.cuh compiled for the host (tests/emu/, test
infrastructure only) over the product's flat image, against the oracle.  Covers the image re-packing (fused blocks, pair
blocks, hashed-pilot tables, free-slot table, exact 32-bit modular reduction) and minimizer_of / phf_position / uni_lookup /
pair_lookup / finish_hval bit for bit -- without a GPU.  Kernels (tiling, LDS phases) are covered by the -m gpu tests."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

import lphash_amd as L
import orc
from util import dedup_records, pack_kmers, random_records

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(d, "libemu.so"))
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
