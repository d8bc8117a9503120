"""CPU side of the real-data fixtures (tests/golden/real/, made by tests/golden/make_real_golden.py from the reference's bundled data):
the fixture is what its generator says it is, the oracle reproduces it from the committed inputs alone, and the PRODUCT's CPU builder writes
the files the oracle builder wrote -- on real unitigs (minimizer skew, 0.29 % of the k-mers behind colliding minimizers), not on uniform bases."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

import lphash_amd as L
import orc
from real_fixture import CASES, HEAD, OUT, load_inputs, pack_2bit, query_sets, unitig_records, valid_windows

EXPECTED = json.load(open(os.path.join(OUT, "expected.json")))
DATA = "/root/reference/data"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def test_fixture_census_and_shape():
    z = load_inputs()
    u31, u63 = unitig_records(z, "ust_k31"), unitig_records(z, "ust_k63")
    assert len(u31) == 611 and sum(map(len, u31)) == 4_951_824  # all of se.ust.k31
    assert sum(map(len, u63)) >= 1_500_000 and all(set(r) <= set(b"ACGT") for r in u63[:5])
    c1 = EXPECTED["cases"]["se_k31_m20_b64_p"]
    assert (c1["nkmers"], c1["distinct_minimizers"], c1["fallback_kmers"]) == (4_933_494, 758_441, 14_084)  # the independent census
    qs = query_sets(z)
    assert len(qs["salmonella"][0]) == 300_000 and len(qs["ecoli"]) == 2 and len(qs["fastq"]) >= 2000
    assert sum(r.count(b"N") for r in qs["ecoli"]) > 1000 and sum(r.count(b"N") for r in qs["fastq"]) >= 200  # dirty queries: runs of N, reads with N
    assert set(EXPECTED["cases"]) == set(CASES)
    for name, e in EXPECTED["cases"].items():
        for q, qe in e["queries"].items():
            assert qe["valid_windows"] <= qe["windows"] and qe["members"] <= qe["valid_windows"]
        assert e["queries"]["salmonella"]["members"] > 10_000  # real members among the queries, real non-members beside them


@pytest.mark.skipif(not os.path.isdir(DATA), reason="bundled reference data is only present in the build container")
def test_fixture_inputs_are_the_bundled_data():
    """The committed inputs against the files they were cut from (only where /root/reference exists)."""
    from test_bundled_cpu import fasta_records

    z = load_inputs()
    u31 = fasta_records(os.path.join(DATA, "unitigs_stitched", "se.ust.k31.fa.gz"))
    assert np.array_equal(z["ust_k31_2bit"], pack_2bit(b"".join(u31))) and [len(r) for r in u31] == z["ust_k31_len"].tolist()
    u63 = fasta_records(os.path.join(DATA, "unitigs_stitched", "se.ust.k63.fa.gz"))
    got = unitig_records(z, "ust_k63")
    assert got == u63[:len(got)]
    sal = fasta_records(os.path.join(DATA, "queries", "salmonella_enterica.fasta.gz"))[0]
    eco = fasta_records(os.path.join(DATA, "queries", "ecoli1.fasta.gz"))[0]
    assert z["q_salmonella"].tobytes() == sal[:300_000] and z["q_ecoli_a"].tobytes() == eco[330_000:560_000] and z["q_ecoli_b"].tobytes() == eco[995_000:1_030_000]
    with gzip.open(os.path.join(DATA, "queries", "SRR5833294.10K.fastq.gz"), "rb") as f:
        reads = f.read().split(b"\n")[1::4]
    assert query_sets(z)["fastq"][:2000] == reads[:2000]


@pytest.mark.parametrize("name", list(CASES))
def test_product_cpu_builder_writes_the_oracle_file_on_real_unitigs(tmp_path, name):
    """lphq_build (host extraction, sequential pilot search, 8 threads) on the real unitigs == the file orc_build wrote in the build container
    (SHA-256 in expected.json; C1's file byte for byte), and the oracle reading that file reproduces every committed code digest."""
    kind, k, m, bits, ust = CASES[name]
    e = EXPECTED["cases"][name]
    z = load_inputs()
    recs = unitig_records(z, ust)
    path = str(tmp_path / "x.lph")
    st = L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8)
    data = open(path, "rb").read()
    assert len(data) == e["lph_bytes"] and hashlib.sha256(data).hexdigest() == e["lph_sha256"]
    assert st["num_kmers"] == e["nkmers"] and st["distinct_minimizers"] == e["distinct_minimizers"]
    if name == "se_k31_m20_b64_p":
        assert data == open(os.path.join(OUT, name + ".lph"), "rb").read()
    o = orc.Index(path, kind, bits)
    assert (o.nkmers, o.distinct_minimizers, o.fallback_keys) == (e["nkmers"], e["distinct_minimizers"], e["fallback_kmers"])
    heads = np.load(os.path.join(OUT, f"codes_{name}.npz"))
    for qname, qrecs in query_sets(z).items():
        rnd = [o(q, False) for q in qrecs]
        flat = np.concatenate(rnd)
        assert np.array_equal(flat[:HEAD], heads[qname]) and sha(flat) == e["queries"][qname]["random_sha256"]
        st_codes = np.concatenate([r[valid_windows(q, k)] for q, r in zip(qrecs, rnd)])
        assert st_codes.size == e["queries"][qname]["valid_windows"] and sha(st_codes) == e["queries"][qname]["streaming_sha256"]


@pytest.mark.parametrize("k,m,bits,kind,ust,n_rec", [(31, 20, 64, 0, "ust_k31", 70), (63, 17, 128, 0, "ust_k63", 40), (47, 20, 128, 1, "ust_k31", 80), (31, 15, 64, 1, "ust_k31", 80)])
def test_python_model_equals_the_oracle_on_real_unitigs(tmp_path, k, m, bits, kind, ust, n_rec):
    """The third restatement (tests/golden/pymodel.py: pure Python, no code shared with the C oracle or the product) on a slice of the REAL unitigs -- 
    medium-sized records of the set, real minimizer skew and colliding minimizers -- and on low-complexity records (homopolymers, tandem repeats: tied minimizer
    hashes): its serialized bytes == the oracle builder's == the product builder's, its codes == the oracle's."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    import pymodel
    from util import low_complexity_index_records

    z = load_inputs()
    real = [r for r in unitig_records(z, ust) if 300 <= len(r) <= 12_000][:n_rec]  # (pure Python: ~10^5 k-mers)
    assert sum(map(len, real)) > 40_000
    for recs in (real, [r for r in low_complexity_index_records(k, 77) if len(r) < 600][:40]):
        po, pp = str(tmp_path / "o.lph"), str(tmp_path / "p.lph")
        orc.build(recs, k, m, po, kind=kind, kmer_bits=bits)
        L.build(recs, k, m, pp, kind=kind, kmer_bits=bits, threads=4)
        mdl = pymodel.Model(recs, k, m, kind, bits)
        want = open(po, "rb").read()
        assert mdl.serialize() == want and open(pp, "rb").read() == want
        ix = orc.Index(po, kind, bits)
        assert (mdl.nkmers, mdl.D, len(mdl.fallback_keys)) == (ix.nkmers, ix.distinct_minimizers, ix.fallback_keys)
        for q in recs[:3] + [query_sets(z)["ecoli"][0][9_900:10_400], b"A" * (k + 30), (b"AC" * k)]:
            clean = bytes(c if c in b"ACGT" else ord("A") for c in q)  # the model takes valid bases only; the random branch reads anything else as A (mphf_utils.hpp:108)
            assert np.array_equal(np.array(mdl.query(clean), dtype=np.uint64), ix(q, False))
