"""In-container checks on the reference's bundled data with the README's own example configurations (README.md:168-201).
Skipped where /root/reference does not exist (the GPU box).  CPU only: the CLI builds with --cpu, the oracle is the checker."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import orc  # noqa: E402

DATA = "/root/reference/data"
pytestmark = pytest.mark.skipif(not os.path.isdir(DATA), reason="bundled reference data is only present in the build container")


def fasta_records(path):
    recs, cur = [], None
    with gzip.open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if cur is not None:
                    recs.append(b"".join(cur))
                cur = []
            elif cur is not None:
                cur.append(line.rstrip(b"\r\n"))
    if cur is not None:
        recs.append(b"".join(cur))
    return recs


@pytest.mark.parametrize("name,k,m,bits,tool", [("se.ust.k31.fa.gz", 31, 15, 64, "build-p"), ("se.ust.k63.fa.gz", 63, 17, 128, "build-p"),
                                                ("se.ust.k47.fa.gz", 47, 20, 128, "build-u")])
def test_readme_build_examples(tmp_path, name, k, m, bits, tool):
    """`lphash build-p -i data/unitigs_stitched/se.ust.kK.fa.gz -k K -m M -o X` (README examples): the CLI's file equals the oracle's
    restatement of the reference builder byte for byte, and the reference's --check properties (src/build.cpp:53-70) hold on it:
    the codes of all indexed k-mers are a permutation of [0, n) and streaming == random."""
    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    src = os.path.join(DATA, "unitigs_stitched", name)
    out, ref = tmp_path / "cli.lph", tmp_path / "orc.lph"
    r = subprocess.run([exe, tool, "-i", src, "-k", str(k), "-m", str(m), "-o", str(out), "--kmer-bits", str(bits), "--cpu", "-t", "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    fields = r.stdout.strip().split(",")
    assert len(fields) == 8 and fields[0] == src and int(fields[1]) == k and int(fields[2]) == m  # README "Output format"
    recs = fasta_records(src)
    kind = orc.PARTITIONED if tool == "build-p" else orc.UNPARTITIONED
    orc.build(recs, k, m, str(ref), kind=kind, kmer_bits=bits)
    assert out.read_bytes() == ref.read_bytes()
    o = orc.Index(str(out), kind, bits)
    n = sum(len(x) - k + 1 for x in recs if len(x) >= k)
    seen = np.zeros(n, dtype=np.uint8)
    for x in recs:
        if len(x) < k:
            continue
        s = o(x, True)
        assert len(s) == len(x) - k + 1 and int(s.max()) < n
        assert np.array_equal(s, o(x, False))          # streaming == random
        assert not seen[s].any()                       # no collision across records
        seen[s] = 1
        assert len(np.unique(s)) == len(s)             # ... nor within one
    assert seen.all()                                  # minimal: every slot is used
    assert abs(float(fields[7]) - os.path.getsize(out) * 8 / n) < 1e-3  # bits / k-mer of the serialized function
    if name == "se.ust.k31.fa.gz":  # README query example: ./lphash query-p -i se_k31_m15.lph -q data/queries/salmonella_enterica.fasta.gz
        q = fasta_records(os.path.join(DATA, "queries", "salmonella_enterica.fasta.gz"))
        assert len(q) == 1 and len(q[0]) == 4857450  # 1 record, 4,857,450 bases -> 4,857,420 k-mers per pass
        a, b = o(q[0], True), o(q[0], False)
        assert len(a) == 4857420 and np.array_equal(a, b)


def test_c1_query_p_bundled_k31_m20(tmp_path):
    """BASELINE.json configs[0] ("C1") as configured: build-p on data/unitigs_stitched/se.ust.k31.fa.gz with k=31 m=20, then the reference's
    query driver (src/query.cpp:24-88, oracle restatement: gz inflate + kseq parsing + both passes) on data/queries/salmonella_enterica.fasta.gz.
    Census of SURVEY 8(c) (computed there independently with numpy): 4,933,494 k-mers, 758,441 distinct minimizers, 14,084 k-mers in the
    fallback function; the codes of the index set are a bijection onto [0, n) and both passes agree; the driver reads 1 record ->
    4,857,420 k-mers per pass."""
    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    src = os.path.join(DATA, "unitigs_stitched", "se.ust.k31.fa.gz")
    qry = os.path.join(DATA, "queries", "salmonella_enterica.fasta.gz")
    out = tmp_path / "se_k31_m20.lph"
    r = subprocess.run([exe, "build-p", "-i", src, "-k", "31", "-m", "20", "-o", str(out), "--kmer-bits", "64", "--cpu", "-t", "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    o = orc.Index(str(out), orc.PARTITIONED, 64)
    assert (o.nkmers, o.distinct_minimizers, o.fallback_keys) == (4_933_494, 758_441, 14_084)
    recs = fasta_records(src)
    assert len(recs) == 611 and sum(len(x) for x in recs) == 4_951_824
    seen = np.zeros(o.nkmers, dtype=np.uint8)
    for x in recs:
        s = o(x, True)
        assert np.array_equal(s, o(x, False)) and int(s.max()) < o.nkmers and not seen[s].any()
        seen[s] = 1
    assert seen.all()
    assert orc.file_digest(qry)[:2] == (1, 4_857_450)
    n_s, n_r, us_s, us_r = o.query_file(qry)
    assert n_s == n_r == 4_857_420 and us_s > 0 and us_r > 0
    q = fasta_records(qry)[0]
    assert np.array_equal(o(q, True), o(q, False))
