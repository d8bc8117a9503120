"""GPU parity on REAL genomic input and on low-complexity input: until round 5 every base a HIP kernel had read was i.i.d.
uniform.  tests/golden/real/ holds the reference's bundled Salmonella unitigs and slices of its query files with the codes the oracle gave them
in the build container; the oracle itself (oracle/liblphoracle.so) is on the GPU box too, so every path is compared twice -- with the committed
digests (what was generated where /root/reference exists) and with the oracle run live on the same inputs."""
import gzip
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import lphash_amd as L  # noqa: E402
import orc  # noqa: E402
from real_fixture import CASES, HEAD, OUT, load_inputs, query_sets, unitig_records, valid_windows  # noqa: E402
from util import low_complexity_index_records, low_complexity_queries, pack_kmers  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXPECTED = json.load(open(os.path.join(OUT, "expected.json")))


def torch_dev():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a device; the product has no CPU fallback"
    return torch, torch.device("cuda", 0)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def run_records(h, torch, dev, recs, k, streaming):
    """lphq_query_records over one device-resident batch; LPHQ_INVALID stays in the slots of windows the streaming pass drops"""
    lens = np.array([len(r) for r in recs], dtype=np.int64)
    seq_off = np.concatenate([[0], np.cumsum(lens)])
    out_off = np.concatenate([[0], np.cumsum(np.maximum(lens - k + 1, 0))])
    flat = np.frombuffer(b"".join(recs), dtype=np.uint8)
    d_seq = torch.from_numpy(flat.copy()).to(dev)
    d_so, d_oo = torch.from_numpy(seq_off).to(dev), torch.from_numpy(out_off).to(dev)
    d_out = torch.full((int(out_off[-1]) + 1,), -7, dtype=torch.int64, device=dev)
    h.query_records(d_seq.data_ptr(), len(flat), d_so.data_ptr(), len(recs), d_out.data_ptr(), d_oo.data_ptr(), streaming)
    torch.cuda.synchronize()
    res = d_out.cpu().numpy().view(np.uint64)
    assert res[-1] == np.uint64(2**64 - 7)
    return res[:-1], out_off


@pytest.mark.parametrize("name", list(CASES))
def test_real_unitigs_build_on_gpu_and_every_query_path(tmp_path, name):
    """One real index per case: (1) lphq_build_device with extraction, sort, classification and pilot search on the GPU writes the file the oracle
    builder wrote in the build container (SHA-256; C1's committed file byte for byte) -- real minimizer skew, 0.3-2 % of the k-mers in the fallback
    function; (2) all unitigs through the streaming and the random kernels: a bijection onto [0, n), both == the oracle code for code;
    (3) the three real query sets (a genome with ~16 % members, another species with runs of N, reads with N) through lphq_query_records,
    lphq_query_contig, lphq_session_* and lphq_pack_kmers + lphq_query_packed: == the committed digests and heads, == the live oracle."""
    torch, dev = torch_dev()
    kind, k, m, bits, ust = CASES[name]
    e = EXPECTED["cases"][name]
    z = load_inputs()
    recs = unitig_records(z, ust)
    path = str(tmp_path / "gpu.lph")
    st = L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8, device=0)
    data = open(path, "rb").read()
    assert len(data) == e["lph_bytes"] and hashlib.sha256(data).hexdigest() == e["lph_sha256"], "GPU-built file on real unitigs differs from the oracle's"
    assert (st["num_kmers"], st["distinct_minimizers"], st["fallback_kmers"]) == (e["nkmers"], e["distinct_minimizers"], e["fallback_kmers"])
    if name == "se_k31_m20_b64_p":
        path = os.path.join(OUT, name + ".lph")  # from here on the committed file itself (written by the oracle builder where /root/reference exists)
        assert data == open(path, "rb").read()
    else:  # bounded memory on the device as well: sorted pieces on disk, the pilot search piece by piece
        p2 = str(tmp_path / "ext.lph")
        L.build_stream(lambda: iter([recs[i:i + 97] for i in range(0, len(recs), 97)]), k, m, p2, kind=kind, kmer_bits=bits, threads=8, device=0,
                       tmp_dir=str(tmp_path / "spill"), max_memory_bytes=4 << 20)
        assert open(p2, "rb").read() == data
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
    n = e["nkmers"]
    assert h.get_kmer_count() == n and h.get_minimizer_L0() == e["distinct_minimizers"] and h.info.fallback_kmers == e["fallback_kmers"]

    # (2) the index set itself
    want = np.concatenate([o(r, True) for r in recs if len(r) >= k])
    for streaming in (True, False):
        got, _ = run_records(h, torch, dev, recs, k, streaming)
        assert np.array_equal(got, want), (name, streaming)
    assert np.array_equal(np.sort(want), np.arange(n, dtype=np.uint64))  # the reference's --check (src/build.cpp:53-70)

    # (3) real queries
    heads = np.load(os.path.join(OUT, f"codes_{name}.npz"))
    for qname, qrecs in query_sets(z).items():
        qe = e["queries"][qname]
        rnd_live = [o(q, False) for q in qrecs]
        val = [valid_windows(q, k) for q in qrecs]
        got_r, off = run_records(h, torch, dev, qrecs, k, False)
        assert got_r.size == qe["windows"] and sha(got_r) == qe["random_sha256"] and np.array_equal(got_r[:HEAD], heads[qname])
        assert np.array_equal(got_r, np.concatenate(rnd_live))
        got_s, _ = run_records(h, torch, dev, qrecs, k, True)
        vflat = np.concatenate(val)
        assert (got_s[~vflat] == np.uint64(L.INVALID)).all()  # a window over a non-ACGT base: no code in streaming mode
        assert sha(got_s[vflat]) == qe["streaming_sha256"] and int(vflat.sum()) == qe["valid_windows"]
        assert np.array_equal(got_s[vflat], got_r[vflat])
        # one record per call (MPHF::operator()), the longest records of the set and a few others
        order = sorted(range(len(qrecs)), key=lambda i: -len(qrecs[i]))[:2] + list(range(2, min(len(qrecs), 40), 7))
        for i in order:
            assert np.array_equal(h(qrecs[i], streaming=False), rnd_live[i])
            assert np.array_equal(h(qrecs[i], streaming=True), rnd_live[i][val[i]])
        # batches in flight
        with h.session(depth=2) as s:
            half = max(1, len(qrecs) // 2)
            s.submit(qrecs[:half], True)
            s.submit(qrecs[half:], False)
            a, b = s.collect(), s.collect()
            assert all(np.array_equal(g, r[v]) for g, r, v in zip(a, rnd_live[:half], val[:half]))
            assert all(np.array_equal(g, r) for g, r in zip(b, rnd_live[half:]))
        # packed k-mers cut on the device from the same bases
        seq = qrecs[0][:60_000]
        pos = np.arange(0, len(seq) - k + 1, dtype=np.int64)
        d_s = torch.from_numpy(np.frombuffer(seq, dtype=np.uint8).copy()).to(dev)
        d_p = torch.from_numpy(pos).to(dev)
        d_k = torch.empty(len(pos) * (bits // 64), dtype=torch.int64, device=dev)
        h.pack_kmers(d_s.data_ptr(), d_p.data_ptr(), len(pos), d_k.data_ptr())
        d_o = torch.empty(len(pos), dtype=torch.int64, device=dev)
        h.query_packed(d_k.data_ptr(), len(pos), d_o.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_o.cpu().numpy().view(np.uint64), rnd_live[0][:len(pos)])
    h.close()


def write_fasta_gz(path, recs, width=70):
    with gzip.open(path, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">q%d real slice\n" % i)
            for j in range(0, len(r), width):
                f.write(r[j:j + width] + b"\n")


@pytest.mark.parametrize("name", ["se_k31_m20_b64_p", "se_k63_m17_b128_p", "se_k47_m20_b128_u"])
def test_real_data_through_the_cli(tmp_path, name):
    """`lphash build-p|build-u` on the unitigs as a FASTA.gz (GPU stages) and `lphash query-p|query-u` on the real query slices as FASTA.gz / FASTQ:
    the file's SHA-256, the 5-field CSV's k-mer count (src/query.cpp:83-86) and, through --output, the codes of the streaming pass == the fixture."""
    torch_dev()
    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    kind, k, m, bits, ust = CASES[name]
    e = EXPECTED["cases"][name]
    z = load_inputs()
    fa = tmp_path / "unitigs.fa.gz"
    write_fasta_gz(fa, unitig_records(z, ust), 80)
    lph = tmp_path / "x.lph"
    tool = "build-p" if kind == 0 else "build-u"
    r = subprocess.run([exe, tool, "-i", str(fa), "-k", str(k), "-m", str(m), "-o", str(lph), "--kmer-bits", str(bits), "-t", "8", "--check"], capture_output=True, text=True)
    assert r.returncode == 0 and "Everything is ok" in r.stderr, r.stderr[-2000:]
    assert hashlib.sha256(open(lph, "rb").read()).hexdigest() == e["lph_sha256"]
    qs = query_sets(z)
    files = {}
    for qname in ("salmonella", "ecoli"):
        files[qname] = tmp_path / (qname + ".fasta.gz")
        write_fasta_gz(files[qname], qs[qname])
    files["fastq"] = tmp_path / "reads.fastq"
    with open(files["fastq"], "wb") as f:
        for i, rd in enumerate(qs["fastq"]):
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, rd, b"I" * len(rd)))
    o = orc.Index(str(lph), kind, bits)
    for qname, qf in files.items():
        qe = e["queries"][qname]
        sink = tmp_path / (qname + ".u64")
        for extra in ([], ["--devices", "0,0,0", "--batch-mb", "1"]):
            r = subprocess.run([exe, "query-p" if kind == 0 else "query-u", "-i", str(lph), "-q", str(qf), "--kmer-bits", str(bits), "--output", str(sink)] + extra,
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            f5 = r.stdout.strip().split(",")
            assert len(f5) == 5 and int(f5[2]) == qe["valid_windows"], (qname, f5)
            got = np.fromfile(sink, dtype=np.uint64)
            assert sha(got) == qe["streaming_sha256"], (name, qname, extra)
        live = np.concatenate([o(q, False)[valid_windows(q, k)] for q in qs[qname]])
        assert np.array_equal(got, live)


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (63, 20, 128, 0), (31, 20, 64, 1), (63, 20, 128, 1), (31, 15, 128, 0), (47, 20, 128, 0), (63, 31, 128, 0), (31, 8, 64, 0)])
def test_low_complexity_ties_match_oracle(tmp_path, k, m, bits, kind):
    """Equal m-mers inside one window at the BASELINE (k, m) pairs: on uniform bases k31/m20 and k63/m20 meet a hash tie with probability ~1e-10
    per window, so the leftmost-minimum rule (mphf_utils.hpp:118-137: strict '<' from the left; partitioned_mphf.hpp:119-160: the ring-buffer
    rescan) was only ever exercised at small m.  Homopolymers, tandem repeats of period < k - m + 1, N runs of k bases and more: index AND queries;
    the GPU-built file == the oracle's, every path == the oracle, streaming == random on clean records."""
    torch, dev = torch_dev()
    recs = low_complexity_index_records(k, 900 + k + m)
    assert sum(1 for r in recs if len(set(r)) == 1) >= 3  # the homopolymers survived the de-duplication
    path, po = str(tmp_path / "gpu.lph"), str(tmp_path / "orc.lph")
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8, device=0)
    orc.build(recs, k, m, po, kind=kind, kmer_bits=bits)
    assert open(path, "rb").read() == open(po, "rb").read()
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
    n = o.nkmers
    members = np.concatenate([o(r, True) for r in recs])
    assert np.array_equal(np.sort(members), np.arange(n, dtype=np.uint64))
    qs = low_complexity_queries(recs, k, 5 + k)
    rnd = [o(q, False) if len(q) >= k else np.zeros(0, dtype=np.uint64) for q in qs]
    val = [valid_windows(q, k) for q in qs]
    for q, r, v in zip(qs, rnd, val):
        if v.size and v.all():
            assert np.array_equal(o(q, True), r)  # clean record: the reference's two branches agree, ties included
    want_r = np.concatenate(rnd)
    vflat = np.concatenate(val)
    for mis in (0, 3):
        batch = qs if mis == 0 else qs[::-1]
        wr = want_r if mis == 0 else np.concatenate(rnd[::-1])
        vf = vflat if mis == 0 else np.concatenate(val[::-1])
        got_r, _ = run_records(h, torch, dev, batch, k, False)
        assert np.array_equal(got_r, wr)
        got_s, _ = run_records(h, torch, dev, batch, k, True)
        assert np.array_equal(got_s[vf], wr[vf]) and (got_s[~vf] == np.uint64(L.INVALID)).all()
    for q, r, v in zip(qs, rnd, val):
        assert np.array_equal(h(q, streaming=False), r) and np.array_equal(h(q, streaming=True), r[v])
    with h.session(depth=1) as s:
        s.submit(qs, True)
        assert all(np.array_equal(g, r[v]) for g, r, v in zip(s.collect(), rnd, val))
    seq = b"".join(q for q in qs if b"N" not in q)
    pos = np.arange(0, len(seq) - k + 1, dtype=np.int64)
    pk = pack_kmers(seq, pos[:20_000], k, bits)
    d_k = torch.from_numpy(pk.view(np.int64)).to(dev)
    d_o = torch.empty(len(pk) // (bits // 64), dtype=torch.int64, device=dev)
    h.query_packed(d_k.data_ptr(), d_o.numel(), d_o.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_o.cpu().numpy().view(np.uint64), o.query_packed(pk))
    h.close()


@pytest.mark.parametrize("kind,k,m,bits,n_rec", [(0, 31, 20, 64, 600), (0, 63, 20, 128, 400), (1, 31, 20, 64, 300)])
def test_large_repeat_rich_index_properties(tmp_path, kind, k, m, bits, n_rec):
    """The fallback path at scale.  Uniform bases put 0.01-0.14 % of the k-mers behind colliding minimizers; a genome has repeats.  lphash_amd/synth.py's
    diverged-copy set (half the records are copies of the other half with one substitution per k - 1 bases: no repeated k-mer, most minimizers repeated) at
    30-60 M k-mers puts 30 % (k=31) / 63 % (k=63) of them there -- fallback functions of 10-25 M keys.  The index built with the GPU stages is byte for byte the
    host builder's; streaming and random passes over all records are bijections onto [0, n) and agree; sampled records and a packed batch equal the oracle."""
    from lphash_amd.synth import diverged_copy_records

    torch, dev = torch_dev()
    L_rec = 100_000 + k - 1
    flat, recs = diverged_copy_records(seed=5, n_records=n_rec, length=L_rec, k=k)
    pg, pc = str(tmp_path / "gpu.lph"), str(tmp_path / "cpu.lph")
    sg = L.build(recs, k, m, pg, kind=kind, kmer_bits=bits, threads=0, device=0)
    n = n_rec * 100_000
    assert sg["num_kmers"] == n and sg["fallback_kmers"] > 0.25 * n, sg
    if n_rec <= 400:  # (the sequential host search over tens of millions of keys takes its time: the two smaller sets)
        sc = L.build(recs, k, m, pc, kind=kind, kmer_bits=bits, threads=0)
        assert sc == sg and hashlib.sha256(open(pc, "rb").read()).hexdigest() == hashlib.sha256(open(pg, "rb").read()).hexdigest()
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(pg, kmer_bits=bits, device=0)
    assert h.get_kmer_count() == n and h.info.fallback_kmers == sg["fallback_kmers"]
    d_seq = torch.from_numpy(flat).to(dev)
    seq_off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * L_rec
    out_off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * 100_000
    outs = []
    for streaming in (True, False):
        d_out = torch.empty(n, dtype=torch.int64, device=dev)
        h.query_records(d_seq.data_ptr(), int(d_seq.numel()), seq_off.data_ptr(), n_rec, d_out.data_ptr(), out_off.data_ptr(), streaming)
        torch.cuda.synchronize()
        assert int(d_out.min()) == 0 and int(d_out.max()) == n - 1
        seen = torch.zeros(n, dtype=torch.uint8, device=dev)
        seen[d_out] = 1
        assert int(seen.sum(dtype=torch.int64)) == n
        del seen
        outs.append(d_out)
    assert torch.equal(outs[0], outs[1])
    o = orc.Index(pg, kind, bits)
    assert o.fallback_keys == sg["fallback_kmers"]
    for i in sorted({0, 1, n_rec // 2 - 1, n_rec // 2, n_rec // 2 + 1, n_rec - 1}):  # originals and their diverged copies
        assert np.array_equal(outs[0][i * 100_000:(i + 1) * 100_000].cpu().numpy().view(np.uint64), o(recs[i], True))
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    Q = 1 << 20
    win = torch.randint(0, n, (Q,), generator=g, device=dev, dtype=torch.int64)
    pos = (win // 100_000) * L_rec + win % 100_000
    d_k = torch.empty(Q * (bits // 64), dtype=torch.int64, device=dev)
    h.pack_kmers(d_seq.data_ptr(), pos.data_ptr(), Q, d_k.data_ptr())
    d_o = torch.empty(Q, dtype=torch.int64, device=dev)
    h.query_packed(d_k.data_ptr(), Q, d_o.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(d_o, outs[0][win])
    S = 1 << 18
    assert np.array_equal(d_o[:S].cpu().numpy().view(np.uint64), o.query_packed(d_k[: S * (bits // 64)].cpu().numpy().view(np.uint64)))
    h.close()


def test_chromosome_sized_record_through_every_path(tmp_path):
    """ONE record of 30 M bases (a chromosome arm; every other test cuts its input into records of <= 100 kbases): index built from it with the GPU stages, then the
    record through lphq_query_records (one record = 29,297 streaming tiles), a session, and the CLI with 1-MB batches (FASTA with 80-column lines: the record is far
    larger than a batch) and with three workers: a bijection onto [0, n), streaming == random, == the oracle on a 2 M-window slice, the CLI's count and codes."""
    from lphash_amd.synth import random_bases

    torch, dev = torch_dev()
    k, m, n_bases = 31, 20, int(os.environ.get("LPHQ_TEST_CHR_BASES", "30000000"))  # (a multiple of 80; 320,000,000 = larger than human chromosome 1: one-off runs)
    flat = random_bases(99, n_bases)
    rec = flat.tobytes()
    n = n_bases - k + 1
    path = str(tmp_path / "chr.lph")
    st = L.build([rec], k, m, path, kmer_bits=64, threads=0, device=0)
    assert st["num_kmers"] == n and st["num_records"] == 1
    h = L.Mphf.load(path, kmer_bits=64, device=0)
    d_seq = torch.from_numpy(flat).to(dev)
    so = torch.tensor([0, n_bases], dtype=torch.int64, device=dev)
    oo = torch.tensor([0, n], dtype=torch.int64, device=dev)
    outs = []
    for streaming in (True, False):
        d_out = torch.empty(n, dtype=torch.int64, device=dev)
        h.query_records(d_seq.data_ptr(), n_bases, so.data_ptr(), 1, d_out.data_ptr(), oo.data_ptr(), streaming)
        torch.cuda.synchronize()
        seen = torch.zeros(n, dtype=torch.uint8, device=dev)
        seen[d_out] = 1
        assert int(d_out.min()) == 0 and int(d_out.max()) == n - 1 and int(seen.sum(dtype=torch.int64)) == n
        outs.append(d_out)
    assert torch.equal(outs[0], outs[1])
    codes = outs[0].cpu().numpy().view(np.uint64)
    o = orc.Index(path, 0, 64)
    a = 13_000_003
    assert np.array_equal(o(rec[a:a + 2_000_000 + k - 1], True), codes[a:a + 2_000_000])
    with h.session(depth=2) as s:
        s.submit([rec], True)
        got = s.collect()
        assert len(got) == 1 and np.array_equal(got[0], codes)
    h.close()
    fa = tmp_path / "chr.fa"
    with open(fa, "wb") as f:
        f.write(b">chr_synthetic one record\n")
        f.write(np.concatenate([flat.reshape(-1, 80), np.full((n_bases // 80, 1), 10, dtype=np.uint8)], axis=1).tobytes())
    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    for extra in (["--batch-mb", "1"], ["--devices", "0,0,0"], []):
        sink = tmp_path / "codes.u64"
        r = subprocess.run([exe, "query-p", "-i", path, "-q", str(fa), "--kmer-bits", "64", "--output", str(sink)] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert int(r.stdout.split(",")[2]) == n
        assert np.array_equal(np.fromfile(sink, dtype=np.uint64), codes), extra
