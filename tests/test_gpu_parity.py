"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle -- bit-exact, integer work.

Sizes: seeded inputs the oracle finishes in seconds, the committed golden fixtures, and larger runs checked
through size-independent properties (bijection onto [0, n), streaming == random, order-independent checksum)."""
import glob
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import lphash_amd as L  # noqa: E402
import orc  # noqa: E402
from lphash_amd.synth import synthetic_records  # noqa: E402
from util import dedup_records, pack_kmers, random_records  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def torch_dev():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a device; the product has no CPU fallback"
    return torch, torch.device("cuda", 0)


def golden_cases():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.lph")))


def split(flat, off):
    return [flat[off[i]:off[i + 1]] for i in range(len(off) - 1)]


@pytest.mark.parametrize("name", golden_cases())
def test_golden_fixtures(name, golden_dir):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cls = L.Mphf if int(z["kind"]) == 0 else L.MphfAlt
    h = cls.load(os.path.join(golden_dir, name + ".lph"), kmer_bits=int(z["kmer_bits"]), device=0)
    queries = [bytes(q) for q in split(z["queries"], z["q_off"])]
    for q, s, r in zip(queries, split(z["streaming"], z["s_off"]), split(z["random"], z["r_off"])):
        assert np.array_equal(h(q, streaming=True), s)
        assert np.array_equal(h(q, streaming=False), r)


CONFIGS = [(31, 20, 64, 0), (31, 20, 128, 0), (31, 15, 128, 0), (63, 20, 128, 0), (63, 17, 128, 0), (31, 20, 64, 1), (15, 9, 64, 0), (31, 31, 64, 0),
           (21, 8, 64, 0), (21, 8, 64, 1), (47, 17, 128, 1), (33, 5, 128, 0), (9, 1, 64, 1), (63, 32, 128, 0), (20, 4, 64, 0),
           (3, 2, 64, 0), (5, 5, 64, 1), (4, 1, 64, 0), (2, 1, 128, 0), (35, 32, 128, 0)]


@pytest.mark.parametrize("k,m,bits,kind", CONFIGS)
def test_contig_queries_match_oracle(tmp_path, k, m, bits, kind):
    """MPHF::operator()(contig, len, streaming) for members, non-members, ragged and too-short records."""
    recs = dedup_records(random_records(k * 131 + m, 16, max(k - 2, 1), 6000 if k > 12 else k + 6), k)
    recs = [r for r in recs if len(r) >= k] or [b"ACGT"[:k] + b"G" * max(0, k - 4)]
    path = str(tmp_path / "t.lph")
    try:
        L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8)
    except L.LphqError as e:  # tiny key sets can be unplaceable for PTHash (builder.cpp: pilot cap); the reference would hang
        assert "no pilot below" in str(e)
        pytest.skip("degenerate tiny key set: PTHash cannot place a bucket")
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
    assert h.get_kmer_count() == o.nkmers
    queries = recs + random_records(7, 4, 1, 4000) + [b"", recs[0][: k - 1], recs[0][:k], recs[0].lower()]
    for q in queries:
        exp_r, exp_s = o(q, False), o(q, True)
        assert np.array_equal(h(q, streaming=False), exp_r)
        assert np.array_equal(h(q, streaming=True), exp_s)
        assert np.array_equal(exp_r, exp_s)  # ACGT-only input: the two reference passes agree


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (63, 20, 128, 0), (31, 20, 64, 1), (21, 8, 64, 0), (47, 17, 128, 1)])
def test_packed_batch_and_pack_kernel(tmp_path, k, m, bits, kind):
    torch, dev = torch_dev()
    recs = random_records(k + 17 * m, 6, 500, 5000)
    path = str(tmp_path / "t.lph")
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8)
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
    seq = b"".join(recs) + random_records(3, 1, 3000, 3000)[0]  # windows across record joins are non-members
    pos = np.arange(0, len(seq) - k + 1, dtype=np.int64)
    pk = pack_kmers(seq, pos, k, bits)
    d_s = torch.from_numpy(np.frombuffer(seq, dtype=np.uint8).copy()).to(dev)
    d_p = torch.from_numpy(pos).to(dev)
    d_k = torch.empty(len(pk), dtype=torch.int64, device=dev)
    h.pack_kmers(d_s.data_ptr(), d_p.data_ptr(), len(pos), d_k.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_k.cpu().numpy().view(np.uint64), pk)
    d_o = torch.empty(len(pos), dtype=torch.int64, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        h.query_packed(d_k.data_ptr(), len(pos), d_o.data_ptr(), s.cuda_stream)
    s.synchronize()
    assert np.array_equal(d_o.cpu().numpy().view(np.uint64), o.query_packed(pk))
    h.query_packed(d_k.data_ptr(), 0, d_o.data_ptr())  # empty batch is a no-op


def run_records(h, torch, dev, recs, k, streaming, misalign=0):
    lens = np.array([len(r) for r in recs], dtype=np.int64)
    seq_off = np.concatenate([[0], np.cumsum(lens)])
    nwin = np.maximum(lens - k + 1, 0)
    out_off = np.concatenate([[0], np.cumsum(nwin)])
    flat = np.frombuffer(b"".join(recs), dtype=np.uint8)
    buf = torch.zeros(len(flat) + 64, dtype=torch.uint8, device=dev)
    buf[misalign:misalign + len(flat)] = torch.from_numpy(flat.copy()).to(dev)
    d_so, d_oo = torch.from_numpy(seq_off).to(dev), torch.from_numpy(out_off).to(dev)
    d_out = torch.full((int(out_off[-1]) + 1,), -7, dtype=torch.int64, device=dev)
    h.query_records(buf.data_ptr() + misalign, len(flat), d_so.data_ptr(), len(recs), d_out.data_ptr(), d_oo.data_ptr(), streaming)
    torch.cuda.synchronize()
    res = d_out.cpu().numpy().view(np.uint64)
    assert res[-1] == np.uint64(2**64 - 7)  # nothing written past the last slot
    return [res[out_off[i]:out_off[i + 1]] for i in range(len(recs))]


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (63, 20, 128, 0), (31, 20, 64, 1), (21, 8, 64, 0)])
def test_record_batches_ragged_empty_and_misaligned(tmp_path, k, m, bits, kind):
    """Batches of many records on device: empty, shorter-than-k and tile-straddling records, any pointer alignment."""
    torch, dev = torch_dev()
    recs = random_records(99 + k, 40, 400, 3000)
    path = str(tmp_path / "t.lph")
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8)
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
    batch = [b"", recs[0], recs[1][: k - 1], b"", recs[2][:k], recs[3]] + recs[4:] + random_records(5, 30, 1, 200) + [b""]
    batch += random_records(11, 150, 0, 5) + [recs[0][:k + 2]] + random_records(12, 60, k, k + 20)  # > 64 and ~20 records per tile
    exp = [o(r, False) for r in batch]
    for streaming in (False, True):
        for mis in (0, 1, 7, 15):
            got = run_records(h, torch, dev, batch, k, streaming, mis)
            for g, e in zip(got, exp):
                assert np.array_equal(g, e)


def test_non_acgt_bases(tmp_path):
    """Defined behaviour outside the reference's clean-input contract (SURVEY section 8a note 4): random mode reads a
    non-ACGT base as 'A' (mphf_utils.hpp:108) and still emits; streaming emits one code per fully valid window."""
    k, m = 31, 20
    recs = random_records(42, 6, 500, 2000)
    path = str(tmp_path / "t.lph")
    L.build(recs, k, m, path, kmer_bits=64, threads=4)
    o = orc.Index(path, 0, 64)
    h = L.Mphf.load(path, kmer_bits=64, device=0)
    rng = np.random.default_rng(0)
    for r in recs:
        q = bytearray(r)
        for p in rng.integers(0, len(q), 5):
            q[p] = ord("N")
        q[0] = ord("R")
        q[-1] = ord("-")
        q[len(q) // 2] = ord("u")  # U/u are valid (seq_nt4_table maps them to 3)
        q = bytes(q)
        exp_r = o(q, False)
        assert np.array_equal(h(q, streaming=False), exp_r)
        bad = np.array([c not in b"ACGTUacgtu" for c in q])
        csum = np.concatenate([[0], np.cumsum(bad)])
        valid = (csum[k:] - csum[:-k]) == 0
        assert np.array_equal(h(q, streaming=True), exp_r[valid])


def test_device_image_handle_and_concurrent_callers(tmp_path):
    """lphq_open_device_image over a copy of the image (what a non-root rank does after the RCCL broadcast) and
    several host threads querying one index at once."""
    torch, dev = torch_dev()
    recs = random_records(11, 10, 1000, 4000)
    path = str(tmp_path / "t.lph")
    L.build(recs, 31, 20, path, kmer_bits=64, threads=4)
    o = orc.Index(path, 0, 64)
    h0 = L.Mphf.load(path, kmer_bits=64, device=0)
    t = torch.from_numpy(h0.image()).to(dev)
    h = L.Mphf.from_device_image(t.data_ptr(), t.numel(), 0)
    assert h.get_kmer_count() == h0.get_kmer_count() and h.info.device == 0
    exp = [o(r, False) for r in recs]
    errs = []

    def work(i):
        try:
            for _ in range(5):
                for streaming in (True, False):
                    if not np.array_equal(h(recs[i], streaming=streaming), exp[i]):
                        errs.append(i)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(recs))]
    [t_.start() for t_ in ths]
    [t_.join() for t_ in ths]
    assert not errs
    # an adopted image is only trusted as far as its header proves itself: the kernels divide by the reciprocal fields (dn / inv / shift) of the
    # table-size divisor alone, so a copy whose reciprocal does not belong to its divisor is refused instead of hashing out of bounds
    img = h0.image().copy()
    ts = np.frombuffer(np.uint64(h0.info.minimizer_table_size).tobytes(), dtype=np.uint8)
    hits = [i for i in range(0, 1024, 4) if np.array_equal(img[i:i + 8], ts)]
    assert len(hits) >= 2  # phf_desc.table_size, then div_table.d64 (fields d, dn, inv, shift before it)
    bad = img.copy()
    bad[hits[1] - 8] ^= 0x10  # div_table.inv
    tb = torch.from_numpy(bad).to(dev)
    with pytest.raises(L.LphqError):
        L.Mphf.from_device_image(tb.data_ptr(), tb.numel(), 0)
    assert L.lib.lphq_abi_version() >= 2


def test_attach_devices_and_stream_ordered_count(tmp_path):
    """lphq_attach_devices / lphq_attached_devices (replicas, primary first; attaching twice is a no-op) and lphq_count_valid on the
    caller's NON-BLOCKING stream: the count must see the codes the query wrote on that stream (it used to run on the null stream)."""
    torch, dev = torch_dev()
    k = 31
    recs = random_records(21, 12, 2000, 9000)
    path = str(tmp_path / "t.lph")
    L.build(recs, k, 20, path, kmer_bits=64, threads=4)
    h = L.Mphf.load(path, kmer_bits=64)
    with pytest.raises(L.LphqError):
        h.query_packed(0, 0, 0)  # not attached: no CPU fallback
    h.attach_devices([0, 0])
    assert h.attached_devices() == [0] and h.info.device == 0
    h.attach(0)
    assert h.attached_devices() == [0]
    with pytest.raises(L.LphqError):
        h.attach_devices([torch.cuda.device_count() + 3])
    q = bytearray(b"".join(recs))
    for p in range(100, len(q), 997):
        q[p] = ord("N")
    flat = np.frombuffer(bytes(q), dtype=np.uint8)
    lens = np.array([len(r) for r in recs], dtype=np.int64)
    seq_off = np.concatenate([[0], np.cumsum(lens)])
    out_off = np.concatenate([[0], np.cumsum(lens - k + 1)])
    d_seq = torch.from_numpy(flat.copy()).to(dev)
    d_so, d_oo = torch.from_numpy(seq_off).to(dev), torch.from_numpy(out_off).to(dev)
    n_out = int(out_off[-1])
    s = torch.cuda.Stream()  # non-blocking with respect to the null stream
    d_out = torch.full((n_out,), -1, dtype=torch.int64, device=dev)  # all LPHQ_INVALID before the query
    torch.cuda.synchronize()
    for _ in range(5):
        d_out.fill_(-1)
        torch.cuda.synchronize()
        h.query_records(d_seq.data_ptr(), len(flat), d_so.data_ptr(), len(recs), d_out.data_ptr(), d_oo.data_ptr(), True, s.cuda_stream)
        got = h.count_valid(d_out.data_ptr(), n_out, s.cuda_stream)
        bad = np.array([c not in b"ACGTUacgtu" for c in bytes(q)])
        want = 0
        for i in range(len(recs)):
            b = bad[seq_off[i]:seq_off[i + 1]]
            cs = np.concatenate([[0], np.cumsum(b)])
            want += int(((cs[k:] - cs[:-k]) == 0).sum())
        assert got == want
    assert h.count_valid(d_out.data_ptr(), 0) == 0
    # result consumer: payload[code] without leaving the device; LPHQ_INVALID (windows with an N) -> `missing`
    for dt, tdt in ((np.uint32, torch.int32), (np.uint64, torch.int64), (np.uint8, torch.uint8)):
        table = np.random.default_rng(1).integers(0, np.iinfo(dt).max, h.get_kmer_count(), dtype=dt)
        d_t = torch.from_numpy(table.view(np.int32 if dt == np.uint32 else (np.int64 if dt == np.uint64 else np.uint8))).to(dev)
        d_p = torch.empty(n_out, dtype=tdt, device=dev)
        h.gather_payload(d_out.data_ptr(), n_out, d_t.data_ptr(), table.itemsize, d_p.data_ptr(), missing=7, stream=s.cuda_stream)
        s.synchronize()
        codes = d_out.cpu().numpy().view(np.uint64)
        want = np.where(codes < h.get_kmer_count(), table[np.minimum(codes, h.get_kmer_count() - 1).astype(np.int64)], dt(7))
        assert np.array_equal(d_p.cpu().numpy().view(dt), want)
    with pytest.raises(L.LphqError):
        h.gather_payload(d_out.data_ptr(), n_out, d_t.data_ptr(), 3, d_p.data_ptr())


@pytest.mark.parametrize("kind,k,m,bits,n_rec", [(0, 31, 20, 64, 1000), (1, 31, 20, 64, 1000), (0, 63, 20, 128, 100), (0, 63, 20, 128, 20_000)])
def test_large_index_properties(tmp_path, kind, k, m, bits, n_rec):
    """BASELINE.json's configurations at FULL size -- C2 / C3 (partitioned, k=31 m=20, 100 M k-mers), C5 (unpartitioned, same set) and the
    per-GPU work of C4 (k=63, 128-bit k-mers, 2 B k-mers; index built with the GPU pilot search, whose bytes equal the CPU build's:
    test_zz_gpu_next.py) -- plus k=63 at 10 M.  Too many k-mers for a full oracle pass in a test: the device codes are a bijection onto
    [0, n) in streaming AND random mode, both modes agree, sampled records equal the oracle, and a packed random batch equals the
    codes of the windows it was cut from."""
    torch, dev = torch_dev()
    L_rec = 100_000 + k - 1
    flat, recs = synthetic_records(seed=3 if n_rec == 100 else 1, n_records=n_rec, length=L_rec)
    path = str(tmp_path / "big.lph")
    # sequential CPU pilot search (the query path is what is under test) except at 2 B k-mers, where it would take minutes
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=0, device=0 if n_rec > 1000 else None)
    # the file just built against the ORACLE builder's on the same set (tests/golden/full_size_digests.json, tools/full_size_digests.py: 20 minutes of one
    # core at 2 B k-mers, so it is derived once in the build container): CPU pilot search at 100 M, GPU extraction + pilot search at 2 B -- same bytes
    import hashlib
    import json

    def sha256_file(p_):
        hs = hashlib.sha256()
        with open(p_, "rb") as f_:
            for blk in iter(lambda: f_.read(1 << 24), b""):
                hs.update(blk)
        return hs.hexdigest()

    pinned = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size_digests.json"))).get(f"{'pu'[kind]}_k{k}_m{m}_b{bits}_seed1_{n_rec}x{L_rec}")
    assert (pinned is not None) == (n_rec >= 1000)
    if pinned is not None:
        assert os.path.getsize(path) == pinned["bytes"] and sha256_file(path) == pinned["sha256"], "full-size build differs from the oracle builder's file"
    if n_rec > 1000:  # ... and with a memory limit of 1 GB (sorted pieces on disk, the pilot search piece by piece on the device): the same file
        p2 = str(tmp_path / "ext.lph")
        L.build_stream(lambda: (recs[i:i + 250] for i in range(0, n_rec, 250)), k, m, p2, kind=kind, kmer_bits=bits, threads=0, device=0,
                       tmp_dir=str(tmp_path / "spill"), max_memory_bytes=1 << 30)
        assert sha256_file(p2) == pinned["sha256"]
        os.remove(p2)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
    n = n_rec * 100_000
    assert h.get_kmer_count() == n
    d_seq = torch.from_numpy(flat).to(dev)
    seq_off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * L_rec
    out_off = torch.arange(0, n_rec + 1, dtype=torch.int64, device=dev) * 100_000
    outs = []
    for streaming in (True, False):
        d_out = torch.empty(n, dtype=torch.int64, device=dev)
        h.query_records(d_seq.data_ptr(), int(d_seq.numel()), seq_off.data_ptr(), n_rec, d_out.data_ptr(), out_off.data_ptr(), streaming)
        torch.cuda.synchronize()
        assert int(d_out.min()) == 0 and int(d_out.max()) == n - 1
        # bijection: every slot exactly once (n codes inside [0, n) that mark n distinct slots)
        seen = torch.zeros(n, dtype=torch.uint8, device=dev)
        seen[d_out] = 1
        assert int(seen.sum(dtype=torch.int64)) == n
        del seen
        outs.append(d_out)
    assert torch.equal(outs[0], outs[1])
    # round 6, at the same full size: the streaming pass with its result as runs (lphq_query_records_runs) expands to exactly these codes, and the pass fused with
    # the payload gather (lphq_query_records_gather) yields table[code] for every window
    W = k - m + 1
    per_arena = int(n * (2.0 / (W + 1) + 1.0 / 400.0) * 1.15 / 32) + 4096
    d_runs = torch.empty(32 * per_arena * 2, dtype=torch.int64, device=dev)
    d_xc = torch.empty(32 * (per_arena // 8 + 4096), dtype=torch.int64, device=dev)
    d_counts = torch.empty(32 * 16, dtype=torch.int64, device=dev)
    h.query_records_runs(d_seq.data_ptr(), int(d_seq.numel()), seq_off.data_ptr(), n_rec, out_off.data_ptr(), d_runs.data_ptr(), 32 * per_arena, d_xc.data_ptr(),
                         32 * (per_arena // 8 + 4096), d_counts.data_ptr())
    d_exp = torch.empty(n, dtype=torch.int64, device=dev)
    d_st = torch.ones(1, dtype=torch.int64, device=dev)
    h.expand_runs(d_runs.data_ptr(), 32 * per_arena, d_xc.data_ptr(), 32 * (per_arena // 8 + 4096), d_counts.data_ptr(), d_exp.data_ptr(), n, d_st.data_ptr())
    torch.cuda.synchronize()
    cnt = d_counts.cpu().numpy().reshape(32, 16)
    assert int(d_st.item()) == 0 and torch.equal(d_exp, outs[0])
    assert 0.8 * 2 * n / (W + 1) < int(cnt[:, 0].sum()) < 1.2 * (2 * n / (W + 1) + n / 400) and int(cnt[:, 1].sum()) == h.info.fallback_kmers  # (members: every fallback k-mer once)
    del d_exp, d_runs, d_xc
    table = (torch.arange(n, dtype=torch.int64, device=dev) * 2654435761 >> 11).to(torch.uint8)
    d_pay = torch.empty(n, dtype=torch.uint8, device=dev)
    h.query_records_gather(d_seq.data_ptr(), int(d_seq.numel()), seq_off.data_ptr(), n_rec, out_off.data_ptr(), table.data_ptr(), 1, d_pay.data_ptr(), missing=255)
    torch.cuda.synchronize()
    for a_ in range(0, n, 1 << 28):
        assert torch.equal(d_pay[a_:a_ + (1 << 28)], table[outs[0][a_:a_ + (1 << 28)]])
    del d_pay, table
    o = orc.Index(path, kind, bits)
    for i in sorted({0, 1, 37, n_rec // 3, n_rec // 2, n_rec - 2, n_rec - 1}):  # 700 K windows against the oracle's streaming branch
        assert np.array_equal(outs[0][i * 100_000:(i + 1) * 100_000].cpu().numpy().view(np.uint64), o(recs[i], True))
    # packed random batch with repeats: the codes of the windows the k-mers were cut from, and the oracle's on a slice of it
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    Q = 1 << 22
    win = torch.randint(0, n, (Q,), generator=g, device=dev, dtype=torch.int64)
    pos = (win // 100_000) * L_rec + win % 100_000
    d_k = torch.empty(Q * (bits // 64), dtype=torch.int64, device=dev)
    h.pack_kmers(d_seq.data_ptr(), pos.data_ptr(), Q, d_k.data_ptr())
    d_o = torch.empty(Q, dtype=torch.int64, device=dev)
    h.query_packed(d_k.data_ptr(), Q, d_o.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(d_o, outs[0][win])
    # 2^20 probes against the oracle's random branch: at C4's fallback rate (0.14 % of the k-mers sit under colliding minimizers) that is ~1,400 lookups
    # through the fallback function, where round 3's 2^16 probes met ~90
    S = 1 << 20
    assert np.array_equal(d_o[:S].cpu().numpy().view(np.uint64), o.query_packed(d_k[: S * (bits // 64)].cpu().numpy().view(np.uint64)))
    expect_fallback = S * h.info.fallback_kmers / n
    assert n_rec <= 1000 or expect_fallback > 500, expect_fallback


def test_cli_query_end_to_end(tmp_path):
    """`lphash build-p --check` then `lphash query-p` on a FASTA.gz: 5-field CSV of src/query.cpp:83-86, exit codes."""
    import gzip
    import subprocess

    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    recs = random_records(5, 20, 50, 5000)
    fa = tmp_path / "idx.fa.gz"
    with gzip.open(fa, "wb") as f:
        for i, rec in enumerate(recs):
            f.write(b">r%d\n%s\n" % (i, rec))
    fq = tmp_path / "q.fastq"
    with open(fq, "wb") as f:
        for i, rec in enumerate(recs[:7]):
            f.write(b"@q%d\n%s\n+\n%s\n" % (i, rec, b"I" * len(rec)))
    lph = tmp_path / "x.lph"
    r = subprocess.run([exe, "build-p", "-i", str(fa), "-k", "31", "-m", "20", "-o", str(lph), "--kmer-bits", "64", "-t", "8", "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "Everything is ok" in r.stderr, r.stderr
    for q, nrec in ((fa, recs), (fq, recs[:7])):
        r = subprocess.run([exe, "query-p", "-i", str(lph), "-q", str(q), "--kmer-bits", "64"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        f = r.stdout.strip().split(",")
        assert len(f) == 5 and f[0] == str(q) and f[1] == str(lph)
        assert int(f[2]) == sum(max(0, len(x) - 30) for x in nrec) and float(f[3]) > 0 and float(f[4]) > 0
    for extra in (["--gpus", "1"], ["--pageable"], ["--gpus", "1", "--batch-mb", "1", "--threads", "3"]):  # sharded / unpinned / many small batches
        r = subprocess.run([exe, "query-p", "-i", str(lph), "-q", str(fa), "--kmer-bits", "64"] + extra, capture_output=True, text=True)
        assert r.returncode == 0 and int(r.stdout.split(",")[2]) == sum(max(0, len(x) - 30) for x in recs), r.stderr
    r = subprocess.run([exe, "query-p", "-i", str(lph), "-q", str(fa), "--kmer-bits", "64", "--gpus", "64"], capture_output=True, text=True)
    assert r.returncode == 2 and "device" in r.stderr  # more GPUs than the node has
    r = subprocess.run([exe, "query-p", "-i", str(lph), "-q", str(tmp_path / "missing.fa"), "--kmer-bits", "64"], capture_output=True, text=True)
    assert r.returncode == 2 and "Unable to open the input file" in r.stderr
    # result sink: --output keeps the codes of the streaming pass (the reference discards them, src/query.cpp:54)
    sink = tmp_path / "codes.u64"
    r = subprocess.run([exe, "query-p", "-i", str(lph), "-q", str(fa), "--kmer-bits", "64", "--output", str(sink)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    o = orc.Index(str(lph), 0, 64)
    assert np.array_equal(np.fromfile(sink, dtype=np.uint64), np.concatenate([o(x, True) for x in recs]))
    # a query-u file through query-p is rejected, not mis-hashed
    lpu = tmp_path / "u.lph"
    subprocess.run([exe, "build-u", "-i", str(fa), "-k", "31", "-m", "20", "-o", str(lpu), "--kmer-bits", "64"], check=True, capture_output=True)
    r = subprocess.run([exe, "query-p", "-i", str(lpu), "-q", str(fa), "--kmer-bits", "64"], capture_output=True, text=True)
    assert r.returncode == 2
    r = subprocess.run([exe, "query-u", "-i", str(lpu), "-q", str(fa), "--kmer-bits", "64"], capture_output=True, text=True)
    assert r.returncode == 0 and int(r.stdout.split(",")[2]) == sum(max(0, len(x) - 30) for x in recs)


def test_cli_many_batches_keep_record_order(tmp_path):
    """`lphash query-p --batch-mb 1 --output`: several batches in flight (two device slots, asynchronous copies from the pinned
    parse arenas); the sink must hold the codes of the streaming pass in record order, the counts of both passes must agree."""
    import subprocess

    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    _, recs = synthetic_records(seed=77, n_records=45, length=70_000)
    fa = tmp_path / "idx.fa"
    with open(fa, "wb") as f:
        for i, rec in enumerate(recs):
            f.write(b">r%d\n" % i)
            for j in range(0, len(rec), 80):
                f.write(rec[j:j + 80] + b"\n")
    lph = tmp_path / "x.lph"
    L.build(recs, 31, 20, str(lph), kmer_bits=64, threads=8)
    o = orc.Index(str(lph), 0, 64)
    want = np.concatenate([o(x, True) for x in recs])
    # --devices 0,0,0: three workers (streams, arenas, buffer slots) on the one GPU of the box -- the N-worker sharding of every batch
    for extra in ([], ["--gpus", "1", "--threads", "5"], ["--pageable"], ["--devices", "0,0,0"], ["--devices", "0,0", "--pageable", "--threads", "3"]):
        sink = tmp_path / "codes.u64"
        r = subprocess.run([exe, "query-p", "-i", str(lph), "-q", str(fa), "--kmer-bits", "64", "--batch-mb", "1", "--output", str(sink)] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert int(r.stdout.split(",")[2]) == len(want)
        assert np.array_equal(np.fromfile(sink, dtype=np.uint64), want)


def test_peer_copied_replica_answers_like_the_uploaded_one(tmp_path):
    """The replica path of lphq_attach_devices on a box with ONE GPU (LPHQ_TEST_SELF_PEER, read once per process: hence the child): the device is
    attached twice, the second replica is a peer copy of the first (hipMemcpyPeerAsync on a stream of the destination, waited for at the end of the
    call) and every query of the process is answered from that copy -- records, packed k-mers, one-record calls and a session, against the oracle."""
    import subprocess
    import sys

    torch_dev()
    script = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import lphash_amd as L, orc
from util import random_records, dedup_records, pack_kmers
k, m = 31, 20
recs = dedup_records(random_records(4711, 20, 500, 20000), k)
path = os.path.join(sys.argv[2], "t.lph")
L.build(recs, k, m, path, kmer_bits=64, threads=4)
h = L.Mphf.load(path, kmer_bits=64)
try:  # LPHQ_TEST_FAIL_PEER_COPY=1: the first peer copy "fails" after its wait -- the half-made replica must not stay attached (ADVICE r04)
    h.attach_devices([0, 0])
    raise SystemExit("the injected copy failure was not reported")
except L.LphqError:
    pass
assert h.replica_info() == [(0, 0)], h.replica_info()
assert np.array_equal(np.asarray(h(recs[0], streaming=True)), np.asarray(orc.Index(path, 0, 64)(recs[0], True)))  # the uploaded replica still answers
h.attach_devices([0, 0])  # the retry makes the copy (it used to return at once: "device already attached")
info = h.replica_info()
assert [d for d, _ in info] == [0, 0] and info[0][1] == 0 and info[1][1] in (1, 2), info
o = orc.Index(path, 0, 64)
dev = torch.device("cuda:0")
want = [np.asarray(o(r, True), dtype=np.uint64) for r in recs]
for r, w in zip(recs[:6], want):
    assert np.array_equal(np.asarray(h(r, streaming=True), dtype=np.uint64), w) and np.array_equal(np.asarray(h(r, streaming=False), dtype=np.uint64), w)
flat = np.frombuffer(b"".join(recs), dtype=np.uint8)
lens = np.array([len(r) for r in recs], dtype=np.int64)
so, oo = np.concatenate([[0], np.cumsum(lens)]), np.concatenate([[0], np.cumsum(lens - k + 1)])
d_seq, d_so, d_oo = torch.from_numpy(flat.copy()).to(dev), torch.from_numpy(so).to(dev), torch.from_numpy(oo).to(dev)
for streaming in (True, False):
    d_out = torch.zeros(int(oo[-1]), dtype=torch.int64, device=dev)
    h.query_records(d_seq.data_ptr(), len(flat), d_so.data_ptr(), len(recs), d_out.data_ptr(), d_oo.data_ptr(), streaming)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), np.concatenate(want)), streaming
s = h.session(depth=2)
s.submit(recs, streaming=True)
got = s.collect()
s.close()
assert len(got) == len(want) and all(np.array_equal(np.asarray(g, dtype=np.uint64), w) for g, w in zip(got, want))
print("replica ok", info)
"""
    env = dict(os.environ, LPHQ_TEST_SELF_PEER="1", LPHQ_TEST_FAIL_PEER_COPY="1")
    r = subprocess.run([sys.executable, "-c", script, ROOT, str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "replica ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_cli_eight_workers_ragged_and_idle_shards(tmp_path):
    """`--devices 0,0,0,0,0,0,0,0`: the sharding `lphash query-* --gpus 8` does on an 8-GPU node, with eight workers on the one GPU here.  Ragged
    records (many shorter than k, so some shards hold bases but no k-mer) and files with fewer records than workers (idle shards): codes and
    counts must equal the single-worker run and the oracle, in record order."""
    import subprocess

    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    _, recs = synthetic_records(seed=78, n_records=12, length=50_000)
    lph = tmp_path / "x.lph"
    L.build(recs, 31, 20, str(lph), kmer_bits=64, threads=8)
    o = orc.Index(str(lph), 0, 64)
    rng = np.random.default_rng(9)
    big = b"".join(recs)
    ragged = []
    for _ in range(400):
        n = int(rng.choice([3, 17, 30, 31, 32, 64, 500, 4000]))
        at = int(rng.integers(0, len(big) - n))
        ragged.append(big[at:at + n])
    files = {"ragged": ragged, "two": [recs[0][:9000], recs[1][:40]], "short_only": [recs[2][:30], recs[3][:12], recs[4][:1]]}
    for name, rs in files.items():
        fa = tmp_path / (name + ".fa")
        with open(fa, "wb") as f:
            for i, rec in enumerate(rs):
                f.write(b">r%d\n%s\n" % (i, rec))
        parts = [o(x, True) for x in rs if len(x) >= 31]
        want = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)
        for extra in (["--device", "0"], ["--devices", "0,0,0,0,0,0,0,0"], ["--devices", "0,0,0", "--batch-mb", "1", "--threads", "2"]):
            sink = tmp_path / (name + ".u64")
            r = subprocess.run([exe, "query-p", "-i", str(lph), "-q", str(fa), "--kmer-bits", "64", "--output", str(sink)] + extra, capture_output=True, text=True)
            assert r.returncode == 0, (name, extra, r.stderr)
            f5 = r.stdout.strip().split(",")
            assert int(f5[2]) == len(want), (name, extra, r.stdout)
            assert np.array_equal(np.fromfile(sink, dtype=np.uint64), want), (name, extra)


def test_empty_and_tiny_indexes(tmp_path):
    """Degenerate functions: no k-mer at all, a single k-mer, a single minimizer -- must load, attach and answer without
    touching memory out of bounds (codes for non-members are unspecified but must equal the oracle's)."""
    k, m = 31, 20
    cases = {"empty": [], "too_short": [b"ACGTACGT"], "one_kmer": [random_records(1, 1, 31, 31)[0]], "one_record": random_records(2, 1, 40, 40)}
    probes = random_records(9, 3, 31, 200)
    for name, recs in cases.items():
        for kind, cls in ((0, L.Mphf), (1, L.MphfAlt)):
            path = str(tmp_path / f"{name}_{kind}.lph")
            L.build(recs, k, m, path, kind=kind, kmer_bits=64, threads=2)
            ref = str(tmp_path / f"{name}_{kind}.ref.lph")
            orc.build(recs, k, m, ref, kind=kind, kmer_bits=64)
            assert open(path, "rb").read() == open(ref, "rb").read()
            h = cls.load(path, kmer_bits=64, device=0)
            nk = sum(max(0, len(r) - k + 1) for r in recs)
            assert h.get_kmer_count() == nk
            o = orc.Index(path, kind, 64)
            for r in recs:
                for streaming in (True, False):
                    assert np.array_equal(h(r, streaming=streaming), o(r, False))
            if nk:  # non-member probes on a non-empty function: same (meaningless) codes as the oracle
                for q in probes:
                    assert np.array_equal(h(q, streaming=False), o(q, False))
                    assert np.array_equal(h(q, streaming=True), o(q, False))
            else:   # nothing is indexed: only the number of codes is defined
                for q in probes:
                    assert len(h(q, streaming=True)) == len(q) - k + 1


# ---- files written by a real reference build (tests/golden/ref/README.md): skipped until one is supplied ----
def test_reference_built_files_gpu_codes():
    """For every case directory under tests/golden/ref that holds queries and code dumps: the gfx950 kernels, through the C ABI, answer the
    REFERENCE's codes in streaming and in random mode."""
    import ref_intake

    found = [ref_intake.Case(d) for d in ref_intake.cases()]
    found = [c for c in found if c.queries and c.codes]
    if not found:
        pytest.skip("no reference-built file supplied (tests/golden/ref/README.md has the recipe)")
    torch_dev()
    for c in found:
        assert ref_intake.check_case_gpu(c, L), c.dir


@pytest.mark.parametrize("kind,k,m,bits", [("p", 31, 20, 64), ("u", 31, 20, 64), ("p", 63, 17, 128)])
def test_reference_intake_dry_run_gpu(tmp_path, kind, k, m, bits):
    """The GPU half of the intake on an oracle-written case in a temporary directory (layout of README.md): both code dumps are reproduced by the
    kernels; a corrupted dump fails."""
    import ref_intake

    torch_dev()
    d = ref_intake.write_oracle_case(str(tmp_path / "case"), orc, kind=kind, k=k, m=m, kmer_bits=bits)
    assert ref_intake.check_case_gpu(ref_intake.Case(d), L) == ["gpu-codes-streaming", "gpu-codes-random"]
    codes = np.fromfile(os.path.join(d, "codes_streaming.u64"), dtype="<u8")
    codes[3] += 1
    codes.tofile(os.path.join(d, "codes_streaming.u64"))
    with pytest.raises(AssertionError, match="GPU codes differ"):
        ref_intake.check_case_gpu(ref_intake.Case(d), L)


@pytest.mark.parametrize("name", ["p_k31_m20_b64", "p_k63_m20_b128", "u_k31_m20_b64", "p_k21_m8_b64_fallback"])
def test_session_batches_match_oracle(name):
    """lphq_session_* (ABI 3): MPHF::operator() over many records per call with batches in flight.  Every record's codes == the oracle's in both
    modes, for batches that mix long records, records shorter than k, empty records, records holding non-ACGT bytes (their windows are dropped in
    streaming mode, read as 'A' in random mode) and lower case; batches are collected in submission order; the session refuses a third batch while two
    are in flight and a collect when none is."""
    golden = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(golden, name + ".npz"))
    kind, bits = int(z["kind"]), int(z["kmer_bits"])
    path = os.path.join(golden, name + ".lph")
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
    k = h.k
    base = [bytes(z["queries"][z["q_off"][i]:z["q_off"][i + 1]]) for i in range(len(z["q_off"]) - 1)]
    rng = np.random.default_rng(5)
    long_rec = b"".join(base)[:50_000]
    dirty = bytearray(long_rec[:5000])
    for pos in rng.integers(0, len(dirty), 12):
        dirty[int(pos)] = ord("N")
    dirty[100:130] = b"n" * 30
    batches = [
        base,
        [long_rec, b"", b"ACGT", bytes(dirty), long_rec[:k], long_rec[:k - 1], long_rec.lower()[:3000], b"NNNN" * 40],
        [bytes(dirty)],
        [b"", b""],
        [long_rec[i:i + 400] for i in range(0, 40_000, 400)],
    ]

    def want(rec, streaming):
        if len(rec) < k:
            return np.zeros(0, dtype=np.uint64)
        exp = o(rec, False)  # non-ACGT read as 'A' (mphf_utils.hpp:108)
        if not streaming:
            return exp
        bad = np.array([c not in b"ACGTUacgtu" for c in rec])
        csum = np.concatenate([[0], np.cumsum(bad)])
        return exp[(csum[k:] - csum[:-k]) == 0]  # streaming: one code per fully valid window

    for streaming in (True, False):
        s = h.session(depth=2)
        if streaming:
            s.reserve(100_000, 64)  # (optional set-up; the random pass lets submit grow the buffers)
        with pytest.raises(L.LphqError):
            s.collect()  # nothing in flight
        s.submit(batches[0], streaming)
        s.submit(batches[1], streaming)
        with pytest.raises(L.LphqError):
            s.submit(batches[2], streaming)  # both slots in flight
        done = 0
        for nxt in range(2, len(batches) + 2):
            got = s.collect()
            assert len(got) == len(batches[done])
            for rec, g in zip(batches[done], got):
                assert np.array_equal(g, want(rec, streaming)), (name, streaming, done, len(rec))
            done += 1
            if nxt < len(batches):
                s.submit(batches[nxt], streaming)
        assert done == len(batches)
        s.close()
    with h.session(depth=1) as s:  # (a context manager too)
        s.submit(batches[0], True)
        assert len(s.collect()) == len(batches[0])
    assert h.replica_info() == [(0, 0)]
    # lifetime: the index closed BEFORE its session -- lphq_close only marks it, the session keeps answering from the replica it holds by value and
    # the last lphq_session_close releases the device memory; a second attach while a session is open does not move anything under it
    s = h.session(depth=2)
    h.attach(0)
    s.submit(batches[0], True)
    h.close()
    with pytest.raises(L.LphqError):
        h(batches[0][0], streaming=True)  # the handle is dead for every other call (the Python mirror holds NULL now)
    s.submit(batches[4], False)
    got0, got4 = s.collect(), s.collect()
    assert all(np.array_equal(g, want(r, True)) for g, r in zip(got0, batches[0])) and all(np.array_equal(g, want(r, False)) for g, r in zip(got4, batches[4]))
    s.close()
