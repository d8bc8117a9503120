"""GPU parity of the run-encoded streaming result and of the fused lookup -> payload pass (include/lphq.h, ABI 4): the HIP path through the C ABI
against the oracle's streaming codes (the reference's `hval +- 1` updates, include/partitioned_mphf.hpp:131-145, unpartitioned_mphf.hpp:130-137).
Bit-exact: expand(runs) == the codes lphq_query_records writes == the oracle; gather == table[oracle codes]."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import lphash_amd as L  # noqa: E402
import orc  # noqa: E402
from runs_util import ARENAS, COUNTS_DTYPE, RUN_DTYPE, check_runs_against_codes, expand_runs_numpy  # noqa: E402
from util import dedup_records, random_records  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def torch_dev():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a device; the product has no CPU fallback"
    return torch, torch.device("cuda", 0)


class DeviceBatch:
    """Records laid out back to back on the device, with the two offset arrays lphq_query_records wants."""

    def __init__(self, torch, dev, recs, k, misalign=0):
        lens = np.array([len(r) for r in recs], dtype=np.int64)
        self.seq_off = np.concatenate([[0], np.cumsum(lens)])
        self.out_off = np.concatenate([[0], np.cumsum(np.maximum(lens - k + 1, 0))])
        flat = np.frombuffer(b"".join(recs), dtype=np.uint8)
        self.n_bases, self.n_out, self.n_records = len(flat), int(self.out_off[-1]), len(recs)
        self.buf = torch.zeros(len(flat) + 64, dtype=torch.uint8, device=dev)
        self.buf[misalign:misalign + len(flat)] = torch.from_numpy(flat.copy()).to(dev)
        self.d_seq = self.buf.data_ptr() + misalign
        self.d_so, self.d_oo = torch.from_numpy(self.seq_off).to(dev), torch.from_numpy(self.out_off).to(dev)


def gpu_codes(h, torch, dev, b):
    d_out = torch.full((b.n_out + 1,), -7, dtype=torch.int64, device=dev)
    h.query_records(b.d_seq, b.n_bases, b.d_so.data_ptr(), b.n_records, d_out.data_ptr(), b.d_oo.data_ptr(), True)
    torch.cuda.synchronize()
    assert int(d_out[-1]) == -7
    return d_out[:-1]


def gpu_runs(h, torch, dev, b, per_arena=None):
    """lphq_query_records_runs -> (runs[ARENAS, per_arena], codes[ARENAS, per_arena], counts) as numpy + the device tensors."""
    per_arena = per_arena if per_arena is not None else min(b.n_out, 1024 * (b.n_out // (ARENAS * 900) + 1)) + 8  # worst case: every window a run of its own
    d_runs = torch.zeros((ARENAS * per_arena + 1) * 2, dtype=torch.int64, device=dev)
    d_runs[-2:] = -9
    d_codes = torch.full((ARENAS * per_arena + 1,), -9, dtype=torch.int64, device=dev)
    d_counts = torch.full((ARENAS * 16,), 77, dtype=torch.int64, device=dev)  # (the call zeroes them)
    h.query_records_runs(b.d_seq, b.n_bases, b.d_so.data_ptr(), b.n_records, b.d_oo.data_ptr(), d_runs.data_ptr(), ARENAS * per_arena, d_codes.data_ptr(),
                         ARENAS * per_arena, d_counts.data_ptr())
    torch.cuda.synchronize()
    assert d_runs[-2:].tolist() == [-9, -9] and int(d_codes[-1]) == -9
    runs = d_runs[:-2].cpu().numpy().view(RUN_DTYPE).reshape(ARENAS, per_arena)
    codes = d_codes[:-1].cpu().numpy().view(np.uint64).reshape(ARENAS, per_arena)
    counts = d_counts.cpu().numpy().view(COUNTS_DTYPE)
    return runs, codes, counts, (d_runs, d_codes, d_counts, per_arena)


def gpu_expand(h, torch, dev, dv, n_out):
    d_runs, d_codes, d_counts, per_arena = dv
    d_out = torch.full((n_out + 1,), -5, dtype=torch.int64, device=dev)
    d_st = torch.full((1,), 9, dtype=torch.int64, device=dev)
    h.expand_runs(d_runs.data_ptr(), ARENAS * per_arena, d_codes.data_ptr(), ARENAS * per_arena, d_counts.data_ptr(), d_out.data_ptr(), n_out, d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_out[-1]) == -5
    return d_out[:-1], int(d_st.item())


def check_gather(h, torch, dev, b, d_codes_exp, seed=1):
    """lphq_query_records_gather for the four element sizes == table[codes] with `missing` where the code is no slot, == lphq_gather_payload of the codes."""
    nk = h.get_kmer_count()
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    slot = d_codes_exp.clamp(0, nk - 1)
    inside = (d_codes_exp >= 0) & (d_codes_exp < nk)  # (int64 view: LPHQ_INVALID and other codes >= 2^63 read as negative)
    for eb, dt in ((1, torch.uint8), (2, torch.int16), (4, torch.int32), (8, torch.int64)):
        table = torch.randint(0, 120, (nk,), generator=g, device=dev, dtype=torch.int64).to(dt)
        missing = 125
        out = torch.full((b.n_out + 16,), 99, dtype=dt, device=dev)
        h.query_records_gather(b.d_seq, b.n_bases, b.d_so.data_ptr(), b.n_records, b.d_oo.data_ptr(), table.data_ptr(), eb, out.data_ptr(), missing=missing)
        torch.cuda.synchronize()
        want = torch.where(inside, table[slot], torch.full_like(table[:1], missing).expand(b.n_out))
        assert torch.equal(out[:b.n_out], want) and bool((out[b.n_out:] == 99).all()), eb
        two = torch.empty(b.n_out, dtype=dt, device=dev)
        h.gather_payload(d_codes_exp.data_ptr(), b.n_out, table.data_ptr(), eb, two.data_ptr(), missing=missing)
        torch.cuda.synchronize()
        assert torch.equal(two, want)


CONFIGS = [(31, 20, 64, 0), (63, 20, 128, 0), (31, 20, 64, 1), (21, 8, 64, 0), (47, 17, 128, 1), (31, 31, 64, 0), (15, 9, 64, 0), (9, 7, 64, 0), (15, 8, 64, 0),
           (35, 20, 128, 0), (31, 15, 64, 0), (37, 20, 128, 0), (38, 20, 128, 1), (41, 20, 128, 0), (33, 5, 128, 0), (63, 32, 128, 0), (3, 2, 64, 0)]


@pytest.mark.parametrize("k,m,bits,kind", CONFIGS)
def test_runs_and_gather_match_the_codes_and_the_oracle(tmp_path, k, m, bits, kind):
    """Ragged batches (empty, short and tile-straddling records, > 64 records per tile, non-members, non-ACGT bytes, every pointer alignment class)."""
    torch, dev = torch_dev()
    recs = dedup_records(random_records(k * 19 + m, 30 if k > 12 else 80, k, 3000 if k > 12 else k + 12), k)
    path = str(tmp_path / "t.lph")
    try:
        L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8)
    except L.LphqError as e:
        assert "no pilot below" in str(e)
        pytest.skip("degenerate tiny key set: PTHash cannot place a bucket")
    o = orc.Index(path, kind, bits)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
    noisy = bytearray(recs[3])
    noisy[len(noisy) // 2] = ord("N")
    noisy[2:2 + k + 3] = b"n" * (k + 3)
    batch = [b"", recs[0], recs[1][: k - 1], b"", recs[2][:k], bytes(noisy)] + recs[3:] + random_records(3, 6, 1, 1500) + [b""]
    batch += random_records(11, 150, 0, 5) + [recs[0][:k + 2]] + random_records(12, 60, k, k + 20) + [recs[1]]
    from test_runs_cpu import streaming_expectation

    exp, valid = streaming_expectation(o, batch, k, with_valid=True)
    for mis in (0, 1, 7, 15):
        b = DeviceBatch(torch, dev, batch, k, mis)
        d_c = gpu_codes(h, torch, dev, b)
        assert np.array_equal(d_c.cpu().numpy().view(np.uint64), exp)
        runs, codes, counts, dv = gpu_runs(h, torch, dev, b)
        check_runs_against_codes(runs, codes, counts, exp, L.INVALID, valid)
        assert np.array_equal(expand_runs_numpy(runs, codes, counts, b.n_out, L.INVALID), exp)
        d_e, st = gpu_expand(h, torch, dev, dv, b.n_out)
        assert st == 0 and torch.equal(d_e, d_c)
    check_gather(h, torch, dev, b, d_c)


def golden_cases():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.lph")))


@pytest.mark.parametrize("name", golden_cases())
def test_runs_on_golden_fixtures(name, golden_dir):
    """The seven committed fixtures: expand(runs) == the committed streaming codes, record by record."""
    torch, dev = torch_dev()
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    k = int(z["k"])
    h = (L.Mphf if int(z["kind"]) == 0 else L.MphfAlt).load(os.path.join(golden_dir, name + ".lph"), kmer_bits=int(z["kmer_bits"]), device=0)
    queries = [bytes(z["queries"][z["q_off"][i]:z["q_off"][i + 1]]) for i in range(len(z["q_off"]) - 1)]
    clean = [q for q in queries if all(c in b"ACGTUacgtu" for c in q)]  # (the committed streaming codes are compacted: non-ACGT windows dropped)
    want = np.concatenate([z["streaming"][z["s_off"][i]:z["s_off"][i + 1]] for i, q in enumerate(queries) if all(c in b"ACGTUacgtu" for c in q)] or [np.zeros(0, np.uint64)])
    b = DeviceBatch(torch, dev, clean, k, 5)
    assert b.n_out == len(want)
    runs, codes, counts, dv = gpu_runs(h, torch, dev, b)
    d_e, st = gpu_expand(h, torch, dev, dv, b.n_out)
    assert st == 0 and np.array_equal(d_e.cpu().numpy().view(np.uint64), want)
    check_runs_against_codes(runs, codes, counts, want, L.INVALID)


@pytest.mark.parametrize("name", ["se_k31_m20_b64_p", "se_k63_m20_b128_p", "se_k47_m20_b128_u", "se_k31_m20_b64_u"])
def test_runs_and_gather_on_real_genomes(tmp_path, name):
    """tests/golden/real (the reference's bundled unitigs and query genomes): all unitigs (members: the runs expand to a bijection), the Salmonella and
    E. coli slices with their runs of N and the FASTQ reads, == lphq_query_records and == the oracle; super-k-mer statistics of real data."""
    import json

    from real_fixture import CASES, OUT, load_inputs, query_sets, unitig_records
    from test_runs_cpu import streaming_expectation

    torch, dev = torch_dev()
    kind, k, m, bits, ust = CASES[name]
    z = load_inputs()
    recs = unitig_records(z, ust)
    path = os.path.join(OUT, name + ".lph")
    if not os.path.exists(path):
        path = str(tmp_path / "x.lph")
        L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8, device=0)
    e = json.load(open(os.path.join(OUT, "expected.json")))["cases"][name]
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
    assert h.get_kmer_count() == e["nkmers"]
    # members: every unitig
    b = DeviceBatch(torch, dev, recs, k, 3)
    d_c = gpu_codes(h, torch, dev, b)
    runs, codes, counts, dv = gpu_runs(h, torch, dev, b, per_arena=b.n_out // ARENAS + 4096)
    d_e, st = gpu_expand(h, torch, dev, dv, b.n_out)
    assert st == 0 and torch.equal(d_e, d_c)
    seen = torch.zeros(e["nkmers"], dtype=torch.uint8, device=dev)
    seen[d_e] = 1
    assert int(seen.sum(dtype=torch.int64)) == e["nkmers"] == b.n_out
    n_runs = int(counts["runs"].sum())
    assert 2 * b.n_out / (k - m + 2) * 0.5 < n_runs < b.n_out  # about 2 / (W + 1) runs per window (+ the cuts at tile and wave ends)
    check_gather(h, torch, dev, b, d_c, seed=2)
    # the query sets: non-members, runs of N, reads
    o = orc.Index(path, kind, bits)
    qs = query_sets(z)
    batch = [qs["salmonella"][0][:300_000], qs["ecoli"][0][:200_000], qs["ecoli"][1][:100_000]] + qs["fastq"][:600]
    exp, valid = streaming_expectation(o, batch, k, with_valid=True)
    assert (~valid).any()
    b = DeviceBatch(torch, dev, batch, k, 9)
    d_c = gpu_codes(h, torch, dev, b)
    assert np.array_equal(d_c.cpu().numpy().view(np.uint64), exp)
    runs, codes, counts, dv = gpu_runs(h, torch, dev, b)
    check_runs_against_codes(runs, codes, counts, exp, L.INVALID, valid)
    d_e, st = gpu_expand(h, torch, dev, dv, b.n_out)
    assert st == 0 and torch.equal(d_e, d_c)
    check_gather(h, torch, dev, b, d_c, seed=3)


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (63, 20, 128, 0), (31, 20, 64, 1)])
def test_runs_on_low_complexity_and_repeat_rich_input(tmp_path, k, m, bits, kind):
    """Hash ties (homopolymers, tandem repeats) and 5-70 % of the k-mers behind colliding minimizers: explicit runs carry real work."""
    from test_runs_cpu import streaming_expectation
    from util import genomic_like_records, low_complexity_index_records, low_complexity_queries

    torch, dev = torch_dev()
    for which in ("ties", "repeats"):
        recs = low_complexity_index_records(k, 900 + k + m) if which == "ties" else genomic_like_records(1000 + k + m, 40, k, 6000)
        path = str(tmp_path / f"{which}.lph")
        L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8)
        o = orc.Index(path, kind, bits)
        h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
        batch = low_complexity_queries(recs, k, 5 + k) if which == "ties" else recs + [recs[0][::-1], b"AC" * 400, b""]
        exp, valid = streaming_expectation(o, batch, k, with_valid=True)
        b = DeviceBatch(torch, dev, batch, k, 11)
        d_c = gpu_codes(h, torch, dev, b)
        assert np.array_equal(d_c.cpu().numpy().view(np.uint64), exp)
        runs, codes, counts, dv = gpu_runs(h, torch, dev, b)
        kinds = check_runs_against_codes(runs, codes, counts, exp, L.INVALID, valid)
        if which == "repeats":
            assert kinds[2] > 0 and int(counts["codes"].sum()) > 0.03 * b.n_out
        d_e, st = gpu_expand(h, torch, dev, dv, b.n_out)
        assert st == 0 and torch.equal(d_e, d_c)
        check_gather(h, torch, dev, b, d_c, seed=4)


def test_runs_overflow_is_reported_and_argument_errors(tmp_path):
    torch, dev = torch_dev()
    k, m = 31, 20
    recs = dedup_records(random_records(77, 12, 1500, 2500), k)
    path = str(tmp_path / "t.lph")
    L.build(recs, k, m, path, kmer_bits=64, threads=4)
    h = L.Mphf.load(path, kmer_bits=64, device=0)
    b = DeviceBatch(torch, dev, recs, k)
    runs, codes, counts, dv = gpu_runs(h, torch, dev, b, per_arena=2)
    assert int(counts["runs"].max()) > 2
    _, st = gpu_expand(h, torch, dev, dv, b.n_out)
    assert st == 1
    with pytest.raises(L.LphqError) as ei:  # capacities are whole arenas
        h.query_records_runs(b.d_seq, b.n_bases, b.d_so.data_ptr(), b.n_records, b.d_oo.data_ptr(), dv[0].data_ptr(), 33, dv[1].data_ptr(), 32, dv[2].data_ptr())
    assert ei.value.code == L._lib.ERR_ARG
    pt = str(tmp_path / "pt.ph")
    L.build_pthash(recs, k, pt, kmer_bits=64, threads=4)
    hp = L.PtHash.load(pt, k, kmer_bits=64, device=0)
    with pytest.raises(L.LphqError) as ei:  # a competitor's codes are not affine
        hp.query_records_runs(b.d_seq, b.n_bases, b.d_so.data_ptr(), b.n_records, b.d_oo.data_ptr(), dv[0].data_ptr(), 64, dv[1].data_ptr(), 64, dv[2].data_ptr())
    assert ei.value.code == L._lib.ERR_ARG
    with pytest.raises(L.LphqError):
        hp.query_records_gather(b.d_seq, b.n_bases, b.d_so.data_ptr(), b.n_records, b.d_oo.data_ptr(), dv[1].data_ptr(), 8, dv[1].data_ptr())
    with pytest.raises(L.LphqError):
        h.query_records_gather(b.d_seq, b.n_bases, b.d_so.data_ptr(), b.n_records, b.d_oo.data_ptr(), dv[1].data_ptr(), 3, dv[1].data_ptr())


def _want(o, rec, k, streaming=True):
    from real_fixture import valid_windows

    if len(rec) < k:
        return np.zeros(0, dtype=np.uint64)
    exp = o(rec, False)
    return exp[valid_windows(rec, k)] if streaming else exp


def _dense_expand(runs, xc, n_out):
    """The dense array lphq_session_collect_runs hands out, expanded by the numpy expander (one arena holds everything)."""
    one = np.zeros(ARENAS, dtype=COUNTS_DTYPE)
    one[0]["runs"], one[0]["codes"] = len(runs), len(xc)
    pr = np.zeros((ARENAS, len(runs) + 1), dtype=RUN_DTYPE)
    pc = np.zeros((ARENAS, len(xc) + 1), dtype=np.uint64)
    pr[0, :len(runs)], pc[0, :len(xc)] = runs, xc
    return expand_runs_numpy(pr, pc, one, n_out, L.INVALID)


@pytest.mark.parametrize("name", ["p_k31_m20_b64", "p_k63_m20_b128", "u_k31_m20_b64", "p_k21_m8_b64_fallback"])
def test_session_runs_transport_and_lanes(name):
    """lphq_session_submit_runs / collect_runs: the runs a batch crosses PCIe as == the oracle's codes (windows holding a non-ACGT byte in no run), n_windows ==
    the reference's vector sizes added up; lphq_session_collect on such a batch expands on the host == the oracle; a batch of low-complexity sequence (one
    super-k-mer per window: the arenas sized for random sequence overflow) is answered again with worst-case arenas; sessions over several lanes
    (lphq_session_open_devices, here three lanes on the one device) keep submission order."""
    from real_fixture import valid_windows

    torch_dev()
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
        [b"A" * 60_000, b"AC" * 20_000, long_rec[:2000]],  # one run per window over 100 K windows: beyond the arenas sized for random sequence
        random_records(9, 300, 0, 300),
    ]
    for lanes in (None, [0, 0, 0]):
        s = h.session(depth=2, devices=lanes)
        nl = 1 if lanes is None else len(lanes)
        with pytest.raises(L.LphqError):
            s.collect_runs()
        for b in batches[:2 * nl]:
            s.submit_runs(b)
        if 2 * nl < len(batches):
            with pytest.raises(L.LphqError):
                s.submit_runs(batches[2 * nl])  # every slot of every lane in flight
        nxt = 2 * nl
        for done, b in enumerate(batches):
            if done % 2 == 0:
                runs, xc, n_windows, offs = s.collect_runs()
                n_out = int(offs[-1])
                assert len(offs) == len(b) + 1 and n_out == sum(max(len(r) - k + 1, 0) for r in b)
                got = _dense_expand(runs, xc, n_out)
                for i, rec in enumerate(b):
                    if len(rec) < k:
                        continue
                    v = valid_windows(rec, k)
                    g = got[int(offs[i]):int(offs[i + 1])]
                    assert np.array_equal(g[v], o(rec, False)[v]), (name, lanes, done, i)
                assert n_windows == sum(int(valid_windows(r, k).sum()) for r in b if len(r) >= k)
            else:  # the same batch kind through the expanding collect
                got = s.collect()
                assert len(got) == len(b)
                for rec, g in zip(b, got):
                    assert np.array_equal(g, _want(o, rec, k)), (name, lanes, done, len(rec))
            if nxt < len(batches):
                s.submit_runs(batches[nxt])
                nxt += 1
        s.close()
    # mixing: a random-mode batch is no runs batch
    with h.session(depth=2) as s:
        s.submit(batches[0], streaming=False)
        with pytest.raises(L.LphqError):
            s.collect_runs()
        got = s.collect()
        assert all(np.array_equal(g, _want(o, r, k, False)) for g, r in zip(got, batches[0]))
    h.close()


def test_session_lanes_over_peer_copied_replicas(tmp_path):
    """lphq_session_open_devices over DISTINCT replicas on a one-GPU box (LPHQ_TEST_SELF_PEER: the device attached twice, the second replica a peer copy):
    lanes 0 / 1 / 2 answer from replica 0 / 1 / 1, batches come back in submission order, codes == oracle; with the 8-byte-code transport too."""
    import subprocess
    import sys

    torch_dev()
    script = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import lphash_amd as L, orc
from util import random_records, dedup_records
k, m = 31, 20
recs = dedup_records(random_records(4712, 40, 500, 20000), k)
path = os.path.join(sys.argv[2], "t.lph")
L.build(recs, k, m, path, kmer_bits=64, threads=4)
h = L.Mphf.load(path, kmer_bits=64)
h.attach_devices([0, 0])
assert [d for d, _ in h.replica_info()] == [0, 0]
o = orc.Index(path, 0, 64)
batches = [recs[i::7] for i in range(7)]
s = h.session(depth=2, devices=[0, 0, 0])
order = []
inflight = 0
for i, b in enumerate(batches):
    if inflight == 6:
        got = s.collect(); inflight -= 1
        j = len(order); order.append(j)
        assert all(np.array_equal(g, o(r, True)) for g, r in zip(got, batches[j])), j
    s.submit(b, streaming=True); inflight += 1
while inflight:
    got = s.collect(); inflight -= 1
    j = len(order); order.append(j)
    assert len(got) == len(batches[j]) and all(np.array_equal(g, o(r, True)) for g, r in zip(got, batches[j])), j
s.close()
try:
    h.session(depth=2, devices=[0, 1])
    raise SystemExit("a lane on a device without a replica was accepted")
except L.LphqError as e:
    assert e.code == L._lib.ERR_STATE
print("lanes ok", order)
"""
    for transport in ("runs", "codes"):
        env = dict(os.environ, LPHQ_TEST_SELF_PEER="1", LPHQ_SESSION_TRANSPORT=transport)
        r = subprocess.run([sys.executable, "-c", script, ROOT, str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "lanes ok [0, 1, 2, 3, 4, 5, 6]" in r.stdout, (transport, r.stdout[-500:], r.stderr[-3000:])


@pytest.mark.parametrize("name", ["se_k47_m20_b128_u", "se_k31_m20_b64_p"])
def test_fallback_rich_batch_under_load(tmp_path, name):
    """The short unitigs of the real fixture (repeat regions: most of the function's colliding minimizers) a hundred times over -- thousands of workgroups, every tile
    multi-record, ~1 window in 10 through the fallback function (src/partitioned_mphf.cpp:312-316) -- through the three sinks: every copy == the oracle, three
    launches each.  (A patched payload sink was right on 47 tiles and wrong on 1 fallback window in 100 on this batch: profiles/r06_transposed_gather_attempt.txt.)"""
    from real_fixture import CASES, OUT, load_inputs, unitig_records

    torch, dev = torch_dev()
    kind, k, m, bits, ust = CASES[name]
    recs = unitig_records(load_inputs(), ust)
    path = os.path.join(OUT, name + ".lph")
    if not os.path.exists(path):
        path = str(tmp_path / "x.lph")
        L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8, device=0)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
    o = orc.Index(path, kind, bits)
    short = [r for r in recs if len(r) < 900]
    one = np.concatenate([o(q, False) for q in short])
    reps = 100
    want = torch.from_numpy(np.tile(one, reps).view(np.int64)).to(dev)
    nk = h.get_kmer_count()
    n_fallback = int((one >= np.uint64(nk - o.fallback_keys)).sum()) if hasattr(o, "fallback_keys") else None
    assert n_fallback is None or n_fallback * 20 > len(one)  # a batch that leans on the fallback function
    b = DeviceBatch(torch, dev, short * reps, k, 5)
    assert b.n_out == len(one) * reps
    table = torch.arange(nk, dtype=torch.int64, device=dev)  # the element IS the slot
    for _ in range(3):
        assert torch.equal(gpu_codes(h, torch, dev, b), want)
        runs, codes, counts, dv = gpu_runs(h, torch, dev, b, per_arena=b.n_out // ARENAS + 4096)
        d_e, st = gpu_expand(h, torch, dev, dv, b.n_out)
        assert st == 0 and torch.equal(d_e, want)
        for eb, dt in ((8, torch.int64), (1, torch.uint8)):
            t = table if eb == 8 else (table % 251).to(dt)
            out = torch.full((b.n_out + 16,), 99, dtype=dt, device=dev)
            h.query_records_gather(b.d_seq, b.n_bases, b.d_so.data_ptr(), b.n_records, b.d_oo.data_ptr(), t.data_ptr(), eb, out.data_ptr(), missing=7)
            torch.cuda.synchronize()
            assert torch.equal(out[:b.n_out], t[want]) and bool((out[b.n_out:] == 99).all()), eb


@pytest.mark.parametrize("k,m,bits,kind", [(15, 9, 64, 0), (63, 20, 128, 0), (21, 8, 64, 1), (13, 10, 64, 1), (31, 20, 64, 1), (38, 20, 128, 1), (47, 17, 128, 1), (31, 15, 64, 0),
                                           (37, 20, 128, 0)])
def test_fallback_rich_synthetic_sets_under_load(tmp_path, k, m, bits, kind):
    """The same under-load check for other instantiations of the streaming kernel (narrow windows: W = 7 and 4; k = 63; both kinds and key widths at W = 12 ... 31: other members of the (W - 8) % 8 family): genomic-like
    sets (diverged copies: shared minimizers, colliding ones) of short records, enough copies for ~4 M windows, three launches per sink, every copy == the oracle."""
    torch, dev = torch_dev()
    from util import genomic_like_records

    recs = genomic_like_records(k * 131 + m, 260, k, 700)
    path = str(tmp_path / "g.lph")
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8, device=0)
    h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
    o = orc.Index(path, kind, bits)
    assert o.fallback_keys * 50 > o.nkmers, (o.fallback_keys, o.nkmers)  # the fallback function holds a real share of the set
    one = np.concatenate([o(q, False) for q in recs])
    reps = max(2, 4_000_000 // len(one))
    want = torch.from_numpy(np.tile(one, reps).view(np.int64)).to(dev)
    b = DeviceBatch(torch, dev, recs * reps, k, 1)
    nk = h.get_kmer_count()
    table = torch.arange(nk, dtype=torch.int64, device=dev)
    for _ in range(3):
        assert torch.equal(gpu_codes(h, torch, dev, b), want)
        runs, codes, counts, dv = gpu_runs(h, torch, dev, b, per_arena=b.n_out // ARENAS + 4096)
        d_e, st = gpu_expand(h, torch, dev, dv, b.n_out)
        assert st == 0 and torch.equal(d_e, want)
        out = torch.full((b.n_out + 16,), 99, dtype=torch.int64, device=dev)
        h.query_records_gather(b.d_seq, b.n_bases, b.d_so.data_ptr(), b.n_records, b.d_oo.data_ptr(), table.data_ptr(), 8, out.data_ptr(), missing=7)
        torch.cuda.synchronize()
        assert torch.equal(out[:b.n_out], want) and bool((out[b.n_out:] == 99).all())


def test_forged_runs_are_skipped_and_reported():
    """lphq_expand_runs takes its runs from the caller: entries whose fields would wrap a sum (explicit-code index 2^64 - 1), point beyond an arena or beyond the output, or
    carry no kind are skipped and reported -- nothing is read or written outside the buffers (the device keeps answering afterwards)."""
    from test_runs_cpu import forged_runs

    torch, dev = torch_dev()
    h = L.Mphf.load(os.path.join(ROOT, "tests", "golden", "p_k31_m20_b64.lph"), kmer_bits=64, device=0)
    runs, codes, counts, n_out, want = forged_runs()
    d_runs = torch.from_numpy(np.ascontiguousarray(runs).view(np.int64).reshape(-1).copy()).to(dev)
    d_codes = torch.from_numpy(codes.view(np.int64).reshape(-1).copy()).to(dev)
    d_counts = torch.from_numpy(np.ascontiguousarray(counts).view(np.int64).reshape(-1).copy()).to(dev)
    d_out = torch.full((n_out + 1,), -5, dtype=torch.int64, device=dev)
    d_st = torch.zeros(1, dtype=torch.int64, device=dev)
    h.expand_runs(d_runs.data_ptr(), ARENAS * 8, d_codes.data_ptr(), ARENAS * 8, d_counts.data_ptr(), d_out.data_ptr(), n_out, d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.item()) == 1 and int(d_out[-1]) == -5
    assert np.array_equal(d_out[:-1].cpu().numpy().view(np.uint64), want)
