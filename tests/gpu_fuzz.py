"""One-off robustness run on the GPU (not part of the test suite: minutes of oracle time): random (k, m, width, kind) configurations, random record
batches with ragged lengths, empty records, non-ACGT bases and every pointer alignment, through lphq_query_records (streaming and random), lphq_query_packed,
lphq_query_contig and lphq_session_* -- every code compared with the CPU oracle.   gpurun --timeout 1500 -- python tests/gpu_fuzz.py [seconds [seed]]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lphash_amd as L  # noqa: E402
import orc  # noqa: E402
from runs_util import ARENAS, COUNTS_DTYPE, RUN_DTYPE, check_runs_against_codes, expand_runs_numpy  # noqa: E402
from util import dedup_records, genomic_like_records, pack_kmers, random_records  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261002)  # (a second run takes another seed: other configurations)
t_end = time.time() + budget
n_cfg = n_codes = n_genomic = n_fallback = n_runs = n_explicit = 0
tmp = tempfile.mkdtemp()
while time.time() < t_end:
    bits = int(rng.choice([64, 128]))
    k = int(rng.integers(2, bits // 2))
    m = int(rng.integers(max(1, k - 61), min(k, 32) + 1))
    kind = int(rng.integers(0, 3))  # 2 = bare PTHash over k-mers
    nrec = int(rng.integers(3, 40))
    genomic = k >= 14 and rng.random() < 0.4  # round 5: not only i.i.d. uniform bases -- GC skew, tandem repeats, diverged copies (shared minimizers: fallback path)
    if genomic:
        recs = genomic_like_records(int(rng.integers(1 << 30)), nrec, k, int(rng.choice([300, 3000, 12000])))
        n_genomic += 1
        if not recs:
            continue
    else:
        recs = random_records(int(rng.integers(1 << 30)), nrec, k, int(rng.choice([k + 5, 300, 3000, 12000])))
    if k < 14:
        recs = dedup_records(recs, k)[:6]
        if not recs:
            continue
    path = os.path.join(tmp, "f.lph")
    try:
        if kind == 2:
            L.build_pthash(recs, k, path, kmer_bits=bits, threads=8)
            o, h = orc.PtHash(path, k, bits), L.PtHash.load(path, k, kmer_bits=bits, device=0)
        else:
            L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8)
            o = orc.Index(path, kind, bits)
            h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
            n_fallback += int(o.fallback_keys)
    except L.LphqError as e:  # duplicate k-mers in random short records / unplaceable tiny sets: not what is under test
        if "duplicate" in str(e) or "no pilot" in str(e):
            continue
        raise
    # batch: members, non-members, short, empty, records with invalid bases
    batch = list(recs) + random_records(int(rng.integers(1 << 30)), 5, 0, 2000) + [b"", recs[0][: k - 1]]
    if genomic:  # queries of that shape too: repeats far longer than a window, runs of N of k bases and more
        unit = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(rng.integers(1, 6)))])
        batch += [unit * (3000 // len(unit)), recs[0][: len(recs[0]) // 2] + b"N" * (k + int(rng.integers(0, 9))) + recs[0][len(recs[0]) // 2:], b"A" * 500 + recs[-1]]
    for i in range(0, len(batch), 3):
        q = bytearray(batch[i])
        for p in rng.integers(0, max(len(q), 1), 3):
            if len(q):
                q[p] = ord("N")
        batch.append(bytes(q))
    order = rng.permutation(len(batch))
    batch = [batch[i] for i in order]
    lens = np.array([len(r) for r in batch], dtype=np.int64)
    seq_off = np.concatenate([[0], np.cumsum(lens)])
    out_off = np.concatenate([[0], np.cumsum(np.maximum(lens - k + 1, 0))])
    flat = np.frombuffer(b"".join(batch), dtype=np.uint8)
    mis = int(rng.integers(0, 16))
    buf = torch.zeros(len(flat) + 64, dtype=torch.uint8, device=dev)
    buf[mis:mis + len(flat)] = torch.from_numpy(flat.copy()).to(dev)
    d_so, d_oo = torch.from_numpy(seq_off).to(dev), torch.from_numpy(out_off).to(dev)
    for streaming in (True, False):
        d_out = torch.full((int(out_off[-1]) + 1,), -7, dtype=torch.int64, device=dev)
        h.query_records(buf.data_ptr() + mis, len(flat), d_so.data_ptr(), len(batch), d_out.data_ptr(), d_oo.data_ptr(), streaming)
        torch.cuda.synchronize()
        res = d_out.cpu().numpy().view(np.uint64)
        assert res[-1] == np.uint64(2**64 - 7)
        for i, r in enumerate(batch):
            got = res[out_off[i]:out_off[i + 1]]
            bad = np.array([c not in b"ACGTUacgtu" for c in r])
            cs = np.concatenate([[0], np.cumsum(bad)])
            valid = (cs[k:] - cs[:-k]) == 0 if len(r) >= k else np.zeros(0, dtype=bool)
            if kind == 2:
                exp_valid = o(r)
                assert np.array_equal(got[valid], exp_valid) and (got[~valid] == np.uint64(2**64 - 1)).all(), (k, m, bits, kind, streaming, i)
            elif streaming:
                exp = o(r, False)  # random-mode codes of the windows; streaming emits them for fully valid windows only
                assert np.array_equal(got[valid], exp[valid]) and (got[~valid] == np.uint64(2**64 - 1)).all(), (k, m, bits, kind, streaming, i)
                assert np.array_equal(h(r, streaming=True), exp[valid]), (k, m, bits, kind, "contig", i)  # one-record call: codes of the fully valid windows
            else:
                assert np.array_equal(got, o(r, False)), (k, m, bits, kind, streaming, i)
            n_codes += len(got)
    if kind != 2:
        # round 6: the streaming pass with its result as runs, expanded on the device and by the plain numpy expander, and fused with the payload gather -- against
        # the streaming codes just checked (`res` of the random pass was last: take the streaming pass again)
        d_c = torch.empty(int(out_off[-1]) + 1, dtype=torch.int64, device=dev)
        h.query_records(buf.data_ptr() + mis, len(flat), d_so.data_ptr(), len(batch), d_c.data_ptr(), d_oo.data_ptr(), True)
        n_out = int(out_off[-1])
        tight = rng.random() < 0.3  # arenas sized as a session sizes them (may overflow on low-complexity input: must be reported) or for the worst case
        per_arena = (int(n_out * 2.8 / (k - m + 2) / ARENAS) + 64) if tight else min(n_out, 1024 * (n_out // (ARENAS * 900) + 1)) + 8
        d_runs = torch.zeros((ARENAS * per_arena + 1) * 2, dtype=torch.int64, device=dev)
        d_xc = torch.full((ARENAS * per_arena + 1,), -9, dtype=torch.int64, device=dev)
        d_cnt = torch.empty(ARENAS * 16, dtype=torch.int64, device=dev)
        h.query_records_runs(buf.data_ptr() + mis, len(flat), d_so.data_ptr(), len(batch), d_oo.data_ptr(), d_runs.data_ptr(), ARENAS * per_arena, d_xc.data_ptr(),
                             ARENAS * per_arena, d_cnt.data_ptr())
        d_e = torch.full((n_out + 1,), -5, dtype=torch.int64, device=dev)
        d_st = torch.ones(1, dtype=torch.int64, device=dev)
        h.expand_runs(d_runs.data_ptr(), ARENAS * per_arena, d_xc.data_ptr(), ARENAS * per_arena, d_cnt.data_ptr(), d_e.data_ptr(), n_out, d_st.data_ptr())
        torch.cuda.synchronize()
        counts = d_cnt.cpu().numpy().view(COUNTS_DTYPE)
        over = int(counts["runs"].max()) > per_arena or int(counts["codes"].max()) > per_arena
        assert int(d_xc[-1]) == -9 and int(d_e[-1]) == -5 and (int(d_st.item()) == 1) == over, (k, m, bits, kind, "runs", over, int(d_st.item()))
        if not over:
            assert torch.equal(d_e[:-1], d_c[:-1]), (k, m, bits, kind, "expand_runs")
            valid_all = np.concatenate([((np.concatenate([[0], np.cumsum(np.array([c not in b"ACGTUacgtu" for c in r]))])[k:] - np.concatenate([[0], np.cumsum(np.array([c not in b"ACGTUacgtu" for c in r]))])[:-k]) == 0)
                                        for r in batch if len(r) >= k] or [np.zeros(0, dtype=bool)])
            runs_np = d_runs[:-2].cpu().numpy().view(RUN_DTYPE).reshape(ARENAS, per_arena)
            xc_np = d_xc[:-1].cpu().numpy().view(np.uint64).reshape(ARENAS, per_arena)
            kinds = check_runs_against_codes(runs_np, xc_np, counts, d_c[:-1].cpu().numpy().view(np.uint64), L.INVALID, valid_all)
            n_runs += sum(kinds)
            n_explicit += int(counts["codes"].sum())
            if n_out < 200_000:
                assert np.array_equal(expand_runs_numpy(runs_np, xc_np, counts, n_out, L.INVALID), d_c[:-1].cpu().numpy().view(np.uint64))
        nk = h.get_kmer_count()
        eb = int(rng.choice([1, 2, 4, 8]))
        dt = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[eb]
        table = torch.randint(0, 100, (max(nk, 1),), device=dev, dtype=torch.int64).to(dt)
        pay = torch.full((n_out + 8,), 111, dtype=dt, device=dev)
        h.query_records_gather(buf.data_ptr() + mis, len(flat), d_so.data_ptr(), len(batch), d_oo.data_ptr(), table.data_ptr(), eb, pay.data_ptr(), missing=120)
        torch.cuda.synchronize()
        cc = d_c[:-1]
        want = torch.where((cc >= 0) & (cc < nk), table[cc.clamp(0, max(nk - 1, 0))], torch.full_like(table[:1], 120).expand(n_out))
        assert torch.equal(pay[:n_out], want) and bool((pay[n_out:] == 111).all()), (k, m, bits, kind, "gather", eb)
        n_codes += 2 * n_out
        # ... and through a session as runs: the dense array == the same codes
        with h.session(depth=2) as ses:
            ses.submit_runs(batch)
            r_d, x_d, n_win, offs = ses.collect_runs()
        one = np.zeros(ARENAS, dtype=COUNTS_DTYPE)
        one[0]["runs"], one[0]["codes"] = len(r_d), len(x_d)
        pr = np.zeros((ARENAS, len(r_d) + 1), dtype=RUN_DTYPE)
        pc = np.zeros((ARENAS, len(x_d) + 1), dtype=np.uint64)
        pr[0, :len(r_d)], pc[0, :len(x_d)] = r_d, x_d
        got_d = expand_runs_numpy(pr, pc, one, n_out, L.INVALID) if n_out < 400_000 else None
        cnp = cc.cpu().numpy().view(np.uint64)
        if got_d is not None:
            vm = got_d != np.uint64(L.INVALID)
            assert np.array_equal(got_d[vm], cnp[vm]) and int(offs[-1]) == n_out, (k, m, bits, kind, "session runs")
    if kind != 2:  # the same batch through a session (ABI 3), two halves in flight: per record what the one-record call returns
        half = len(batch) // 2
        for streaming in (True, False):
            with h.session(depth=2) as ses:
                ses.submit(batch[:half], streaming)
                ses.submit(batch[half:], streaming)
                got = ses.collect() + ses.collect()
            assert len(got) == len(batch)
            for i, r in enumerate(batch):
                exp = o(r, False) if len(r) >= k else np.zeros(0, dtype=np.uint64)
                if streaming and len(r) >= k:
                    bad = np.array([c not in b"ACGTUacgtu" for c in r])
                    cs = np.concatenate([[0], np.cumsum(bad)])
                    exp = exp[(cs[k:] - cs[:-k]) == 0]
                assert np.array_equal(got[i], exp), (k, m, bits, kind, "session", streaming, i)
                n_codes += len(exp)
    seq = b"".join(recs)[:5000]
    if len(seq) >= k:
        pos = np.arange(0, len(seq) - k + 1, dtype=np.int64)
        pk = pack_kmers(seq, pos, k, bits)
        d_k = torch.from_numpy(pk.view(np.int64)).to(dev)
        d_o = torch.empty(len(pos), dtype=torch.int64, device=dev)
        h.query_packed(d_k.data_ptr(), len(pos), d_o.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_o.cpu().numpy().view(np.uint64), o.query_packed(pk)), (k, m, bits, kind, "packed")
    h.close()
    n_cfg += 1
print(f"gpu fuzz ok: {n_cfg} random configurations ({n_genomic} drawn with genomic-like records; {n_fallback} k-mers in fallback functions in total), "
      f"{n_codes} codes compared with the oracle in {budget:.0f} s; run-encoded results: {n_runs} runs, {n_explicit} explicit codes checked")
