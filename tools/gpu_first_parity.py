"""First GPU parity run (scratch tool): product HIP path vs the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import lphash_amd as L, orc
from util import random_records, pack_kmers

dev = torch.device("cuda:0")
ok_all = True
for (k, m, kb, kind, n, lo, hi) in [(31, 20, 64, 0, 30, 20, 8000), (31, 15, 128, 0, 20, 31, 5000), (63, 20, 128, 0, 10, 100, 5000),
                                    (31, 20, 64, 1, 20, 50, 5000), (15, 9, 64, 0, 50, 10, 300), (31, 31, 64, 0, 5, 31, 1000),
                                    (21, 8, 64, 0, 30, 21, 3000), (21, 8, 64, 1, 30, 21, 3000), (47, 17, 128, 1, 10, 40, 9000),
                                    (63, 17, 128, 0, 10, 100, 5000)]:
    recs = random_records(k * 1000 + m, n, lo, hi)
    path = f"/tmp/t_{k}_{m}_{kb}_{kind}.lph"
    L.build(recs, k, m, path, kind=kind, kmer_bits=kb, threads=8)
    o = orc.Index(path, kind, kb)
    cls = L.Mphf if kind == 0 else L.MphfAlt
    h = cls.load(path, kmer_bits=kb, device=0)
    bad = 0
    for r in recs + random_records(99, 5, 10, 3000):  # members + non-members
        exp = o(r, False)
        got = h(r, streaming=False)
        if len(exp) != len(got) or not np.array_equal(exp, got):
            bad += 1
    # packed batch
    big = max(recs, key=len)
    pos = np.arange(0, len(big) - k + 1)
    pk = pack_kmers(big, pos, k, kb)
    d_k = torch.from_numpy(pk.view(np.int64)).to(dev)
    d_o = torch.empty(len(pos), dtype=torch.int64, device=dev)
    h.query_packed(d_k.data_ptr(), len(pos), d_o.data_ptr())
    torch.cuda.synchronize()
    got = d_o.cpu().numpy().view(np.uint64)
    exp = o.query_packed(pk)
    okp = np.array_equal(got, exp)
    # pack kernel
    d_s = torch.from_numpy(np.frombuffer(big, dtype=np.uint8).copy()).to(dev)
    d_p = torch.from_numpy(pos.astype(np.int64)).to(dev)
    d_k2 = torch.empty_like(d_k)
    h.pack_kmers(d_s.data_ptr(), d_p.data_ptr(), len(pos), d_k2.data_ptr())
    torch.cuda.synchronize()
    okk = torch.equal(d_k, d_k2)
    print(f"k={k} m={m} bits={kb} kind={kind}: contig mismatches={bad} packed={okp} packkernel={okk} nk={h.get_kmer_count()} fb={h.info.fallback_kmers}")
    ok_all &= (bad == 0) and okp and okk
print("ALL OK" if ok_all else "FAILURES")
sys.exit(0 if ok_all else 1)
