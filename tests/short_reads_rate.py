"""Streaming / random-over-records kernels on many short records (150-bp reads) -- scratch measurement."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import lphash_amd as L, orc
from lphash_amd.synth import synthetic_records
dev = torch.device("cuda:0")
k, m = 31, 20
flat, recs = synthetic_records(1, 100, 100_030)
L.build(recs, k, m, "/tmp/sr.lph", kmer_bits=64, threads=32, device=0)
h = L.Mphf.load("/tmp/sr.lph", kmer_bits=64, device=0)
rng = np.random.default_rng(0)
for rl, nreads in ((150, 2_000_000), (100, 2_000_000), (1000, 300_000), (36, 2_000_000)):
    starts = rng.integers(0, len(flat) - rl, nreads)
    idx = (starts[:, None] + np.arange(rl)[None, :]).reshape(-1)
    reads = torch.from_numpy(flat[idx].copy()).to(dev)
    seq_off = torch.arange(0, nreads + 1, dtype=torch.int64, device=dev) * rl
    nw = max(rl - k + 1, 0)
    out_off = torch.arange(0, nreads + 1, dtype=torch.int64, device=dev) * nw
    out = torch.empty(nreads * nw + 1, dtype=torch.int64, device=dev)
    for streaming in (True, False):
        h.query_records(reads.data_ptr(), nreads * rl, seq_off.data_ptr(), nreads, out.data_ptr(), out_off.data_ptr(), streaming)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            h.query_records(reads.data_ptr(), nreads * rl, seq_off.data_ptr(), nreads, out.data_ptr(), out_off.data_ptr(), streaming)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        print(f"reads of {rl} bp x {nreads}: streaming={streaming}: {nreads * nw / dt / 1e9:.1f} G k-mers/s ({nreads * rl / dt / 1e9:.1f} G bases/s)")
    # spot check vs oracle
    o = orc.Index("/tmp/sr.lph", 0, 64)
    got = out[: 50 * nw].cpu().numpy().view(np.uint64)
    exp = np.concatenate([o(flat[s:s + rl].tobytes(), True) for s in starts[:50]]) if nw else np.zeros(0, np.uint64)
    assert np.array_equal(got, exp)
print("spot checks vs oracle ok")
