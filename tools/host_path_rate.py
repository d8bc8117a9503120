"""PCIe-inclusive rate of the host-buffer convenience call lphq_query_contig (scratch measurement for DESIGN.md)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lphash_amd as L
from lphash_amd.synth import synthetic_records
flat, recs = synthetic_records(seed=1, n_records=100, length=100_030)
L.build(recs, 31, 20, "/tmp/hp.lph", kmer_bits=64, threads=32, device=0)
h = L.Mphf.load("/tmp/hp.lph", kmer_bits=64, device=0)
big = flat.tobytes()  # one 10 Mbase record
for name, rec in (("100 kbase record", recs[0]), ("10 Mbase record", big)):
    for streaming in (True, False):
        h(rec, streaming=streaming)
        t = time.perf_counter(); reps = 20 if len(rec) < 1e6 else 5
        for _ in range(reps): out = h(rec, streaming=streaming)
        dt = (time.perf_counter() - t) / reps
        print(f"{name} streaming={streaming}: {len(out) / dt / 1e9:.3f} G k-mers/s ({dt * 1e3:.3f} ms per call)")
