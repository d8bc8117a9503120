#!/usr/bin/env python3
"""One GPU build of a synthetic set (child of tools/search_sweep.py; also what a kernel trace of the construction is taken on):  <repo root> k m kmer_bits n_records out.lph"""
import hashlib
import sys
import time

sys.path.insert(0, sys.argv[1])
import lphash_amd as L
from lphash_amd.synth import synthetic_records

k, m, bits, n_rec = map(int, sys.argv[2:6])
_, recs = synthetic_records(seed=1, n_records=n_rec, length=100_000 + k - 1)
t = time.time()
L.build(recs, k, m, sys.argv[6], kmer_bits=bits, threads=0, device=0)
print("build_s", round(time.time() - t, 2))
h = hashlib.sha256()
with open(sys.argv[6], "rb") as f:
    for b in iter(lambda: f.read(1 << 24), b""):
        h.update(b)
print("sha256", h.hexdigest())
