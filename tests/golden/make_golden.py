#!/usr/bin/env python3
"""Generate the committed golden fixtures with the CPU oracle (oracle/lph_oracle.c).

The reference holds no golden vectors for the query path and cannot be built in this image (SURVEY.md
section 8c), so these fixtures pin the ORACLE (against regressions) and give the GPU tests fixed inputs;
they do not pin the oracle to the reference ("parity unpinned").  tests/golden/pymodel.py -- a second, independently structured
restatement in pure Python, written from the reference sources -- regenerates every .lph here byte for byte and every query code
(tests/test_oracle_cpu.py::test_python_model_*): transcription errors are covered; the PTHash recall itself is not.

Each case: <name>.lph (index, reference serialization), <name>.npz with
  records   : the indexed records (uint8, concatenated) + rec_off
  queries   : query records (members + non-members) + q_off
  streaming : oracle codes of MPHF::operator()(q, streaming=true), concatenated + s_off
  random    : oracle codes of streaming=false, concatenated + r_off
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402
from util import random_records  # noqa: E402

CASES = [
    # name, k, m, kmer_bits, kind, n_records, lmin, lmax
    ("p_k31_m20_b64", 31, 20, 64, orc.PARTITIONED, 6, 200, 900),
    ("p_k31_m15_b128", 31, 15, 128, orc.PARTITIONED, 6, 200, 900),
    ("p_k63_m20_b128", 63, 20, 128, orc.PARTITIONED, 5, 300, 900),
    ("u_k31_m20_b64", 31, 20, 64, orc.UNPARTITIONED, 6, 200, 900),
    ("p_k21_m8_b64_fallback", 21, 8, 64, orc.PARTITIONED, 8, 100, 700),
    ("u_k21_m8_b64_fallback", 21, 8, 64, orc.UNPARTITIONED, 8, 100, 700),
    ("p_k47_m17_b128", 47, 17, 128, orc.PARTITIONED, 4, 300, 800),
]


def cat(arrs, dtype):
    off = np.zeros(len(arrs) + 1, dtype=np.int64)
    for i, a in enumerate(arrs):
        off[i + 1] = off[i] + len(a)
    flat = np.concatenate([np.asarray(a, dtype=dtype) for a in arrs]) if arrs else np.zeros(0, dtype=dtype)
    return flat, off


def main():
    for ci, (name, k, m, bits, kind, n, lo, hi) in enumerate(CASES):
        recs = random_records(1000 + ci, n, lo, hi)
        # make sure the k-mer set has no duplicates (the format requires it, README.md:112-116)
        seen = set()
        for r in recs:
            for i in range(len(r) - k + 1):
                km = r[i:i + k]
                assert km not in seen, "duplicate k-mer in fixture input; change the seed"
                seen.add(km)
        path = os.path.join(HERE, name + ".lph")
        orc.build(recs, k, m, path, kind=kind, kmer_bits=bits)
        ix = orc.Index(path, kind, bits)
        queries = list(recs) + random_records(2000 + ci, 2, k + 5, 400) + [recs[0][: k - 1], recs[0][:k], b""]
        s = [ix(q, True) for q in queries]
        r = [ix(q, False) for q in queries]
        rec_flat, rec_off = cat([np.frombuffer(x, dtype=np.uint8) for x in recs], np.uint8)
        q_flat, q_off = cat([np.frombuffer(x, dtype=np.uint8) for x in queries], np.uint8)
        s_flat, s_off = cat(s, np.uint64)
        r_flat, r_off = cat(r, np.uint64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), k=k, m=m, kmer_bits=bits, kind=kind, nkmers=ix.nkmers,
                            distinct_minimizers=ix.distinct_minimizers, fallback_keys=ix.fallback_keys,
                            records=rec_flat, rec_off=rec_off, queries=q_flat, q_off=q_off,
                            streaming=s_flat, s_off=s_off, random=r_flat, r_off=r_off)
        print(name, "kmers", ix.nkmers, "minimizers", ix.distinct_minimizers, "fallback", ix.fallback_keys,
              "bytes", os.path.getsize(path))


if __name__ == "__main__":
    main()
