#!/usr/bin/env python3
"""SHA-256 of the index files the ORACLE builder (oracle/lph_oracle.c: the restatement of src/partitioned_mphf.cpp:33-268 and
src/unpartitioned_mphf.cpp:23-170) writes for bench.py's synthetic sets at the FULL size of BASELINE.json's configurations.

  C2 / C3   partitioned,   k=31 m=20, 1,000 records x 100,030 bases  (100 M k-mers), kmer_t 64 bit
  C5        unpartitioned, the same set
  C4        partitioned,   k=63 m=20, 20,000 records x 100,062 bases (2 B k-mers),   kmer_t 128 bit

Test infrastructure: the table it writes (tests/golden/full_size_digests.json) is what tests/test_gpu_parity.py::test_large_index_properties
and bench.py compare the files the PRODUCT builders write against (CPU search at 100 M, GPU pilot search at 2 B).  About 40 minutes on 8 cores
and ~25 GB of memory for C4 (the oracle builder is single-threaded); run in the build container, never on the GPU box.

    python tools/full_size_digests.py [--only C2,C5,C4] [--out tests/golden/full_size_digests.json]
"""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {  # name -> (kind, k, m, kmer_bits, n_records)   records of 100,000 k-mers each, SplitMix64 seed 1 (lphash_amd/synth.py)
    "C2": (0, 31, 20, 64, 1000),
    "C5": (1, 31, 20, 64, 1000),
    "C4": (0, 63, 20, 128, 20_000),
}


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for b in iter(lambda: f.read(1 << 24), b""):
            h.update(b)
    return h.hexdigest()


def key_of(kind, k, m, bits, n_records):
    return f"{'p' if kind == 0 else 'u'}_k{k}_m{m}_b{bits}_seed1_{n_records}x{100_000 + k - 1}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="C2,C5,C4")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "full_size_digests.json"))
    args = ap.parse_args()
    import orc
    from lphash_amd.synth import synthetic_records

    table = {}
    if os.path.exists(args.out):
        table = json.load(open(args.out))
    for name in args.only.split(","):
        kind, k, m, bits, n_rec = CASES[name]
        t0 = time.time()
        _flat, recs = synthetic_records(seed=1, n_records=n_rec, length=100_000 + k - 1)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.lph")
            orc.build(recs, k, m, path, kind=kind, kmer_bits=bits)
            entry = {"config": name, "bytes": os.path.getsize(path), "sha256": sha256_file(path), "builder": "oracle/lph_oracle.c orc_build",
                     "mm_seed": 42, "c": 3.0, "seconds": round(time.time() - t0, 1)}
        table[key_of(kind, k, m, bits, n_rec)] = entry
        print(name, json.dumps(entry), flush=True)
        with open(args.out, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
