#!/usr/bin/env python3
"""Golden fixtures of the BBHash half of the competitor harness, written by the REFERENCE's own BooPHF (external/BooPHF.hpp compiled from
/root/reference into oracle/_ref/bbhash_ref by oracle/Makefile; only the 8-line hasher around it is restated, see oracle/bbhash_ref.cpp).
Runs in the build container only.  Each case: <name>.bbh (boomphf::mphf::save), <name>.npz with the indexed records, the query records
(members, non-members, records with non-ACGT bases, a record shorter than k) and the reference's lookup() of every window of k valid bases."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import random_records  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "bbhash_ref")
CASES = [("k31_b64_g1", 31, 64, 1.0, 12, 200, 2500), ("k31_b128_g2", 31, 128, 2.0, 10, 200, 2500), ("k63_b128_g1", 63, 128, 1.0, 8, 300, 2000),
         ("k13_b64_g1_deep", 13, 64, 1.0, 30, 100, 1500)]


def fasta(path, recs):
    with open(path, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">r%d\n%s\n" % (i, r))


def cat(arrs):
    off = np.zeros(len(arrs) + 1, dtype=np.int64)
    for i, a in enumerate(arrs):
        off[i + 1] = off[i] + len(a)
    return (np.concatenate([np.frombuffer(a, dtype=np.uint8) for a in arrs]) if arrs else np.zeros(0, dtype=np.uint8)), off


def main():
    out_dir = os.path.join(HERE, "bbhash")
    os.makedirs(out_dir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)  # BooPHF writes its level files into the working directory
        for ci, (name, k, bits, gamma, n, lo, hi) in enumerate(CASES):
            recs, seen = [], set()
            for r in random_records(4000 + ci, n, lo, hi):  # BBHash needs distinct keys
                ks = {r[i:i + k] for i in range(len(r) - k + 1)}
                if len(ks) == len(r) - k + 1 and not (ks & seen):
                    seen |= ks
                    recs.append(r)
            fa, bbh = os.path.join(tmp, "i.fa"), os.path.join(out_dir, name + ".bbh")
            fasta(fa, recs)
            subprocess.run([REF, "build", fa, str(k), str(bits), str(gamma), bbh], check=True, capture_output=True)
            q = bytearray(recs[1])
            q[10] = q[len(q) // 2] = ord("N")
            queries = list(recs) + random_records(5000 + ci, 3, k, 600) + [bytes(q), recs[0][: k - 1], recs[0][:k]]
            qa, codes = os.path.join(tmp, "q.fa"), os.path.join(tmp, "c.u64")
            fasta(qa, queries)
            subprocess.run([REF, "query", bbh, str(k), str(bits), qa, codes], check=True, capture_output=True)
            rec_flat, rec_off = cat(recs)
            q_flat, q_off = cat(queries)
            np.savez_compressed(os.path.join(out_dir, name + ".npz"), k=k, kmer_bits=bits, gamma=gamma, records=rec_flat, rec_off=rec_off,
                                queries=q_flat, q_off=q_off, codes=np.fromfile(codes, dtype=np.uint64))
            print(name, "keys", sum(len(r) - k + 1 for r in recs), "file", os.path.getsize(bbh), "bytes")


if __name__ == "__main__":
    main()
