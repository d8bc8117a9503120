#!/usr/bin/env python3
"""Fixtures derived from the reference's BUNDLED REAL DATA (read as data from /root/reference/data; reference README.md:168-201), generated
with the CPU oracle (oracle/lph_oracle.c) in the build container.  The GPU box has no /root/reference: these files are how real genomic
input -- unitigs of a Salmonella pangenome with their minimizer skew and 0.29 % of the k-mers under colliding minimizers, a genome whose
k-mers are mostly members, a genome of another species with runs of N, sequencing reads with N -- reaches the HIP path there.

Written under tests/golden/real/:
  inputs.npz          ust_k31_2bit / ust_k31_len   data/unitigs_stitched/se.ust.k31.fa.gz: all 611 records, 2 bits per base (pure ACGT), 4,951,824 bases
                      ust_k63_2bit / ust_k63_len   data/unitigs_stitched/se.ust.k63.fa.gz: its first records up to ~1.5 Mbases
                      q_salmonella                 data/queries/salmonella_enterica.fasta.gz: bases [0, 300,000) of its record
                      q_ecoli_a, q_ecoli_b         data/queries/ecoli1.fasta.gz: bases [330,000, 560,000) and [995,000, 1,030,000): 24 runs of N
                      q_fastq / q_fastq_off        data/queries/SRR5833294.10K.fastq.gz: the first 2,000 reads + every later read with an N
  se_k31_m20_b64_p.lph  BASELINE.json configs[0] ("C1") as configured: build-p -k 31 -m 20 on se.ust.k31, written by the oracle builder
  expected.json       per case: size + SHA-256 of the oracle-built file, census, and per query set the SHA-256 of the codes (u64 LE) of
                      MPHF::operator()(streaming=false) and of the streaming pass the product defines (one code per window
                      of k valid bases == the random pass restricted to those windows), with the number of windows of each
  codes_<case>.npz    the first HEAD codes of the random pass per query set (a mismatch can be located, not only detected)

Streaming vs random on clean input (ACGT only) is asserted equal HERE with the oracle's own streaming branch (partitioned_mphf.hpp:78-184);
where the input holds non-ACGT bases the reference's streaming branch carries a stale min_pos across the break (SURVEY 8a note 4), which
the product deliberately does not reproduce: expected streaming there is the random pass on the valid windows.

    python tests/golden/make_real_golden.py          (about a minute; needs /root/reference/data)
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import orc  # noqa: E402
from real_fixture import CASES, HEAD, OUT, pack_2bit, query_sets, unitig_records, valid_windows  # noqa: E402

DATA = "/root/reference/data"


def fasta_records(path):
    recs, cur = [], None
    with gzip.open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if cur is not None:
                    recs.append(b"".join(cur))
                cur = []
            elif cur is not None:
                cur.append(line.rstrip(b"\r\n"))
    if cur is not None:
        recs.append(b"".join(cur))
    return recs


def fastq_reads(path):
    with gzip.open(path, "rb") as f:
        lines = f.read().split(b"\n")
    return [ln.rstrip(b"\r") for ln in lines[1::4] if ln]


def kmer_keys(rec, k):
    """(windows, 2) uint64: the k-mer of every window as (high, low) words, 2 bits per base, non-ACGT read as A"""
    from real_fixture import _CODE
    c = _CODE[np.frombuffer(rec, dtype=np.uint8)].astype(np.uint64)
    n = len(c) - k + 1
    out = np.zeros((max(n, 0), 2), dtype=np.uint64)
    for j in range(k):
        sh = 2 * (k - 1 - j)
        out[:, sh // 64] |= c[j:j + n] << np.uint64(sh % 64)
    return out[:, ::-1].copy()


def count_members(index_keys, query_keys):
    """how many rows of query_keys are rows of index_keys (which must be distinct); one sort of both together"""
    keys = np.concatenate([index_keys, query_keys])
    tag = np.concatenate([np.zeros(len(index_keys), dtype=np.uint8), np.ones(len(query_keys), dtype=np.uint8)])
    order = np.lexsort((tag, keys[:, 1], keys[:, 0]))  # by (high, low), index rows before query rows of the same k-mer
    sk, st = keys[order], tag[order]
    first = np.concatenate([[True], (sk[1:] != sk[:-1]).any(axis=1)])
    assert int((first & (st == 0)).sum()) == len(index_keys), "repeated k-mer in the index set (README.md:112-116)"
    group = np.cumsum(first) - 1
    has_index = st[first] == 0
    return int((has_index[group] & (st == 1)).sum())


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def main():
    os.makedirs(OUT, exist_ok=True)
    u31 = fasta_records(os.path.join(DATA, "unitigs_stitched", "se.ust.k31.fa.gz"))
    u63_all = fasta_records(os.path.join(DATA, "unitigs_stitched", "se.ust.k63.fa.gz"))
    u63, tot = [], 0
    for r in u63_all:
        if tot >= 1_500_000:
            break
        u63.append(r)
        tot += len(r)
    sal = fasta_records(os.path.join(DATA, "queries", "salmonella_enterica.fasta.gz"))[0]
    eco = fasta_records(os.path.join(DATA, "queries", "ecoli1.fasta.gz"))[0]
    reads = fastq_reads(os.path.join(DATA, "queries", "SRR5833294.10K.fastq.gz"))
    assert len(reads) == 10_000
    picked = reads[:2000] + [r for r in reads[2000:] if b"N" in r]
    fq_off = np.concatenate([[0], np.cumsum([len(r) for r in picked])]).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "inputs.npz"),
                        ust_k31_2bit=pack_2bit(b"".join(u31)), ust_k31_len=np.array([len(r) for r in u31], dtype=np.int32),
                        ust_k63_2bit=pack_2bit(b"".join(u63)), ust_k63_len=np.array([len(r) for r in u63], dtype=np.int32),
                        q_salmonella=np.frombuffer(sal[:300_000], dtype=np.uint8),
                        q_ecoli_a=np.frombuffer(eco[330_000:560_000], dtype=np.uint8), q_ecoli_b=np.frombuffer(eco[995_000:1_030_000], dtype=np.uint8),
                        q_fastq=np.frombuffer(b"".join(picked), dtype=np.uint8), q_fastq_off=fq_off)
    z = np.load(os.path.join(OUT, "inputs.npz"))
    assert unitig_records(z, "ust_k31") == u31 and unitig_records(z, "ust_k63") == u63  # the packing round-trips
    qs = query_sets(z)
    expected = {"source": "reference data/ (README.md:168-201), read as data; codes and files by oracle/lph_oracle.c", "head": HEAD, "cases": {}}
    for name, (kind, k, m, bits, ust) in CASES.items():
        recs = unitig_records(z, ust)
        path = os.path.join(OUT, name + ".lph")
        orc.build(recs, k, m, path, kind=kind, kmer_bits=bits)
        o = orc.Index(path, kind, bits)
        n = sum(len(r) - k + 1 for r in recs if len(r) >= k)
        assert o.nkmers == n
        # the reference's --check (src/build.cpp:53-70) on the real set: bijection onto [0, n), streaming == random
        seen = np.zeros(n, dtype=np.uint8)
        for r in recs:
            if len(r) < k:
                continue
            s = o(r, True)
            assert np.array_equal(s, o(r, False)) and int(s.max()) < n and not seen[s].any()
            seen[s] = 1
        assert seen.all()
        entry = {"kind": kind, "k": k, "m": m, "kmer_bits": bits, "unitigs": ust, "lph_bytes": os.path.getsize(path), "lph_sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(),
                 "nkmers": int(o.nkmers), "distinct_minimizers": int(o.distinct_minimizers), "fallback_kmers": int(o.fallback_keys), "queries": {}}
        heads = {}
        index_keys = np.concatenate([kmer_keys(r, k) for r in recs if len(r) >= k])
        for qname, qrecs in qs.items():
            rnd = [o(q, False) for q in qrecs]
            val = [valid_windows(q, k) for q in qrecs]
            stream = [r[v] for r, v in zip(rnd, val)]
            clean_same = all(np.array_equal(o(q, True), r) for q, r, v in zip(qrecs, rnd, val) if v.all())  # the oracle's streaming branch on ACGT-only records
            assert clean_same, (name, qname)
            flat_r = np.concatenate(rnd) if rnd else np.zeros(0, dtype=np.uint64)
            flat_s = np.concatenate(stream) if stream else np.zeros(0, dtype=np.uint64)
            entry["queries"][qname] = {"records": len(qrecs), "windows": int(flat_r.size), "valid_windows": int(flat_s.size), "members": count_members(index_keys, np.concatenate([kmer_keys(q, k)[v] for q, v in zip(qrecs, val) if len(q) >= k])),
                                       "random_sha256": sha(flat_r), "streaming_sha256": sha(flat_s)}
            heads[qname] = flat_r[:HEAD]
        np.savez_compressed(os.path.join(OUT, f"codes_{name}.npz"), **heads)
        if name != "se_k31_m20_b64_p":
            os.remove(path)  # only C1's file is committed; the others are pinned by digest and rebuilt where they are needed
        expected["cases"][name] = entry
        print(name, json.dumps({k_: v for k_, v in entry.items() if k_ != "queries"}), {q: (e["windows"], e["valid_windows"], e["members"]) for q, e in entry["queries"].items()}, flush=True)
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump(expected, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
