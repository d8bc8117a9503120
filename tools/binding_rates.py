#!/usr/bin/env python3
"""The reference-side binding (examples/query_example*: gpu_mphf<> under the reference's query driver) on plain FASTA and .fa.gz of the bench's C2 records, by
transport of the streaming result: codes (8 B per k-mer over PCIe, ABI 3's way), runs expanded on the host (the default), runs handed out as runs (--runs);
one GPU and --devices 0,0 (two lanes).  Prints the 5-field CSV lines of every run: ns per k-mer, streaming then random pass.

    python tools/binding_rates.py [--records 300] [--reps 3]
"""
import argparse
import gzip
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=300)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cache-dir", default=os.environ.get("LPHQ_BENCH_CACHE", "/tmp/lphq_bench_cache"))
    args = ap.parse_args()
    import bench
    import lphash_amd as L
    from lphash_amd.synth import synthetic_records

    wl = bench.make_workload("x", "stream", "p", 31, 20, 100_000_000, 0, 1, 0, 20, True)
    os.makedirs(args.cache_dir, exist_ok=True)
    path = bench.index_path(args.cache_dir, wl)
    flat, _ = synthetic_records(seed=1, n_records=wl["n_records"], length=wl["rec_len"], views=False)
    recs = bench.RecordSet(wl["n_records"], wl["rec_len"], flat=flat)
    if not os.path.exists(path):
        L.build(recs, 31, 20, path + ".tmp", kind=L.PARTITIONED, kmer_bits=64, threads=0, device=0)
        os.replace(path + ".tmp", path)
    n = min(args.records, wl["n_records"])
    fa = os.path.join(args.cache_dir, f"binding_seed1_r{n}.fa")
    if not os.path.exists(fa):
        with open(fa + ".tmp", "wb") as f:
            for i in range(n):
                f.write(b">s%d\n" % i)
                f.write(recs[i])
                f.write(b"\n")
        os.replace(fa + ".tmp", fa)
    gz = fa + ".gz"
    if not os.path.exists(gz) and n <= 600:
        with open(fa, "rb") as fi, gzip.open(gz + ".tmp", "wb", compresslevel=6) as fo:
            fo.write(fi.read())
        os.replace(gz + ".tmp", gz)
    exe = os.path.join(ROOT, "examples", "query_example64")
    for file in [fa] + ([gz] if os.path.exists(gz) else []):
        for label, env, extra in (("codes over PCIe", {"LPHQ_SESSION_TRANSPORT": "codes"}, []), ("runs over PCIe, expanded on the host", {"LPHQ_SESSION_TRANSPORT": "runs"}, []),
                                  ("runs over PCIe, handed out as runs", {}, ["--runs"]), ("runs, expanded, two lanes", {"LPHQ_SESSION_TRANSPORT": "runs"}, ["--devices", "0,0"]),
                                  ("runs as runs, two lanes", {}, ["--runs", "--devices", "0,0"])):
            vals = []
            for _ in range(args.reps):
                r = subprocess.run([exe, "query-p", "-i", path, "-q", file] + extra, capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
                f5 = r.stdout.strip().split(",")
                if r.returncode != 0 or len(f5) != 5:
                    print("FAILED", label, r.returncode, r.stderr[-300:])
                    break
                vals.append((float(f5[3]), float(f5[4])))
            if vals:
                vals.sort()
                print(f"{os.path.basename(file):>28}  {label:<40} streaming {vals[len(vals) // 2][0]:.4f} ns/k-mer (all: {[round(v[0], 4) for v in vals]})  random {vals[len(vals) // 2][1]:.4f}", flush=True)


if __name__ == "__main__":
    main()
