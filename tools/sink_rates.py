#!/usr/bin/env python3
"""Kernel times of the three sinks of the streaming pass on bench.py's synthetic sets: codes (lphq_query_records), runs (lphq_query_records_runs) and
payload (lphq_query_records_gather, 1 / 8-byte elements), HIP events around each launch, with the runs expanded and compared once.

    python tools/sink_rates.py [--k 31 --m 20 --kmers 100000000] [--steps 10]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--m", type=int, default=20)
    ap.add_argument("--kind", default="p")
    ap.add_argument("--kmers", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--read-len", type=int, default=0, help="query the index records cut into reads of this many bases (every tile multi-record) instead of whole")
    ap.add_argument("--cache-dir", default=os.environ.get("LPHQ_BENCH_CACHE", "/tmp/lphq_bench_cache"))
    args = ap.parse_args()
    import numpy as np
    import torch

    import bench
    import lphash_amd as L
    from lphash_amd.synth import synthetic_records

    dev = torch.device("cuda", 0)
    wl = bench.make_workload("x", "stream", args.kind, args.k, args.m, args.kmers, 0, args.steps, 2, 20, True)
    k, n_records, rec_len, nk, bits = wl["k"], wl["n_records"], wl["rec_len"], wl["kmers"], wl["kmer_bits"]
    flat, _ = synthetic_records(seed=1, n_records=n_records, length=rec_len, views=False)
    recs = bench.RecordSet(n_records, rec_len, flat=flat)
    os.makedirs(args.cache_dir, exist_ok=True)
    path = bench.index_path(args.cache_dir, wl)
    kind = L.PARTITIONED if args.kind == "p" else L.UNPARTITIONED
    if not os.path.exists(path):
        t0 = time.time()
        L.build(recs, k, args.m, path + ".tmp", kind=kind, kmer_bits=bits, threads=0, device=0)
        os.replace(path + ".tmp", path)
        print(f"built {path} in {time.time() - t0:.1f}s", file=sys.stderr)
    h = (L.Mphf if kind == L.PARTITIONED else L.MphfAlt).load(path, kmer_bits=bits, device=0)
    d_seq = recs.to_device(torch, dev)
    total = int(d_seq.numel())
    if args.read_len:  # the same bases as reads: pieces of read_len bases of the records, back to back (all members; windows across two pieces are not asked)
        assert args.read_len >= k and rec_len >= args.read_len
        per = rec_len // args.read_len
        d_seq = d_seq.view(n_records, rec_len)[:, : per * args.read_len].contiguous().view(-1)
        total = int(d_seq.numel())
        n_records, rec_len = n_records * per, args.read_len
    wpr = rec_len - k + 1
    n_out = n_records * wpr
    seq_off = torch.arange(0, n_records + 1, dtype=torch.int64, device=dev) * rec_len
    out_off = torch.arange(0, n_records + 1, dtype=torch.int64, device=dev) * wpr
    stream = torch.cuda.current_stream().cuda_stream
    W = k - args.m + 1
    arenas = L._lib.RUN_ARENAS
    per_arena = int(n_out * (2.0 / (W + 1) + 1 / 400.0) * 1.15 / arenas) + 4096
    d_runs = torch.empty(arenas * per_arena * 2, dtype=torch.int64, device=dev)
    d_xc = torch.empty(arenas * per_arena, dtype=torch.int64, device=dev)
    d_counts = torch.empty(arenas * 16, dtype=torch.int64, device=dev)
    d_out = torch.empty(n_out, dtype=torch.int64, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    t1 = torch.randint(0, 250, (nk,), generator=g, device=dev, dtype=torch.int64).to(torch.uint8)
    t8 = torch.randint(0, 1 << 40, (nk,), generator=g, device=dev, dtype=torch.int64)
    p1 = torch.empty(n_out, dtype=torch.uint8, device=dev)
    p8 = torch.empty(n_out, dtype=torch.int64, device=dev)

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs]))

    res = {"k": k, "m": args.m, "kind": args.kind, "kmers": nk, "read_len": args.read_len, "records": n_records, "windows": n_out, "image_MB": h.info.image_bytes / 1e6}
    res["codes_ms"] = timed(lambda: h.query_records(d_seq.data_ptr(), total, seq_off.data_ptr(), n_records, d_out.data_ptr(), out_off.data_ptr(), True, stream))
    res["runs_ms"] = timed(lambda: h.query_records_runs(d_seq.data_ptr(), total, seq_off.data_ptr(), n_records, out_off.data_ptr(), d_runs.data_ptr(), arenas * per_arena,
                                                        d_xc.data_ptr(), arenas * per_arena, d_counts.data_ptr(), stream))
    cnt = d_counts.cpu().numpy().reshape(arenas, 16)
    res["runs"], res["explicit_codes"], res["arena_fill_max"] = int(cnt[:, 0].sum()), int(cnt[:, 1].sum()), float(cnt[:, 0].max() / per_arena)
    res["windows_per_run"] = n_out / max(res["runs"], 1)
    d_exp = torch.empty(n_out, dtype=torch.int64, device=dev)
    d_st = torch.zeros(1, dtype=torch.int64, device=dev)
    res["expand_ms"] = timed(lambda: h.expand_runs(d_runs.data_ptr(), arenas * per_arena, d_xc.data_ptr(), arenas * per_arena, d_counts.data_ptr(), d_exp.data_ptr(), n_out,
                                                   d_st.data_ptr(), stream))
    res["expand_equals_codes"] = bool(torch.equal(d_exp, d_out)) and int(d_st.item()) == 0
    res["gather1_ms"] = timed(lambda: h.query_records_gather(d_seq.data_ptr(), total, seq_off.data_ptr(), n_records, out_off.data_ptr(), t1.data_ptr(), 1, p1.data_ptr(), 255, stream))
    res["gather1_ok"] = bool(torch.equal(p1, t1[d_out]))
    res["gather8_ms"] = timed(lambda: h.query_records_gather(d_seq.data_ptr(), total, seq_off.data_ptr(), n_records, out_off.data_ptr(), t8.data_ptr(), 8, p8.data_ptr(), 0, stream))
    res["gather8_ok"] = bool(torch.equal(p8, t8[d_out]))
    two = torch.empty(n_out, dtype=torch.uint8, device=dev)
    res["codes_then_gather1_ms"] = res["codes_ms"] + timed(lambda: h.gather_payload(d_out.data_ptr(), n_out, t1.data_ptr(), 1, two.data_ptr(), 255, stream))
    for key in ("codes", "runs", "gather1", "gather8"):
        res[key + "_Gkmers_per_s"] = n_out / res[key + "_ms"] / 1e6
    print(json.dumps(res))


if __name__ == "__main__":
    main()
