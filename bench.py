#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native LPHash query engine.

Headline workload (BASELINE.json configs[1], "C2"): partitioned LP-MPHF, k=31 m=20, 100 M k-mers; one *step* =
one batch of 2^28 random positive packed k-mer lookups per GPU (the reference's `streaming == false`
branch, include/partitioned_mphf.hpp:185-195), inputs and outputs resident in HBM.  `--mode stream`
measures C3 (all windows of the index FASTA, streaming kernel) instead.

After the (unchanged) headline the default run measures the other BASELINE.json configurations the same way and
reports them in a `secondary` list of the same JSON line: C3 (streaming, same index), C5 (unpartitioned), and the
per-GPU work of C4 (k=63, 128-bit k-mers, 2 B k-mers) in random and in streaming mode -- each entry with its kernel
time (HIP events), value, roofline, parity against the oracle and the oracle timed beside it.  `--no-secondary`
skips them; a secondary that fails is reported in `secondary_errors` and never takes the headline with it (a parity failure
in one still prints the headline line, then ends the job with exit code 3).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU: the index is built once on rank 0, its flat device image is broadcast with RCCL
(torch.distributed backend "nccl"), and every rank processes its own batch -- no data-path collective
 => weak scaling; value = lookups of all ranks / max-over-ranks time.  Every rank checks a
slice of ITS OWN timed batch against the oracle, the verdicts are all-reduced, and a failure on any rank ends the
whole job with exit code 3.  Rank 0 prints ONE JSON line.  The headline stays the weak line (N = 1 equals the
single-GPU bench); at N > 1 every workload also carries a `strong` entry: a FIXED job (2^31 probes of one seeded set;
all windows of the index records) sharded over the ranks by lphash_amd/dist.py's shard_range / shard_records, value =
total units / max-over-ranks time -- "query batches sharded across the GPUs" as BASELINE.json's C4 words it.

`--plan-only` (no GPU needed; gloo when WORLD_SIZE > 1) stops before the first GPU call: it resolves the workloads
and cache paths, checks that every rank resolved the same plan and prints it -- what tests/test_dist_cpu.py runs.
"""
import argparse
import gc
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def device_source_sha():
    """sha256 (first 16 hex digits) over the device sources of the QUERY kernels: what a counter pass in profiles/pmc_traffic.json (and the ISA listing in
    profiles/isa_rest.json) is evidence for.  The construction kernels (extract_device.cuh, search_device.cuh; extract.hip, search.hip) are not part of it: a
    change there cannot move a query kernel's counters.  (Until round 5 the two construction headers were hashed along: the sha of the same query sources
    was 2767c5d441027723 then.)"""
    import glob
    import hashlib

    hsh = hashlib.sha256()
    csrc = os.path.join(ROOT, "lphash_amd", "csrc")
    for pth in sorted([os.path.join(csrc, f) for f in ("kernels_device.cuh", "device_query.cuh", "stream8_device.cuh", "stream_sinks.h", "kernels.hip", "stream8.hip", "image_layout.h")] +
                      glob.glob(os.path.join(csrc, "gfx950", "*.cuh"))):
        hsh.update(os.path.basename(pth).encode())
        hsh.update(open(pth, "rb").read())
    return hsh.hexdigest()[:16]


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable
# VALU issue peak in lane-instructions/s: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (one wave64 instruction = 4 cycles of a SIMD)
VALU_PEAK_LANE_INSTR = 256 * 4 * 16 * 2.4e9
RECORD_KMERS = 100_000  # k-mers per synthetic record
# Line requests per second the fabric behind the L2s sustains for scattered 128-B gathers, and the rate of the bare dependent chain of a lookup (pilot word ->
# block line): re-measured on the box by tools/gather_ceiling.hip with the section sizes of the image at hand (`bench.py --probe-ceiling` ->
# profiles/gather_ceiling.json) -- an INDEPENDENT ceiling.  The constant below is only the fallback when that file is absent or was taken with another version
# of the probe: it is the k=31 random kernel's own best observed line rate (r04 counters x r04 kernel time), so a `frac_line_requests` against it is ~1 by
# construction and the line says "inferred".
FABRIC_LINE_PEAK_FALLBACK = 57.9e9
RUN_ARENAS = 32  # LPHQ_RUN_ARENAS (include/lphq.h)
# single-GPU rates (units/s) of the workloads as measured in round 5 (profiles/r05_final_bench_default.json): what the strong-scaling plan sizes its fixed
# job with (a shard must be long against launch + barrier cost), never a reported number
NOMINAL_RATE = {"C2": 38e9, "C3": 172e9, "C5": 38.8e9, "C4-random": 23.4e9, "C4-stream": 265e9}
STRONG_STREAM_UNITS_LOG2 = 32  # the fixed job of a streaming workload: whole passes over the index records adding up to >= 2^32 windows
STRONG_CHUNK_LOG2 = 24  # the fixed probe set of the strong-scaling line is drawn in chunks of 2^24 probes, chunk c from seed 5000 + c: any rank can draw any part of it


def probe_sha():
    """version of the ceiling probe: sha256 (16 hex digits) over tools/gather_ceiling.hip and the image layout whose section sizes it is run with"""
    hsh = hashlib.sha256()
    for pth in (os.path.join(ROOT, "tools", "gather_ceiling.hip"), os.path.join(ROOT, "lphash_amd", "csrc", "image_layout.h")):
        hsh.update(open(pth, "rb").read())
    return hsh.hexdigest()[:16]


def image_sections(info, W, partitioned):
    """(small, big) bytes of an image as the lookup chain sees it: the pilot side (codeword arrays + hashed-pilot dictionaries + header) and the line side
    (fused / pair blocks + tail table), from lphq_get_info alone (image_layout.h: one 128-B block per 64 minimizers, 32 when W > 15; 64-B pair blocks of 32)."""
    M, T = int(info.distinct_minimizers), int(info.minimizer_table_size)
    if partitioned:
        per = 32 if W > 15 else 64
        blocks = ((M + per - 1) // per + 1) * 128
    else:
        blocks = ((M + 31) // 32 + 1) * 64
    big = blocks + (T - M) * 8
    # the pilot side a lookup touches: not the u32 free-slot table of the minimizer function ((T - M) * 4 bytes in the image; tail[] answers those positions)
    return max(int(info.image_bytes) - big - (T - M) * 4, 4096), big


def ceiling_for(wl_key):
    """The committed probe results for one index ('p|k31|m20|n100000000'), or None when profiles/gather_ceiling.json is absent / taken with another probe."""
    try:
        cj = json.load(open(os.path.join(ROOT, "profiles", "gather_ceiling.json")))
    except (OSError, ValueError):
        return None
    if cj.get("probe_sha") != probe_sha():
        return None
    e = cj.get("entries", {}).get(wl_key)
    return dict(e, source="profiles/gather_ceiling.json (tools/gather_ceiling.hip, bench.py --probe-ceiling)") if e else None


def sha256_file(path):
    hsh = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            hsh.update(blk)
    return hsh.hexdigest()


def pinned_index_digest(wl):
    """The SHA-256 the ORACLE builder's file has for this synthetic set (tests/golden/full_size_digests.json, made by tools/full_size_digests.py in the
    build container: C2 / C5 at 100 M, C4 at 2 B k-mers), or None when the set is not one of those."""
    try:
        table = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size_digests.json")))
    except (OSError, ValueError):
        return None
    if wl["kind"] not in ("p", "u") or wl["index_set"] != "uniform":
        return None
    return table.get(f"{wl['kind']}_k{wl['k']}_m{wl['m']}_b{wl['kmer_bits']}_seed1_{wl['n_records']}x{wl['rec_len']}")


def strong_plan(wl, world, strong_total_log2):
    """What every rank does in the STRONG-scaling pass of a workload (fixed total work, split without overlap): random mode = a fixed set of
    2^strong_total_log2 probes cut by shard_range; streaming = the index records cut by shard_records (contiguous, by base count).  Pure function of
    the plan: [(begin, end)] per rank, in units of probes / records."""
    from lphash_amd.dist import shard_range, shard_records

    if wl["mode"] == "random":
        total = 1 << strong_total_log2
        return {"unit": "probes", "total": total, "shards": [list(shard_range(total, r, world)) for r in range(world)]}
    # streaming: the fixed job is `passes` whole passes over the index records laid end to end (one index pass is 0.5 ms of one GPU at k = 31: cut eight
    # ways that would be launch overhead, not scaling -- VERDICT r05 item 4); a shard is a contiguous range of that list of passes x records
    wpr = wl["rec_len"] - wl["k"] + 1
    passes = min(64, max(1, -(-(1 << STRONG_STREAM_UNITS_LOG2) // (wl["n_records"] * wpr))))  # (at most 64: a toy index is not made a 2^32-window job)
    lens = [wl["rec_len"]] * (wl["n_records"] * passes)
    return {"unit": "records", "passes": passes, "total": wl["n_records"] * passes, "shards": [list(shard_records(lens, r, world)) for r in range(world)]}


def strong_shard_ms(wl, sp, rank):
    """What a rank's shard of the strong-scaling job costs one GPU, by the round-5 single-GPU rates (NOMINAL_RATE): the plan's own estimate, so that a
    test can hold it against launch + barrier cost (profiles/r06_launch_barrier_cost.json)"""
    lo, hi = sp["shards"][rank]
    units = (hi - lo) if sp["unit"] == "probes" else (hi - lo) * (wl["rec_len"] - wl["k"] + 1)
    name = config_name(wl)
    rate = NOMINAL_RATE.get(name + ("-random" if name == "C4" and wl["mode"] == "random" else "-stream" if name == "C4" else ""), NOMINAL_RATE["C2"])
    return units / rate * 1e3


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ----------------------------------------------------------------------------------------------------------
# the plan: pure functions of the arguments (no GPU, no torch) -- every rank must resolve the same one
# ----------------------------------------------------------------------------------------------------------
def make_workload(name, mode, kind, k, m, kmers, batch_log2, steps, warmup, cpu_sample_log2, gpu_build, nonmember_frac=0.0,
                  shuffle_records=False, bbhash_gamma=1.0, headline=False, index_set="uniform", sink="codes"):
    n_records = max(1, kmers // RECORD_KMERS)
    return {"name": name, "mode": mode, "kind": kind, "k": k, "m": m, "kmers": n_records * RECORD_KMERS, "n_records": n_records,
            "rec_len": RECORD_KMERS + k - 1, "kmer_bits": 64 if k <= 31 else 128, "batch_log2": batch_log2, "steps": steps, "warmup": warmup,
            "cpu_sample_log2": cpu_sample_log2, "gpu_build": bool(gpu_build), "nonmember_frac": nonmember_frac,
            "shuffle_records": bool(shuffle_records), "bbhash_gamma": bbhash_gamma, "headline": headline, "index_set": index_set,
            # what the streaming pass leaves behind: "codes" (8 B per k-mer, the BASELINE metric), "runs" (lphq_query_records_runs: 16 B per super-k-mer) or
            # "gather<E>" (lphq_query_records_gather: table[code], E-byte elements; the codes never reach memory)
            "sink": sink}


def config_name(wl):
    if wl["kind"] == "bbhash":
        return "BBHash competitor (SURVEY 8f-4)"
    if wl["kind"] == "pthash":
        return "PTHash competitor (SURVEY 8f-4)"
    if wl["kind"] == "u":
        return "C5"
    return "C4" if wl["k"] > 31 else ("C2" if wl["mode"] == "random" else "C3")


def index_path(cache_dir, wl):
    g = wl["bbhash_gamma"] if wl["kind"] == "bbhash" else ""
    d = "" if wl["index_set"] == "uniform" else "_" + wl["index_set"]
    return os.path.join(cache_dir, f"synth_seed1{d}_{wl['kind']}{g}_k{wl['k']}_m{wl['m']}_n{wl['kmers']}_b{wl['kmer_bits']}.lph")


def headline_workload(args):
    wl = make_workload("headline", args.mode, args.kind, args.k, args.m, args.kmers, args.batch_log2, args.steps, args.warmup, args.cpu_sample_log2,
                       args.gpu_build, args.nonmember_frac, args.shuffle_records, args.bbhash_gamma, headline=True, index_set=args.index_set,
                       sink=args.sink if args.mode == "stream" else "codes")
    wl["name"] = config_name(wl) + ("" if wl["sink"] == "codes" else "-" + wl["sink"])
    return wl


def is_default_headline(wl):
    return (wl["mode"], wl["kind"], wl["k"], wl["m"], wl["kmers"], wl["batch_log2"], wl["nonmember_frac"], wl["index_set"], wl["sink"]) == ("random", "p", 31, 20, 100_000_000, 28, 0.0, "uniform", "codes")


def secondary_workloads(args, head):
    """The other BASELINE.json configurations, measured after the headline of a default run (C2): same steps / warmup, a smaller oracle
    sample (k=63 costs the oracle 4x more per k-mer), C4's 2 B-k-mer index built with the GPU pilot search (its bytes equal the
    sequential build's: tests/test_zz_gpu_next.py; the sequential search would take minutes)."""
    if args.no_secondary or args.no_cpu_baseline or not is_default_headline(head):  # (--no-cpu-baseline = profiling runs: the named workload only)
        return []
    st, wu = args.steps, args.warmup
    c4 = args.secondary_c4_kmers
    return [make_workload("C3", "stream", "p", 31, 20, 100_000_000, 0, st, wu, 22, args.gpu_build),
            # the same pass with the result the reference's streaming branch computes before it expands it: runs; and fused with its consumer
            make_workload("C3-runs", "stream", "p", 31, 20, 100_000_000, 0, st, wu, 20, args.gpu_build, sink="runs"),
            make_workload("C5", "random", "u", 31, 20, 100_000_000, 28, st, wu, 22, args.gpu_build),
            make_workload("C4-random", "random", "p", 63, 20, c4, 28, st, wu, 21, True),
            make_workload("C4-stream", "stream", "p", 63, 20, c4, 0, st, wu, 21, True),
            make_workload("C4-stream-runs", "stream", "p", 63, 20, c4, 0, st, wu, 19, True, sink="runs"),
            make_workload("C4-stream-gather1", "stream", "p", 63, 20, c4, 0, st, wu, 19, True, sink="gather1"),
            # BASELINE.md section 3's secondary inputs: half the probes uniformly random non-members; the index records in shuffled order
            make_workload("C2-nonmember50", "random", "p", 31, 20, 100_000_000, 28, st, wu, 22, args.gpu_build, nonmember_frac=0.5),
            make_workload("C3-shuffled", "stream", "p", 31, 20, 100_000_000, 0, st, wu, 22, args.gpu_build, shuffle_records=True)]


def plan_of(args):
    head = headline_workload(args)
    wls = [head] + secondary_workloads(args, head)
    plan = {"gpus": args.gpus, "cache_dir": args.cache_dir, "workloads": wls, "index_paths": [index_path(args.cache_dir, w) for w in wls]}
    if args.gpus > 1 or args.dist_single or args.strong:
        plan["strong"] = [dict(strong_plan(w, args.gpus, strong_total_log2_of(args, w)), name=w["name"]) for w in wls if w["sink"] == "codes"]
        for sp_, w in zip(plan["strong"], [w for w in wls if w["sink"] == "codes"]):
            sp_["planned_shard_ms"] = [strong_shard_ms(w, sp_, r_) for r_ in range(args.gpus)]
    return plan


def strong_total_log2_of(args, wl):
    """log2 of the fixed probe set of the strong-scaling pass: what 8 GPUs answer in one step of the weak line (2^31 for the BASELINE workloads)"""
    return args.strong_total_log2 if args.strong_total_log2 > 0 else wl["batch_log2"] + 3


def algorithmic_bytes_per_lookup(h, mode, k, W, partitioned):
    """bytes of the reference's structures one lookup must touch, each word once at
    natural width, + the compulsory input/output stream; recomputed from this index's measured type mix.
    Partitioned: PTHash ranks word 8 + free-slot EF 26 * P(free) + wavelet tree 48 + EF by type (LEFT 26, RIGHT 34, NONE 60).
    Unpartitioned (query-u): PTHash 8 + 26 * P(free) + sizes.pair 34 + positions.diff 34."""
    info = h.info
    nk = info.num_kmers
    p_free = 1.0 - info.distinct_minimizers / info.minimizer_table_size
    if partitioned:
        f_l, f_r, f_max, f_n = (info.kmers_by_type[i] / nk for i in range(4))
        chain = 8 + 26 * p_free + 48 + 26 * f_l + 34 * f_r + 60 * f_n
        mix = {"maximal": round(f_max, 4), "left": round(f_l, 4), "right": round(f_r, 4), "none": round(f_n, 4)}
    else:
        chain = 8 + 26 * p_free + 34 + 34
        mix = None
    kmer_bytes = 8 if k <= 31 else 16
    if mode == "random":
        return kmer_bytes + 8 + chain, chain, mix
    return 1 + 8 + chain / ((W + 1) / 2.0), chain, mix


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process (this
    process never touches the GPU), relay rank 0's JSON line and exit with the children's code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    log("[bench] WORLD_SIZE unset and --gpus", n, "-> launching", " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_quota_cores():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota); None = unlimited or unknown.  The visible CPU count
    (sched_getaffinity) says how many threads may run, the quota how much CPU time they share."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = float(f.read())
        return None if quota <= 0 else quota / period
    except Exception:
        return None


def hbm_ceiling_gbs(torch, dev):
    """Practical HBM ceiling of this box, measured in-run: read-only pass and device-to-device copy over 2-GiB
    buffers (far beyond the 256-MiB Infinity Cache).  Returns (read GB/s, copy read+write GB/s)."""
    n = (2 << 30) // 8
    x = torch.ones(n, dtype=torch.int64, device=dev)
    y = torch.empty_like(x)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps / 1e3

    t_read = timed(lambda: x.sum())
    t_copy = timed(lambda: y.copy_(x))
    del x, y
    return n * 8 / t_read / 1e9, 2 * n * 8 / t_copy / 1e9


def gather_device(dist, torch, dev):
    """where the small tensors of an all_gather live: the GPU under RCCL; the host under gloo (its all_gather has no CUDA form -- the LPHQ_BENCH_SHARE_DEVICE0 test mode
    and --plan-only)"""
    return torch.device("cpu") if dist.get_backend() == "gloo" else dev


def plans_agree(dist, torch, dev, plan):
    """Every rank resolved the same plan (arguments, workloads, cache paths)?  all-gather of a digest; collective, same verdict on every rank."""
    digest = hashlib.sha256(json.dumps(plan, sort_keys=True).encode()).digest()
    mine = torch.tensor(list(digest[:16]), dtype=torch.int64, device=gather_device(dist, torch, dev))
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    return all(bool(torch.equal(e, mine)) for e in every)


def all_ranks_ok(dist, torch, dev, ok):
    """Logical AND of a per-rank verdict (collective).  A rank that fails a check does not leave on its own -- the others would wait
    for it in the next collective until the watchdog fires -- it reports here and every rank ends the job together."""
    if dist is None:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


class RecordSet:
    """The synthetic index records of one workload: n_records records of `length` bases.  With one rank they live in this process (one contiguous
    array; a record is cut out when somebody asks for it).  With several ranks on a node, rank 0 generates the set ONCE and writes it to the cache
    directory; the other ranks never hold it in host memory: they stream the file to their GPU in chunks (page cache is shared, not per-process) and
    pread the few records their oracle sample needs.  Eight ranks at C4 (2 GB of bases) used to cost 8 x (2 GB + 2 GB of per-record copies) and
    8 x the generation time."""
    CHUNK = 64 << 20

    def __init__(self, n_records, length, flat=None, path=None, source="generated"):
        self.n_records, self.length, self.flat, self.path, self.source = n_records, length, flat, path, source

    def __len__(self):
        return self.n_records

    def __getitem__(self, i):
        if i < 0 or i >= self.n_records:
            raise IndexError(i)
        if self.flat is not None:
            return self.flat[i * self.length:(i + 1) * self.length].tobytes()
        with open(self.path, "rb") as f:
            return os.pread(f.fileno(), self.length, i * self.length)

    def __iter__(self):
        return (self[i] for i in range(self.n_records))

    def chunks(self):
        """The bases in order, as uint8 arrays of at most CHUNK bytes (views of the array, or reads of the file into one reused buffer)."""
        import numpy as np

        total = self.n_records * self.length
        if self.flat is not None:
            for a in range(0, total, self.CHUNK):
                yield self.flat[a:a + self.CHUNK]
            return
        buf = np.empty(min(self.CHUNK, total), dtype=np.uint8)
        with open(self.path, "rb", buffering=0) as f:
            done = 0
            while done < total:
                n = f.readinto(memoryview(buf)[: min(self.CHUNK, total - done)])
                if not n:
                    raise IOError(f"{self.path} is shorter than {total} bytes")
                yield buf[:n]
                done += n

    def to_device(self, torch, dev):
        total = self.n_records * self.length
        d = torch.empty(total, dtype=torch.uint8, device=dev)
        at = 0
        for c in self.chunks():
            d[at:at + len(c)].copy_(torch.from_numpy(c))
            at += len(c)
        return d

    def checksum(self):
        """CRC-32 of the bases as this rank reads them (what the ranks compare in --plan-only --plan-records)."""
        import zlib

        acc = 0
        for c in self.chunks():
            acc = zlib.crc32(memoryview(c), acc)
        return acc


class BenchFailure(SystemExit):
    """job.fail: ends the process with `code` like sys.exit, and carries the reason (exit code 3 = GPU results differ from the oracle; 2 = set-up,
    e.g. a missing shared file in --cache-dir)."""

    def __init__(self, code, msg):
        super().__init__(code)
        self.msg = msg


class Job:
    """What the workloads of one bench process share."""

    def __init__(self, args, rank, local_rank, world, dev, dist, torch, np, L):
        self.args, self.rank, self.local_rank, self.world, self.dev, self.dist = args, rank, local_rank, world, dev, dist
        self.torch, self.np, self.L = torch, np, L
        self._records = {}  # (n_records, rec_len) -> (flat, recs): the k=31 workloads share one set
        self.hbm = None
        self.valu = {}

    def records(self, wl):
        from lphash_amd.synth import diverged_copy_records, synthetic_records

        key = (wl["n_records"], wl["rec_len"], wl["index_set"])
        if key not in self._records:
            self._records.clear()  # one set at a time (C4's is 2 GB)
            t0 = time.time()
            n, ln, which = key
            path = os.path.join(self.args.cache_dir, f"records_seed1_{n}x{ln}" + ("" if which == "uniform" else "_" + which) + ".u8") if self.world > 1 else None
            rs = None
            if self.rank == 0:
                if which == "uniform":
                    flat, _ = synthetic_records(seed=1, n_records=n, length=ln, views=False)
                else:  # a set with the repeat structure of a genome (lphash_amd/synth.py): tens of per cent of the k-mers behind colliding minimizers
                    flat, _ = diverged_copy_records(seed=1, n_records=n, length=ln, k=wl["k"], views=False)
                rs = RecordSet(n, ln, flat=flat, source="generated")
                log(f"[bench r{self.rank}] generated {n} records x {ln} bases in {time.time() - t0:.1f}s")
                if path is not None and not (os.path.exists(path) and os.path.getsize(path) == n * ln):
                    os.makedirs(self.args.cache_dir, exist_ok=True)
                    tmp = path + f".tmp{os.getpid()}"
                    with open(tmp, "wb") as f:
                        for c in rs.chunks():
                            f.write(memoryview(c))
                    os.replace(tmp, path)
                    log(f"[bench r{self.rank}] wrote {path} for the other ranks in {time.time() - t0:.1f}s")
            self.barrier()  # the file is complete
            if rs is None:
                if not (os.path.exists(path) and os.path.getsize(path) == n * ln):
                    self.fail(2, f"{path} is missing or short: the ranks of one node must share --cache-dir")
                rs = RecordSet(n, ln, path=path, source="shared_file")
            self._records[key] = rs
        return self._records[key]

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def fail(self, code, msg):
        log(f"[bench r{self.rank}] ERROR: {msg}")
        if self.dist is not None:
            try:
                self.dist.destroy_process_group()
            except Exception:
                pass
        raise BenchFailure(code, msg)


def load_oracle(job, wl, path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc  # the checker: never on the measured path

    if wl["kind"] == "pthash":
        return orc.PtHash(path, wl["k"], wl["kmer_bits"])
    return orc.Index(path, orc.PARTITIONED if wl["kind"] == "p" else orc.UNPARTITIONED, wl["kmer_bits"])


def strong_pass(job, wl, h, recs, d_seq, o, stream, rec_order):
    """The STRONG-scaling line of one workload (north_star: "query batches sharded across 8 x MI355X"): a FIXED job -- 2^strong_total_log2 probes of one
    common, seeded set (random mode) or every window of the index records (streaming) -- split over the ranks without overlap (lphash_amd/dist.py:
    shard_range / shard_records), each rank's shard resident in its HBM before the timed region, no data-path collective.  value = total units x steps /
    max-over-ranks time.  Checked on every rank: the first 2^18 codes of its shard against the oracle; streaming: the codes of ALL shards together are a
    permutation of [0, n) by two all-reduced checksums (sum and sum of squares, wrapping 64-bit, against their closed forms)."""
    args, rank, world, dev, dist, torch, np = job.args, job.rank, job.world, job.dev, job.dist, job.torch, job.np
    k, mode, kmer_bits = wl["k"], wl["mode"], wl["kmer_bits"]
    rec_len, n_records, nk = wl["rec_len"], wl["n_records"], wl["kmers"]
    steps, warmup = wl["steps"], wl["warmup"]
    sp = strong_plan(wl, world, strong_total_log2_of(args, wl))
    lo, hi = sp["shards"][rank]
    tag = f"[bench r{rank} {wl['name']} strong]"
    wpk = kmer_bits // 64
    if mode == "random":
        mine = hi - lo
        total_units = sp["total"]
        d_k = torch.empty(max(mine, 1) * wpk, dtype=torch.int64, device=dev)
        C = 1 << STRONG_CHUNK_LOG2
        for c in range(lo // C, (hi + C - 1) // C if mine else 0):  # chunk c of the common set: the same numbers on whichever rank draws it
            g = torch.Generator(device=dev)
            g.manual_seed(5000 + c)
            win = torch.randint(0, nk, (C,), generator=g, device=dev, dtype=torch.int64)
            a, b = max(lo, c * C), min(hi, (c + 1) * C)
            win = win[a - c * C:b - c * C]
            pos = (win // RECORD_KMERS) * rec_len + win % RECORD_KMERS
            h.pack_kmers(d_seq.data_ptr(), pos.data_ptr(), b - a, d_k.data_ptr() + (a - lo) * wpk * 8, stream)
            torch.cuda.synchronize()
            del win, pos
        d_o = torch.empty(max(mine, 1), dtype=torch.int64, device=dev)

        def step():
            if mine:
                h.query_packed(d_k.data_ptr(), mine, d_o.data_ptr(), stream)
    else:
        # records lo .. hi - 1 of the list "passes x all index records": whole and partial passes, one launch each (a batch list)
        wpr = rec_len - k + 1
        n_mine = hi - lo
        mine = n_mine * wpr
        total_units = sp["total"] * wpr
        pieces = []  # (first record, records, first window of the piece in this rank's output)
        at, r = 0, lo
        while r < hi:
            a_ = r % n_records
            n_ = min(hi - r, n_records - a_)
            pieces.append((a_, n_, at))
            at += n_ * wpr
            r += n_
        seq_off = torch.arange(0, n_records + 1, dtype=torch.int64, device=dev) * rec_len
        out_off = torch.arange(0, n_records + 1, dtype=torch.int64, device=dev) * wpr
        d_o = torch.empty(max(mine, 1), dtype=torch.int64, device=dev)

        def step():
            for a_, n_, at_ in pieces:
                h.query_records(d_seq.data_ptr() + a_ * rec_len, n_ * rec_len, seq_off.data_ptr(), n_, d_o.data_ptr() + at_ * 8, out_off.data_ptr(), True, stream)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    job.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    job.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    per_rank = [kernel_ms]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        me = torch.tensor([kernel_ms], dtype=torch.float64, device=gather_device(dist, torch, dev))
        every = [torch.zeros_like(me) for _ in range(world)]
        dist.all_gather(every, me)
        per_rank = [float(e.item()) for e in every]
    # what a step costs besides its kernel on this box: the same bracket (barrier, synchronize, events) around ONE minimal launch of the same entry point
    # (64 probes / one record) -- the yardstick the plan's shard sizes are held against (tests/test_dist_cpu.py, profiles/r06_launch_barrier_cost.json)
    if mode == "random":
        tiny = (lambda: h.query_packed(d_k.data_ptr(), min(mine, 64), d_o.data_ptr(), stream)) if mine else (lambda: None)
    else:
        tiny = (lambda: h.query_records(d_seq.data_ptr(), rec_len, seq_off.data_ptr(), 1, d_o.data_ptr(), out_off.data_ptr(), True, stream)) if mine else (lambda: None)
    tiny()
    torch.cuda.synchronize()
    job.barrier()
    t1 = time.perf_counter()
    for _ in range(steps):
        tiny()
    torch.cuda.synchronize()
    job.barrier()
    empty_step_ms = (time.perf_counter() - t1) / steps * 1e3
    for _ in range(1):  # (the tiny launch overwrote the head of d_o: the step once more before the checks)
        step()
    torch.cuda.synchronize()
    # parity: this rank's shard against the oracle (a sample), all shards together against the closed forms
    ok = True
    if o is not None and mine:
        S = min(mine, 1 << 18)
        if mode == "random":
            exp = o.query_packed(d_k[:S * wpk].cpu().numpy().view(np.uint64))
        else:
            nr = max(1, min(pieces[0][1], S // (rec_len - k + 1)))
            first = pieces[0][0]
            exp = np.concatenate([o(recs[(first + i) if rec_order is None else int(rec_order[first + i])], True) for i in range(nr)])
            S = len(exp)
        ok = bool(np.array_equal(d_o[:S].cpu().numpy().view(np.uint64), exp))
    sums_ok = None
    if mode == "stream" or wl["nonmember_frac"] == 0.0:
        v = d_o[:mine]
        inside = bool(mine == 0 or (int(v.min()) >= 0 and int(v.max()) < nk))
        ok = ok and inside
        if mode == "stream":  # every window of the set exactly once over all ranks: the codes are 0 .. n-1 in some order
            t = torch.zeros(2, dtype=torch.int64, device=dev)
            for a_ in range(0, mine, 1 << 26):  # (a chunk at a time: v * v of a 2 G-code shard is 16 GB)
                c_ = v[a_:a_ + (1 << 26)]
                t += torch.stack([c_.sum(), (c_ * c_).sum()])
            if dist is not None:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)  # (wrapping int64 addition)
            n, P = nk, sp["passes"]  # every pass covers [0, n) once
            want = [(P * (n * (n - 1) // 2)) % (1 << 64), (P * ((n - 1) * n * (2 * n - 1) // 6)) % (1 << 64)]
            got = [int(x) % (1 << 64) for x in t.tolist()]
            sums_ok = got == want
            ok = ok and sums_ok
    if not all_ranks_ok(dist, torch, dev, ok):
        job.fail(3, f"{wl['name']} strong-scaling pass: GPU results differ from the oracle, leave [0, n) or do not cover it (this rank: {ok}, checksums: {sums_ok})")
    log(f"{tag} shard [{lo}, {hi}) of {sp['total']} {sp['unit']}" + (f" ({sp['passes']} passes over the index records)" if "passes" in sp else "")
        + f": {mine} units, {kernel_ms:.3f} ms per step on this rank (all ranks: {', '.join(f'{x:.3f}' for x in per_rank)}; max {max(per_rank):.3f}); all ranks == oracle on their samples"
        + ("" if sums_ok is None else "; codes of all shards = a permutation of [0, n) by sum and sum of squares"))
    del d_o
    if rank != 0:
        return None
    return {"scaling": "strong", "value": total_units * steps / elapsed, "unit": "k-mer lookups/s", "n_gpus": world, "total_units_per_step": total_units,
            "split": f"{sp['unit']} by " + ("shard_range" if mode == "random" else "shard_records") + " (lphash_amd/dist.py), contiguous, no overlap",
            "shards": sp["shards"], "passes": sp.get("passes"), "ms_per_step": elapsed / steps * 1e3, "kernel_ms_per_rank": per_rank, "kernel_ms_max": max(per_rank),
            "kernel_ms_min": min(per_rank), "empty_step_ms": empty_step_ms, "planned_shard_ms": [strong_shard_ms(wl, sp, r_) for r_ in range(world)],
            "probes": ("members only (the fixed probe set of the strong pass is drawn from the index; the weak line above carries this workload's non-members)"
                       if mode == "random" and wl["nonmember_frac"] > 0 else None),
            "parity_vs_oracle": True if o is not None else None, "all_shards_cover_0_n": sums_ok,
            "note": "fixed job: the same total work at every N (the weak line above fixes the work PER GPU); at N = 1 this is one GPU doing all of it"}


def run_workload(job, wl):
    """One workload on every rank: index (rank 0 builds / loads, RCCL broadcast of the image), per-rank batch resident in HBM, warmup, timed
    steps between barriers, parity of the timed output against the oracle on every rank, CPU baseline on rank 0.  Returns the result
    dictionary on rank 0 (None elsewhere)."""
    args, rank, local_rank, world, dev, dist, torch, np, L = job.args, job.rank, job.local_rank, job.world, job.dev, job.dist, job.torch, job.np, job.L
    k, m, mode = wl["k"], wl["m"], wl["mode"]
    W = k - m + 1
    kmer_bits = wl["kmer_bits"]
    kind = {"p": L.PARTITIONED, "u": L.UNPARTITIONED, "pthash": L.PTHASH, "bbhash": L.BBHASH}[wl["kind"]]
    cls = {L.PARTITIONED: L.Mphf, L.UNPARTITIONED: L.MphfAlt, L.PTHASH: L.PtHash, L.BBHASH: L.BbHash}[kind]
    competitor = kind in (L.PTHASH, L.BBHASH)
    load = (lambda p_, dev_: cls.load(p_, k, kmer_bits=kmer_bits, device=dev_)) if competitor else (lambda p_, dev_: cls.load(p_, kmer_bits=kmer_bits, device=dev_))
    rec_len, n_records, nk = wl["rec_len"], wl["n_records"], wl["kmers"]
    steps, warmup = wl["steps"], wl["warmup"]
    tag = f"[bench r{rank} {wl['name']}]"

    # ---- synthetic index set: generated by rank 0 (deterministic, ~1 s per 100 M bases); further ranks read rank 0's copy from the cache directory ----
    recs = job.records(wl)

    # ---- index: built once (rank 0), cached on local disk, image broadcast over RCCL ----
    path = index_path(args.cache_dir, wl)
    build_s = None
    h = None
    index_sha = digest_ok = None
    if rank == 0:
        if not os.path.exists(path):
            t0 = time.time()
            tmp = path + f".tmp{os.getpid()}"
            bdev = local_rank if wl["gpu_build"] else None
            if kind == L.BBHASH:
                L.build_bbhash(recs, k, tmp, kmer_bits=kmer_bits, gamma=wl["bbhash_gamma"], threads=args.build_threads or 0)
            elif kind == L.PTHASH:
                L.build_pthash(recs, k, tmp, kmer_bits=kmer_bits, threads=args.build_threads or 0, device=bdev)
            else:
                L.build(recs, k, m, tmp, kind=kind, kmer_bits=kmer_bits, threads=args.build_threads or 0, device=bdev)
            os.replace(tmp, path)
            build_s = time.time() - t0
            log(f"{tag} built index over {nk} k-mers in {build_s:.1f}s -> {path} ({os.path.getsize(path)} bytes)")
        index_sha = sha256_file(path)
        want_sha = pinned_index_digest(wl)
        digest_ok = None if want_sha is None else (want_sha["sha256"] == index_sha and want_sha["bytes"] == os.path.getsize(path))
        log(f"{tag} index file {path}: {os.path.getsize(path)} bytes, sha256 {index_sha}"
            + ("" if want_sha is None else f" -- the oracle builder's file (tests/golden/full_size_digests.json): {'SAME' if digest_ok else 'DIFFERENT: ' + want_sha['sha256']}"))
        t0 = time.time()
        h = load(path, local_rank)
        log(f"{tag} loaded + attached in {time.time() - t0:.1f}s: {h.get_kmer_count()} k-mers, {h.get_minimizer_L0()} minimizers, "
            f"{h.info.fallback_kmers} fallback k-mers, image {h.info.image_bytes / 1e6:.1f} MB")
        assert h.get_kmer_count() == nk, "synthetic set has duplicate k-mers?"
    bcast_s = None
    image_t = None
    if dist is not None:
        from lphash_amd.dist import broadcast_image

        dist.barrier()  # the index file exists and rank 0 is attached
        if rank == 0 and not getattr(job, "peer_logged", False):
            # so that the first record of a real N-GPU run explains itself: which pairs of the node's GPUs can reach each other directly, and that the
            # collective spans all ranks
            job.peer_logged = True
            nd = torch.cuda.device_count()
            pairs = {f"{i}->{j}": bool(torch.cuda.can_device_access_peer(i, j)) for i in range(nd) for j in range(nd) if i != j}
            job.peer_access = {"devices_visible": nd, "pairs_with_peer_access": sum(pairs.values()), "pairs": len(pairs), "all_pairs": all(pairs.values()) if pairs else None,
                               "without_peer_access": [p_ for p_, ok_ in pairs.items() if not ok_][:16]}
            log(f"{tag} peer access (hipDeviceCanAccessPeer) over {nd} visible devices: {job.peer_access}; collective ranks {dist.get_world_size()} == world {world}: "
                f"{dist.get_world_size() == world}; backend {dist.get_backend()}")
        torch.cuda.synchronize()
        t0 = time.time()
        # RCCL over xGMI: the only collective of the data path.  No fallback: a failure ends this rank (and, through the launcher, the job)
        image_t = broadcast_image(h.image() if rank == 0 else None, dist, dev, src=0)
        torch.cuda.synchronize()
        bcast_s = time.time() - t0
        if rank != 0:
            h = cls.from_device_image(image_t.data_ptr(), image_t.numel(), local_rank)
        log(f"{tag} image broadcast {image_t.numel() / 1e6:.1f} MB in {bcast_s * 1e3:.1f} ms")

    stream = torch.cuda.current_stream().cuda_stream
    d_seq = recs.to_device(torch, dev)
    total_bases = int(d_seq.numel())
    rec_order = None
    if wl["shuffle_records"] and mode == "stream":
        rec_order = np.random.default_rng(3).permutation(n_records)
        d_seq = d_seq.view(n_records, rec_len)[torch.from_numpy(rec_order).to(dev)].contiguous().view(-1)

    # ---- per-rank batch, resident in HBM before the timed region ----
    d_kmers = None
    if mode == "random":
        Q = 1 << wl["batch_log2"]
        g = torch.Generator(device=dev)
        g.manual_seed(2 + rank)
        rec_i = torch.randint(0, n_records, (Q,), generator=g, device=dev, dtype=torch.int64)
        off_i = torch.randint(0, RECORD_KMERS, (Q,), generator=g, device=dev, dtype=torch.int64)
        d_pos = rec_i * rec_len + off_i
        del rec_i, off_i
        d_kmers = torch.empty(Q * (kmer_bits // 64), dtype=torch.int64, device=dev)
        h.pack_kmers(d_seq.data_ptr(), d_pos.data_ptr(), Q, d_kmers.data_ptr(), stream)
        torch.cuda.synchronize()
        del d_pos
        n_neg = int(Q * min(max(wl["nonmember_frac"], 0.0), 1.0))
        if n_neg:  # uniformly random k-mers (members with probability ~nk / 4^k ~ 0), scattered over the batch
            where = torch.randperm(Q, generator=g, device=dev)[:n_neg]
            lo_bits = min(2 * k, 64)
            lo = torch.randint(-(1 << 63), (1 << 63) - 1, (n_neg,), generator=g, device=dev, dtype=torch.int64)
            if lo_bits < 64:
                lo &= (1 << lo_bits) - 1
            if kmer_bits == 64:
                d_kmers[where] = lo
            else:
                hi = torch.randint(0, 1 << max(2 * k - 64, 1), (n_neg,), generator=g, device=dev, dtype=torch.int64) if 2 * k > 64 else torch.zeros_like(lo)
                d_kmers.view(Q, 2)[where, 0] = lo
                d_kmers.view(Q, 2)[where, 1] = hi
            del where, lo
        d_out = torch.empty(Q, dtype=torch.int64, device=dev)
        units = Q

        def step():
            h.query_packed(d_kmers.data_ptr(), Q, d_out.data_ptr(), stream)
    else:
        seq_off = torch.arange(0, n_records + 1, dtype=torch.int64, device=dev) * rec_len
        out_off = torch.arange(0, n_records + 1, dtype=torch.int64, device=dev) * (rec_len - k + 1)
        units = n_records * (rec_len - k + 1)
        d_out = torch.empty(units, dtype=torch.int64, device=dev)
        sink = wl["sink"]
        if sink == "runs":
            # arenas for ~2 / (W + 1) runs per window + the cuts at tile and wave ends, 15 % of slack (include/lphq.h); explicit codes: an eighth of that
            per_arena = int(units * (2.0 / (W + 1) + 1.0 / 400.0) * (1.15 if wl["index_set"] == "uniform" else 1.6) / RUN_ARENAS) + 4096
            xper_arena = (per_arena // 8 if wl["index_set"] == "uniform" else units // RUN_ARENAS) + 4096  # (repeat-rich sets: up to every window under a colliding minimizer)
            d_runs = torch.empty(RUN_ARENAS * per_arena * 2, dtype=torch.int64, device=dev)
            d_xc = torch.empty(RUN_ARENAS * xper_arena, dtype=torch.int64, device=dev)
            d_counts = torch.empty(RUN_ARENAS * 16, dtype=torch.int64, device=dev)

            def step():
                h.query_records_runs(d_seq.data_ptr(), total_bases, seq_off.data_ptr(), n_records, out_off.data_ptr(), d_runs.data_ptr(), RUN_ARENAS * per_arena,
                                     d_xc.data_ptr(), RUN_ARENAS * xper_arena, d_counts.data_ptr(), stream)
        elif sink.startswith("gather"):
            elem = int(sink[6:])
            tdt = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[elem]
            # the per-k-mer data a caller indexes by the codes: a fixed function of the slot (checkable anywhere without a second copy)
            d_table = ((torch.arange(nk, dtype=torch.int64, device=dev) * 2654435761) >> 9).bitwise_and((1 << (8 * elem - 1)) - 1 if elem < 8 else (1 << 62) - 1).to(tdt)
            d_pay = torch.empty(units, dtype=tdt, device=dev)

            def step():
                h.query_records_gather(d_seq.data_ptr(), total_bases, seq_off.data_ptr(), n_records, out_off.data_ptr(), d_table.data_ptr(), elem, d_pay.data_ptr(), 0, stream)
        else:

            def step():
                h.query_records(d_seq.data_ptr(), total_bases, seq_off.data_ptr(), n_records, d_out.data_ptr(), out_off.data_ptr(), True, stream)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    job.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    host_ms, gc_seen = [], []
    gc_cb = lambda phase, info: gc_seen.append((phase, info.get("generation"), time.perf_counter()))
    gc.callbacks.append(gc_cb)
    t0 = time.perf_counter()
    for a, b in evs:
        h0 = time.perf_counter()
        a.record()  # HIP events on the stream the kernel is launched on (torch's current stream)
        step()
        b.record()
        host_ms.append((time.perf_counter() - h0) * 1e3)  # what the host spent queueing this step: a stall between two launches of one step shows here AND in the events
    torch.cuda.synchronize()
    job.barrier()
    elapsed = time.perf_counter() - t0
    gc.callbacks.remove(gc_cb)
    gc_ms = sum((e[2] - s_[2]) * 1e3 for s_, e in zip(gc_seen[0::2], gc_seen[1::2]))  # (start, stop) pairs of Python's collector inside the timed region
    step_ms = [float(a.elapsed_time(b)) for a, b in evs]
    kernel_ms = float(np.mean(step_ms))
    kernel_ms_per_rank = [kernel_ms]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([kernel_ms], dtype=torch.float64, device=gather_device(dist, torch, dev))
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)  # a straggler shows here, not only in the max
        kernel_ms_per_rank = [float(e.item()) for e in every]

    # ---- the other sinks: what the timed launches left behind, turned into codes for the parity legs below (outside the timed region) ----
    sink_ok, sink_info = True, None
    if mode == "stream" and wl["sink"] == "runs":
        cnt = d_counts.cpu().numpy().reshape(RUN_ARENAS, 16)
        n_runs, n_xc = int(cnt[:, 0].sum()), int(cnt[:, 1].sum())
        d_st = torch.zeros(1, dtype=torch.int64, device=dev)
        h.expand_runs(d_runs.data_ptr(), RUN_ARENAS * per_arena, d_xc.data_ptr(), RUN_ARENAS * xper_arena, d_counts.data_ptr(), d_out.data_ptr(), units, d_st.data_ptr(), stream)
        torch.cuda.synchronize()
        sink_ok = int(d_st.item()) == 0 and int(cnt[:, 0].max()) <= per_arena and int(cnt[:, 1].max()) <= xper_arena
        sink_info = {"runs_per_step": n_runs, "windows_per_run": units / max(n_runs, 1), "explicit_codes_per_step": n_xc, "arena_fill_max": float(cnt[:, 0].max()) / per_arena,
                     "run_bytes_per_window": 16.0 * n_runs / units + 8.0 * n_xc / units,
                     "checked": "lphq_expand_runs of the timed launch's runs == the oracle on the sample" + ("; all windows together cover [0, n)" if not wl["shuffle_records"] else "")}
        del d_st
    elif mode == "stream" and wl["sink"].startswith("gather"):
        h.query_records(d_seq.data_ptr(), total_bases, seq_off.data_ptr(), n_records, d_out.data_ptr(), out_off.data_ptr(), True, stream)
        torch.cuda.synchronize()
        sink_ok = True
        for a_ in range(0, units, 1 << 28):  # payload of the timed launch == table[codes of the plain streaming pass], every window
            sink_ok = sink_ok and bool(torch.equal(d_pay[a_:a_ + (1 << 28)], d_table[d_out[a_:a_ + (1 << 28)]]))
        sink_info = {"elem_bytes": elem, "checked": "payload of the timed launch == table[codes] for every window; codes == the oracle on the sample"}

    # ---- parity of what was just measured, on EVERY rank (oracle = checker), + the CPU baseline on rank 0 ----
    cpu = cpu_all = cpu_b1 = cli_b1 = binding_b1 = None
    parity = None
    o = None
    if not args.no_cpu_baseline:
        if kind == L.BBHASH:
            job.fail(2, "--kind bbhash has no CPU oracle leg (its checker is the reference's own BooPHF, tests/golden/bbhash); use --no-cpu-baseline")
        o = load_oracle(job, wl, path)
        S = min(1 << (wl["cpu_sample_log2"] if rank == 0 else min(20, wl["cpu_sample_log2"])), units)
        if mode == "random":
            sample = d_kmers[: S * (kmer_bits // 64)].cpu().numpy().view(np.uint64)
            t0 = time.perf_counter()
            exp = o.query_packed(sample)
            cpu_s = time.perf_counter() - t0
            got = d_out[:S].cpu().numpy().view(np.uint64)
            what = f"first 2^{int(np.log2(S))} probes of the timed batch, oracle random branch, single thread"
        else:
            nrec_s = max(1, min(n_records, S // (rec_len - k + 1)))
            t0 = time.perf_counter()
            exp = np.concatenate([o(recs[i if rec_order is None else int(rec_order[i])], True) for i in range(nrec_s)])
            cpu_s = time.perf_counter() - t0
            S = len(exp)
            got = d_out[:S].cpu().numpy().view(np.uint64)
            what = f"first {nrec_s} records of the batch, oracle streaming branch, single thread"
        parity = bool(np.array_equal(got, exp)) and sink_ok
        if mode == "stream" and wl["sink"] != "codes" and not bool(int(d_out.min()) == 0 and int(d_out.max()) == nk - 1):  # the index records: every code is a slot
            parity = False
        if mode == "random" and wl["nonmember_frac"] == 0.0:
            parity = parity and bool(int(d_out.max()) < nk) and bool(int(d_out.min()) >= 0)  # positive probes: every code of the batch is a valid slot
        log(f"{tag} cpu oracle: {S / cpu_s / 1e6:.2f} M lookups/s on {S} units; GPU == oracle on the sample of this rank's timed batch: {parity}")
        if rank == 0:
            cpu = {"value": S / cpu_s, "unit": "k-mer lookups/s", "cores": 1, "kind": "port", "sample": what}
        if rank == 0 and wl["headline"] and mode == "random" and not competitor and parity:
            # the same oracle on all host cores (POSIX threads over contiguous slices, inside the oracle library)
            quota = cpu_quota_cores()  # threads beyond the container's CPU-time quota only buy throttling (tools/host_probe.py)
            nthreads = len(os.sched_getaffinity(0)) if quota is None else max(1, min(len(os.sched_getaffinity(0)), 2 * int(np.ceil(quota))))  # 2x: fastest in the probe
            wpk = kmer_bits // 64
            S3 = min(units, S * 8)
            big = d_kmers[: S3 * wpk].cpu().numpy().view(np.uint64)
            res = np.zeros(S3, dtype=np.uint64)  # pages exist before the timer starts
            o.query_packed_threads(big[: wpk * min(S3, 1 << 16)], nthreads, res)
            t0 = time.perf_counter()
            parts = [o.query_packed_threads(big, nthreads, res)]
            cpu3_s = time.perf_counter() - t0
            parity = parity and bool(np.array_equal(np.concatenate(parts), d_out[:S3].cpu().numpy().view(np.uint64)))
            cpu_all = {"value": S3 / cpu3_s, "unit": "k-mer lookups/s", "cores": nthreads, "kind": "port", "cpus_visible": len(os.sched_getaffinity(0)),
                       "cpu_quota_cores": quota,
                       "sample": f"first {S3} probes of the timed batch, {nthreads} POSIX threads over contiguous slices (= the CPUs visible, capped at twice the "
                                 f"container's CPU-time quota: the fastest setting on the box, tools/host_probe.py)"}
    if digest_ok is False:  # the index the GPU just answered from is not the file the oracle builder writes for this set: no parity claim stands on it
        parity = False
    if not all_ranks_ok(dist, torch, dev, parity is not False):
        job.fail(3, f"{wl['name']}: GPU results differ from the oracle on at least one rank (this rank: {parity}; index digest pinned: {digest_ok})")

    strong = None
    if (dist is not None or args.strong) and wl["sink"] == "codes":  # (the sinks share the codes workload's strong line: same lookups)
        strong = strong_pass(job, wl, h, recs, d_seq, o, stream, rec_order)

    if rank == 0 and o is not None and wl["headline"] and args.b1_records > 0 and not competitor:
        # BASELINE.md B1, "the number the reference tool would print": the oracle's restatement of src/query.cpp:24-88 over a
        # .fa.gz of the first records of the index set -- one thread, inflate + record parsing inside the timers, one result
        # vector per record, streaming pass then random pass.  Beside it the product's own `lphash query-*` on the same file
        # (same CSV, PCIe and parsing included): the drop-in comparison, never the reported `value`.
        import gzip
        import subprocess

        nb1 = min(n_records, args.b1_records)
        fa = os.path.join(args.cache_dir, f"b1_seed1_k{k}_r{nb1}.fa.gz")
        if not os.path.exists(fa):
            with gzip.open(fa + ".tmp", "wb", compresslevel=6) as f:
                for i in range(nb1):
                    f.write(b">s%d\n" % i)
                    f.write(recs[i])
                    f.write(b"\n")
            os.replace(fa + ".tmp", fa)
        n_s, n_r, us_s, us_r = o.query_file(fa)
        assert n_s == n_r == nb1 * (rec_len - k + 1)
        cpu_b1 = {"value": n_r / (us_r * 1e-6), "unit": "k-mer lookups/s", "cores": 1, "kind": "port",
                  "streaming_ns_per_kmer": us_s * 1e3 / n_s, "random_ns_per_kmer": us_r * 1e3 / n_r,
                  "sample": f"oracle restatement of the reference driver (src/query.cpp:24-88) on a .fa.gz of the first {nb1} index records "
                            f"({n_s} k-mers per pass); gz inflate + parse inside the timers; value = random pass"}
        exe = os.path.join(ROOT, "lphash_amd", "lphash")

        def csv_runs(cmd, reps):
            """(streaming, random) ns per k-mer of `reps` runs of a query driver that prints the reference's 5-field CSV (a whole run takes
            ~0.6 s, the timed passes 7-40 ms: one late interrupt inside the HIP runtime is +8 ms on this pool, profiles/r04_cli_stalls.txt -- hence
            the median of three, with every run listed)"""
            runs = []
            for _ in range(reps):
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                f5 = r.stdout.strip().split(",")
                if r.returncode == 0 and len(f5) == 5 and int(f5[2]) == n_s:
                    runs.append((float(f5[3]), float(f5[4])))
                else:
                    log(f"{tag} {os.path.basename(cmd[0])} failed ({r.returncode}): {r.stderr[-500:]}")
                    return None
            return runs

        if os.path.exists(exe):
            runs = csv_runs([exe, "query-p" if kind == L.PARTITIONED else "query-u", "-i", path, "-q", fa, "--kmer-bits", str(kmer_bits)], 3)
            if runs:
                cli_b1 = {"streaming_ns_per_kmer": float(np.median([a for a, _ in runs])), "random_ns_per_kmer": float(np.median([b for _, b in runs])),
                          "kmers": n_s, "runs": runs,
                          "note": "lphash query-* (this repo's CLI, GPU path) on the same .fa.gz: gz decode + parse + PCIe inside the timers; median of three runs"}
        # the reference-side binding (examples/gpu_mphf.hpp under the reference's query<MPHF> driver, examples/query_main.cpp; INTEGRATION.md
        # section 2) on the same file: batches of records through lphq_session_* (EVERY code comes back to the host: 8 B per k-mer over PCIe, which
        # the CLI's default pass does not pay), and the reference's literal one-record-per-call loop over lphq_query_contig
        exb = os.path.join(ROOT, "examples", f"query_example{kmer_bits}")
        if os.path.exists(exb):
            binding_b1 = {"note": "examples/query_example (gpu_mphf<> binding, compiled with g++ against include/lphq.h) on the same .fa.gz; batched = "
                                  "submit / collect over lphq_session_*, all codes returned to the host (streaming batches cross PCIe as runs and are expanded by "
                                  "host threads); batched_runs = collect_runs, nothing expanded; per_record = the reference's literal loop"}
            # batched: codes expanded on the host for every k-mer (the reference's vector<uint64_t>); batched_runs: the streaming pass collected as runs
            # (gpu_mphf::collect_runs -- what the reference computes before it expands it; the random pass is the batched one's)
            for label, extra, reps in (("batched", [], 3), ("batched_runs", ["--runs"], 3), ("per_record", ["--per-record"], 1)):
                runs = csv_runs([exb, "query-p" if kind == L.PARTITIONED else "query-u", "-i", path, "-q", fa] + extra, reps)
                if runs:
                    binding_b1[label + "_streaming_ns_per_kmer"] = float(np.median([a for a, _ in runs]))
                    binding_b1[label + "_random_ns_per_kmer"] = float(np.median([b for _, b in runs]))
                    binding_b1[label + "_runs"] = runs
            # plain FASTA of the first 1000 index records (100 Mbases: seven batches, a pipeline that reaches its steady state): what the transport of the result
            # costs once the .gz decoder (0.13 ns per k-mer on the file above, and only two batches) is out of the way
            n_plain = min(n_records, 1000)
            fa_plain = os.path.join(args.cache_dir, f"b1_seed1_k{k}_r{n_plain}.fa")
            if not os.path.exists(fa_plain):
                with open(fa_plain + ".tmp", "wb") as fo:
                    for i in range(n_plain):
                        fo.write(b">s%d\n" % i)
                        fo.write(recs[i])
                        fo.write(b"\n")
                os.replace(fa_plain + ".tmp", fa_plain)
            n_keep, n_s = n_s, n_plain * (rec_len - k + 1)  # (csv_runs checks the k-mer count of the file it runs on)
            plain = {"records": n_plain, "kmers": n_s}
            for label, extra in (("batched", []), ("batched_runs", ["--runs"]), ("batched_runs_two_lanes", ["--runs", "--devices", f"{local_rank},{local_rank}"])):
                runs = csv_runs([exb, "query-p" if kind == L.PARTITIONED else "query-u", "-i", path, "-q", fa_plain] + extra, 3)
                if runs:
                    plain[label + "_streaming_ns_per_kmer"] = float(np.median([a for a, _ in runs]))
                    plain[label + "_runs"] = runs
            n_s = n_keep
            binding_b1["plain_fasta"] = plain
            log(f"{tag} binding: {binding_b1}")
        log(f"{tag} B1 reference-faithful driver: {cpu_b1['streaming_ns_per_kmer']:.1f} / {cpu_b1['random_ns_per_kmer']:.1f} ns per k-mer (streaming / random); CLI: {cli_b1}")

    # ---- N > 1: every replica (rank 0 parsed the file, the others received the image over RCCL) must answer alike ----
    replicas_agree = None
    if dist is not None:
        # a probe set common to all ranks, in both modes (random: the timed batches differ per rank; streaming: the weak line streams the same records
        # on every rank -- their checksum rides along -- but the strong line's shards differ): packed k-mers of seeded windows through the random kernel
        gcm = torch.Generator(device=dev)
        gcm.manual_seed(777)
        Qc = 1 << 20
        win = torch.randint(0, nk, (Qc,), generator=gcm, device=dev, dtype=torch.int64)
        posc = (win // RECORD_KMERS) * rec_len + win % RECORD_KMERS
        d_kc = torch.empty(Qc * (kmer_bits // 64), dtype=torch.int64, device=dev)
        h.pack_kmers(d_seq.data_ptr(), posc.data_ptr(), Qc, d_kc.data_ptr(), stream)
        d_chk = torch.empty(Qc, dtype=torch.int64, device=dev)
        h.query_packed(d_kc.data_ptr(), Qc, d_chk.data_ptr(), stream)
        torch.cuda.synchronize()
        def ordered_checksum(v):  # (sum, order-dependent weighted sum) in wrapping int64 arithmetic, a chunk at a time: C4's 2 G codes would need 32 GB of temporaries at once
            tot = torch.zeros(2, dtype=torch.int64, device=dev)
            step_ = 1 << 26
            for a_ in range(0, int(v.numel()), step_):
                c_ = v[a_:a_ + step_]
                w_ = torch.arange(a_, a_ + int(c_.numel()), dtype=torch.int64, device=dev) * 2 + 1
                tot += torch.stack([c_.sum(), (c_ * w_).sum()])
            return [tot[0], tot[1]]

        parts = ordered_checksum(d_chk) + [d_chk.max()]
        if mode == "stream":  # every rank streamed the same records in the weak line
            parts += ordered_checksum(d_out)
        t = torch.stack(parts)
        tmin, tmax = t.clone(), t.clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        replicas_agree = bool(torch.equal(tmin, tmax)) and (competitor or int(t[2].item()) < nk)
        common_ok = True
        if rank == 0 and o is not None:  # the common probes against the oracle (checker only)
            common_ok = bool(np.array_equal(o.query_packed(d_kc.cpu().numpy().view(np.uint64)), d_chk.cpu().numpy().view(np.uint64)))
            log(f"{tag} {Qc} probes common to all ranks: GPU == oracle: {common_ok}")
            parity = parity and common_ok
        log(f"{tag} replicas agree on the common probes: {replicas_agree}")
        if not all_ranks_ok(dist, torch, dev, replicas_agree and common_ok):
            job.fail(3, f"{wl['name']}: replicas disagree or differ from the oracle")

    out = None
    if rank == 0:
        if kind == L.BBHASH:  # per probed level a bit-array word (8 B), on the hit a rank sample (8 B) + up to 7 more words; ~1/(1 - 0.63) levels at gamma 1
            chain, mix = 2.7 * 8 + 8 + 3.5 * 8, None
            bpl = (8 if k <= 31 else 16) + 8 + chain
        elif kind == L.PTHASH:  # PTHash ranks word 8 + free-slot EF 26 * P(free) + k-mer in + code out
            p_free = 1.0 - h.info.distinct_minimizers / h.info.minimizer_table_size
            chain, mix = 8 + 26 * p_free, None
            bpl = (8 if k <= 31 else 16) + 8 + chain
        else:
            bpl, chain, mix = algorithmic_bytes_per_lookup(h, mode, k, W, kind == L.PARTITIONED)
        sk = (W + 1) / 2.0  # windows per super-k-mer on random sequence
        if mode == "stream" and wl["sink"] == "runs":
            # 1 B of base in, one lookup chain and ONE 16-byte run out per super-k-mer (k = 63 / m = 20: 1 + 74.7 / 22.5 + 16 / 22.5 = 5.0 B per k-mer)
            bpl = 1 + (chain + 16) / sk
        elif mode == "stream" and wl["sink"].startswith("gather"):
            # 1 B of base in, the chain per super-k-mer, the k-mer's element read once (consecutive codes: contiguous) and written once
            bpl = 1 + chain / sk + 2 * elem
        cfg_name = config_name(wl)
        if job.hbm is None and not args.no_hbm_probe:
            job.hbm = hbm_ceiling_gbs(torch, dev)
        peak_read, peak_copy = job.hbm if job.hbm is not None else (None, None)
        value = world * units * steps / elapsed
        out = {
            "metric": f"k-mer lookups/sec (whole node), k={k} {'BBHash' if kind == L.BBHASH else 'plain PTHash' if kind == L.PTHASH else ('partitioned' if kind == L.PARTITIONED else 'unpartitioned')} MPHF" + ("" if mode == "random" else ", streaming"),
            "value": value,
            "unit": "k-mer lookups/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": cfg_name + (f": boomphf::mphf over the k-mers, gamma {wl['bbhash_gamma']}," if kind == L.BBHASH else ": pthash::single_phf over the k-mers," if kind == L.PTHASH else (": partitioned LP-MPHF" if kind == L.PARTITIONED else ": unpartitioned LP-MPHF"))
                + f" k={k} m={m}, {nk} k-mers (SplitMix64 seed 1, {n_records} records x {rec_len} bases"
                + ("" if wl["index_set"] == "uniform" else f"; the second half of the records = diverged copies of the first half: {h.info.fallback_kmers / nk:.1%} of the k-mers in the fallback function") + "); "
                + ((f"2^{wl['batch_log2']} random positive packed k-mer lookups per GPU per step" if wl["nonmember_frac"] == 0.0 else
                    f"2^{wl['batch_log2']} random packed k-mer lookups per GPU per step, {wl['nonmember_frac']:.0%} of them uniformly random non-members")
                   if mode == "random"
                   else "all windows of the index records per step, streaming kernel" + (", records in shuffled order" if rec_order is not None else "")
                   + ("" if wl["sink"] == "codes" else ", result as runs of affine codes (lphq_query_records_runs: one 16-byte entry per super-k-mer)" if wl["sink"] == "runs"
                      else f", fused with the payload gather (lphq_query_records_gather: table[code], {wl['sink'][6:]}-byte elements, {nk} of them)")),
                "k": k, "m": m, "kmers": nk, "units_per_step_per_gpu": units, "kmer_bits": kmer_bits,
                "index_image_bytes": int(h.info.image_bytes), "type_mix": mix,
            },
            "kernel_ms": kernel_ms,
            "kernel_ms_steps": [round(x, 4) for x in step_ms],  # every timed step (HIP events): a host stall between two launches of one step shows here, not in a mean
            "kernel_ms_median": float(np.median(step_ms)),
            "host_queue_ms_steps": [round(x, 4) for x in host_ms], "python_gc_in_timed_region": {"collections": len(gc_seen) // 2, "ms": round(gc_ms, 3)},
            "kernel_ms_per_rank": kernel_ms_per_rank,
            "kernel_ms_min": min(kernel_ms_per_rank),
            "kernel_ms_max": max(kernel_ms_per_rank),
            "parity_vs_oracle": parity,
            "parity_ranks_checked": world if parity is not None else 0,
            "rccl_ranks": int(dist.get_world_size()) if dist is not None else 1,
            "collective_backend": (dist.get_backend() if dist is not None else None),
            "replicas_agree": replicas_agree,
        }
        if getattr(job, "peer_access", None) is not None:
            out["peer_access"] = job.peer_access
        traffic = valu = traffic_stale = traffic_src = l2_miss = rocprof_us = None
        try:  # HBM bytes and VALU instructions per unit from the committed PMC passes of this exact workload (never measured inside this run)
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = (f"{mode}|{wl['kind']}|k{k}|m{m}|n{nk}|b{wl['batch_log2'] if mode == 'random' else 0}" + ("" if wl["sink"] == "codes" else "|" + wl["sink"])
                   + ("" if wl["nonmember_frac"] == 0.0 else f"|nonmember{wl['nonmember_frac']}") + ("" if rec_order is None else "|shuffled"))
            variant = wl["index_set"] != "uniform"
            if key in tj and not variant:  # (counter passes exist for the default kernels on the primary workloads only)
                # the passes are evidence for the kernels they were taken on: an entry whose device sources are not the ones in this tree is reported
                # as stale and contributes no number
                if tj[key].get("tree") == device_source_sha():
                    traffic = tj[key]["bytes_per_unit"] * units
                    traffic_src = tj[key].get("source")
                    valu = tj[key].get("valu_lane_instr_per_unit"), tj[key].get("valu_source")
                    l2_miss = tj[key].get("l2_misses_per_unit")
                    if tj[key].get("kernel_us_rocprof"):
                        rocprof_us = tj[key]["kernel_us_rocprof"], tj[key].get("kernel_stats_source")
                else:
                    traffic_stale = {"counters_taken_on_tree": tj[key].get("tree"), "this_tree": device_source_sha(), "source": tj[key].get("source"),
                                     "note": "the committed counter passes predate a change of the device sources: traffic / instr_per_unit withheld"}
        except (OSError, ValueError, KeyError):
            pass
        kernel_s = max(kernel_ms_per_rank) * 1e-3 if world > 1 else kernel_ms * 1e-3
        achieved = bpl * units / kernel_s / 1e9
        # The strict lower bound beside the algorithmic figure: only what MUST cross HBM once -- the input (8 / 16 B per packed k-mer, 1 B
        # per base when streaming) and the 8-byte code -- as if the index cost nothing.  An image below the 256-MiB Infinity Cache does stay on chip
        # between steps, so for it the index part of `achieved` is fabric traffic, not HBM traffic, and frac_compulsory is the honest HBM fraction.
        compulsory = ((8 if k <= 31 else 16) + 8) if mode == "random" else (1.0 * (units + n_records * (k - 1)) / units + 8)
        if mode == "stream" and wl["sink"] == "runs":
            compulsory += 16 / sk - 8
        elif mode == "stream" and wl["sink"].startswith("gather"):
            compulsory += 2 * elem - 8
        image_bytes = int(h.info.image_bytes)
        if mode == "random":
            kernel_name = "k_query_packed"
        else:
            kernel_name = "k_query_tiles_stream8" if W >= 8 else "k_query_tiles_stream"  # (kStream8MinW, lphash_amd/csrc/stream8_device.cuh)
            if wl["sink"] != "codes":
                kernel_name = "k_query_tiles_stream8<.., kSinkRuns>" if wl["sink"] == "runs" else "k_query_tiles_stream8<.., kSinkGather>"
        # What binds the kernel, from the evidence in this line -- the HBM `frac` stays the figure north_star asks for:
        #   valu_issue            the committed SQ_INSTS_VALU pass x this run's rate is >= 0.88 of the VALU issue peak at 2.4 GHz (boxes of this pool clock
        #                         the VALU-bound kernels +-4 %: C4 random measures 0.89-0.97, C4 streaming 0.80-0.85)
        #   fabric_line_requests  the image is Infinity-Cache resident: the index lines never reach HBM; the L2 misses of the committed TCC pass x this run's
        #                         rate against the line-request rate the fabric sustains for scattered gathers (FABRIC_LINE_PEAK)
        #   hbm                   otherwise (the image exceeds the Infinity Cache, or no counters for this tree / variant)
        residency = "infinity_cache" if image_bytes < (256 << 20) else "hbm"
        ceil_key = f"{wl['kind']}|k{k}|m{m}|n{nk}" + ("" if wl["index_set"] == "uniform" else "|" + wl["index_set"])
        ceiling = None if competitor else ceiling_for(ceil_key)
        FABRIC_LINE_PEAK = ceiling["lines_per_s"] if ceiling else FABRIC_LINE_PEAK_FALLBACK
        peak_note = ("measured by tools/gather_ceiling.hip on a table of this image's line side" if ceiling else
                     "INFERRED: no probe result for this index (profiles/gather_ceiling.json) -- the fallback is the k=31 random kernel's own best observed rate")
        valu_frac = (valu[0] * units / kernel_s / VALU_PEAK_LANE_INSTR) if (valu is not None and valu[0]) else None
        line_rate = (l2_miss * units / kernel_s) if l2_miss else None
        if valu_frac is not None and valu_frac >= 0.88:
            bound, why = "valu_issue", f"roofline_valu.frac {valu_frac:.2f} >= 0.88" + ("" if not line_rate else f"; {line_rate / 1e9:.1f} G line requests/s of {FABRIC_LINE_PEAK / 1e9:.1f}")
        elif residency == "infinity_cache":
            bound = "fabric_line_requests"
            why = (f"index on chip; {l2_miss} L2 misses per unit x rate = {line_rate / 1e9:.1f} G lines/s of {FABRIC_LINE_PEAK / 1e9:.1f} G lines/s ({'probe' if ceiling else 'inferred ceiling'})" if line_rate else
                   "index on chip (no counter pass for this tree / variant: the line rate is not stated)")
        else:
            bound, why = "hbm", ("index beyond the Infinity Cache: its lines come from HBM" + ("" if valu_frac is None else f"; roofline_valu.frac {valu_frac:.2f}")
                                 + ("" if not (traffic and peak_read) else f"; counted traffic {traffic / kernel_s / 1e12:.2f} TB/s against {peak_read / 1e3:.2f} TB/s measured read"))
        wall_s = elapsed / steps  # the time base the driver's clock bounds (barrier to barrier, launch overheads included)
        out["roofline"] = {"bound": bound, "bound_evidence": why, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "time_base": "HIP events around each launch on the launch stream, mean over the timed steps (max over ranks at N > 1)",
                           "frac_from_ms_per_step": bpl * units / wall_s / 1e9 / HBM_PEAK_GBS,
                           "kernel_us_rocprof": rocprof_us[0] if rocprof_us else None, "kernel_stats_source": rocprof_us[1] if rocprof_us else None,
                           "frac_from_rocprof": (bpl * units / (rocprof_us[0] * 1e-6) / 1e9 / HBM_PEAK_GBS) if rocprof_us else None,
                           "peak_measured": peak_read, "peak_measured_copy": peak_copy,
                           "frac_of_measured": (achieved / peak_read) if peak_read else None,
                           "traffic": traffic, "traffic_source": traffic_src, "bytes_per_lookup": bpl, "kernel": kernel_name,
                           "compulsory_bytes_per_lookup": compulsory, "achieved_compulsory": compulsory * units / kernel_s / 1e9,
                           "frac_compulsory": compulsory * units / kernel_s / 1e9 / HBM_PEAK_GBS,
                           "index_residency": residency,
                           "index_residency_note": (f"the {image_bytes / 1e6:.1f}-MB image is smaller than the 256-MiB Infinity Cache: its lines are re-read from on-chip "
                                                    "memory, `traffic` counts fabric requests, and the HBM proper carries the compulsory stream only"
                                                    if image_bytes < (256 << 20) else
                                                    f"the {image_bytes / 1e6:.1f}-MB image exceeds the 256-MiB Infinity Cache: index lines come from HBM")}
        if line_rate:
            out["roofline"].update({"l2_misses_per_unit": l2_miss, "line_requests_per_s": line_rate, "line_requests_peak": FABRIC_LINE_PEAK,
                                    "line_requests_peak_source": peak_note, "frac_line_requests": line_rate / FABRIC_LINE_PEAK})
        if ceiling and mode == "random":
            # the kernel against the bare dependent chain of its lookups (pilot word -> block line, NO hashing, no other work) measured by the probe on tables
            # of this image's section sizes: how much of what the memory system allows for this access shape the whole lookup reaches
            out["roofline"].update({"gather_chain_items_per_s": ceiling["chain_items_per_s"], "frac_of_gather_chain": units / kernel_s / ceiling["chain_items_per_s"],
                                    "gather_chain_source": ceiling["source"], "gather_chain_tables": {"small_bytes": ceiling["small_bytes"], "big_bytes": ceiling["big_bytes"]}})
        if sink_info is not None:
            out["sink"] = dict(sink_info, kind=wl["sink"])
        if traffic_stale is not None:
            out["roofline"]["traffic_stale"] = traffic_stale
        if valu is not None and valu[0]:
            # the second roofline of the same kernel: VALU issue.  instr_per_unit = SQ_INSTS_VALU of the committed counter pass x 64 lanes / units of
            # one launch (a wave64 instruction occupies a SIMD's 16 lanes for 4 cycles whatever the number of live lanes)
            rate = valu[0] * units / kernel_s
            out["roofline_valu"] = {"bound": "valu_issue", "instr_per_unit": valu[0], "achieved_lane_instr_per_s": rate, "peak": VALU_PEAK_LANE_INSTR,
                                    "frac": rate / VALU_PEAK_LANE_INSTR, "unit": "lane-instructions/s", "source": valu[1]}
            if mode == "random" and not competitor:
                # distance to the floor: a lookup hashes W m-mers; MurmurHash2_64 of one word is 24 VALU instructions on gfx950 and cannot be
                # fewer, the scan spends 29 per m-mer (24 + 2 to cut the m-mer out + 3 for min / argmin).  The rest is a LISTING of the compiled kernel
                # (profiles/isa_rest.json from tools/isa_rest_listing.py; readable: profiles/r05_isa_rest.txt), so the floor = 24 W + listed rest is a count of
                # instructions in the binary, with the counter pass (measured) beside it
                out["roofline_valu"].update({"hash_floor_instr_per_unit": 24 * W, "scan_instr_per_unit": 29 * W, "rest_instr_per_unit_measured": valu[0] - 29 * W})
                listed = None
                try:
                    ij = json.load(open(os.path.join(ROOT, "profiles", "isa_rest.json")))
                    tkey = f"{0 if kind == L.PARTITIONED else 1},{'true' if kmer_bits == 128 else 'false'},{6 if W > 15 else 4}"
                    if ij.get("tree") == device_source_sha() and tkey in ij and ij[tkey]["W"] == W:
                        listed = ij[tkey]
                except (OSError, ValueError, KeyError):
                    pass
                if listed is not None:
                    floor = listed["floor_instr_per_unit_listed"]
                    out["roofline_valu"].update({"rest_instr_per_unit": listed["rest_instr_per_unit_listed"], "rest_source": "profiles/isa_rest.json (ISA listing, profiles/r05_isa_rest.txt)",
                                                 "listed_instr_per_unit": listed["listed_instr_per_unit"], "floor_instr_per_unit": floor,
                                                 "floor_source": "24 W + listed rest", "above_floor": valu[0] / floor - 1.0})
                else:
                    out["roofline_valu"].update({"rest_instr_per_unit": valu[0] - 29 * W, "rest_source": "counted - 29 W (no listing for this tree / kernel)",
                                                 "floor_instr_per_unit": valu[0] - 5 * W, "floor_source": "counted - 5 W", "above_floor": valu[0] / (valu[0] - 5 * W) - 1.0})
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if cpu_all is not None:
            out["cpu_baseline_all_cores"] = cpu_all
        if cpu_b1 is not None:
            out["cpu_baseline_reference_driver"] = cpu_b1
        if cli_b1 is not None:
            out["cli_end_to_end"] = cli_b1
        if binding_b1 is not None:
            out["binding_end_to_end"] = binding_b1
        out["index_sha256"] = index_sha
        out["index_digest_pinned"] = digest_ok
        if strong is not None:
            out["strong"] = strong
        if build_s is not None:
            out["index_build_s"] = build_s
        if bcast_s is not None:
            out["image_broadcast_ms"] = bcast_s * 1e3
    h.close()
    del h, d_seq, d_out, d_kmers, image_t
    torch.cuda.empty_cache()
    return out


def probe_ceiling(args, torch, np, L, result_out):
    """bench.py --probe-ceiling: the independent ceilings of the random-lookup kernels on this box (see FABRIC_LINE_PEAK_FALLBACK above)."""
    import subprocess

    exe = os.path.join(ROOT, "tools", "gather_ceiling")
    if not os.path.exists(exe):
        log(f"[bench] {exe} is missing: python -c 'import __graft_entry__ as g; g.build()' builds it")
        return 2
    os.makedirs(args.cache_dir, exist_ok=True)
    from lphash_amd.synth import synthetic_records

    entries, sizes = {}, []
    for name, kind, k, kmers in (("C2", "p", 31, 100_000_000), ("C5", "u", 31, 100_000_000), ("C4", "p", 63, args.secondary_c4_kmers)):
        wl = make_workload(name, "random", kind, k, 20, kmers, 28, 1, 0, 20, True)
        path = index_path(args.cache_dir, wl)
        if not os.path.exists(path):
            flat, _ = synthetic_records(seed=1, n_records=wl["n_records"], length=wl["rec_len"], views=False)
            L.build(RecordSet(wl["n_records"], wl["rec_len"], flat=flat), k, 20, path + ".tmp", kind=L.PARTITIONED if kind == "p" else L.UNPARTITIONED, kmer_bits=wl["kmer_bits"],
                    threads=0, device=0)
            os.replace(path + ".tmp", path)
            del flat
        h = (L.Mphf if kind == "p" else L.MphfAlt).load(path, kmer_bits=wl["kmer_bits"])
        small, big = image_sections(h.info, k - 20 + 1, kind == "p")
        key = f"{kind}|k{k}|m20|n{wl['kmers']}"
        entries[key] = {"workload": name, "image_bytes": int(h.info.image_bytes), "small_bytes": small, "big_bytes": big}
        sizes += [str(small), str(big)]
        h.close()
    r = subprocess.run([exe] + sizes, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        log(f"[bench] gather_ceiling failed ({r.returncode}): {r.stderr[-500:]}")
        return 2
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    gathers = [ln for ln in lines if ln["probe"] == "gather"]
    for (key, e), g in zip(entries.items(), gathers):
        e.update({"small_bytes": g["small_bytes"], "big_bytes": g["big_bytes"], "lines_per_s": g["lines_per_s"], "chain_items_per_s": g["chain_items_per_s"], "items": g["items"]})
    out = {"probe_sha": probe_sha(), "device": torch.cuda.get_device_name(0), "entries": entries, "append": [ln for ln in lines if ln["probe"] == "append"],
           "note": "tools/gather_ceiling.hip run by bench.py --probe-ceiling: lines_per_s = scattered 128-B-line gathers from a table of big_bytes; chain_items_per_s = "
                   "the dependent chain small[4 B] -> big[128-B line] (a lookup's pilot word -> block line) with no hashing; append = one atomicAdd per workgroup"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gather_ceiling.json"), "w") as f:
        json.dump(out, f, indent=1)
    result_out.write(json.dumps(out) + "\n")
    result_out.flush()
    return 0


def summary_table(out):
    """<= 1,500 characters, one row per workload: what the driver's 2,000-character tail of stderr must still hold when the JSON line is 20 KB."""
    rows = [("headline " + out["config"]["workload"].split(":")[0], out)] + [(e["name"], e) for e in out.get("secondary", [])]
    lines = ["[bench summary] workload            G units/s  frac   bound                 parity pinned"]
    for name, e in rows:
        r = e.get("roofline", {})
        lines.append(f"[bench summary] {name[:19]:<19} {e['value'] / 1e9:9.2f}  {r.get('frac', 0):.3f}  {str(r.get('bound'))[:21]:<21} {str(e.get('parity_vs_oracle')):<6} {e.get('index_digest_pinned')}")
    for err in out.get("secondary_errors", []):
        lines.append(f"[bench summary] {err['name'][:19]:<19} FAILED: {err['error'][:60]}")
    return "\n".join(lines)[:1500]


def secondary_entry(out, wl):
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "kernel_ms", "kernel_ms_steps", "kernel_ms_median", "host_queue_ms_steps", "python_gc_in_timed_region", "kernel_ms_min", "kernel_ms_max", "parity_vs_oracle",
            "parity_ranks_checked", "replicas_agree", "roofline", "roofline_valu", "cpu_baseline", "index_build_s", "image_broadcast_ms", "scaling", "strong",
            "index_sha256", "index_digest_pinned", "sink", "binding_end_to_end")
    e = {"name": wl["name"], "workload": out["config"]["workload"], "units_per_step_per_gpu": out["config"]["units_per_step_per_gpu"],
         "index_image_bytes": out["config"]["index_image_bytes"]}
    e.update({k_: out[k_] for k_ in keep if k_ in out})
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["random", "stream"], default="random")
    ap.add_argument("--sink", default="codes", help="--mode stream: what the pass leaves behind -- codes (the BASELINE metric), runs (lphq_query_records_runs) or gather1 / "
                                                    "gather2 / gather4 / gather8 (lphq_query_records_gather, element bytes)")
    ap.add_argument("--kmers", type=int, default=100_000_000, help="k-mers in the index (C2: 100 M)")
    ap.add_argument("--batch-log2", type=int, default=28, help="random mode: 2^x lookups per step per GPU")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--m", type=int, default=20)
    ap.add_argument("--kind", choices=["p", "u", "pthash", "bbhash"], default="p",
                    help="p / u: LPHash partitioned / unpartitioned; pthash / bbhash: the competitors of SURVEY 8(f)-4 over the same k-mers (random mode)")
    ap.add_argument("--bbhash-gamma", type=float, default=1.0, help="--kind bbhash: load factor (the reference's ptbb_build default is 1)")
    ap.add_argument("--build-threads", type=int, default=0)
    ap.add_argument("--gpu-build", action="store_true",
                    help="run the PTHash pilot search of the index build on the GPU (same file bytes; default is the sequential CPU search so that "
                         "the measured query path does not depend on construction code)")
    ap.add_argument("--cpu-sample-log2", type=int, default=24, help="oracle-timed sample size (rank 0; the other ranks check 2^20 of their batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="no oracle leg at all (profiling runs): neither parity nor cpu_baseline")
    ap.add_argument("--nonmember-frac", type=float, default=0.0,
                    help="random mode: this fraction of the probes are uniformly random k-mers (non-members; SURVEY 8(d) secondary run)")
    ap.add_argument("--shuffle-records", action="store_true", help="stream mode: the index records in a random order (SURVEY 8(d) variant)")
    ap.add_argument("--index-set", choices=["uniform", "diverged"], default="uniform",
                    help="uniform: i.i.d. bases (BASELINE's synthetic sets); diverged: half the records are diverged copies of the other half -- the repeat structure of a "
                         "genome: 30 % (k=31) to 60 % (k=63) of the k-mers behind colliding minimizers, answered by the fallback function (not a BASELINE configuration)")
    ap.add_argument("--cache-dir", default=os.environ.get("LPHQ_BENCH_CACHE", "/tmp/lphq_bench_cache"))
    ap.add_argument("--no-hbm-probe", action="store_true", help="skip the in-run measurement of the HBM copy ceiling (roofline.peak_measured)")
    ap.add_argument("--dist-single", action="store_true",
                    help="with one rank, still initialise torch.distributed (RCCL) and run the image broadcast and the replica check: exercises the N > 1 code on a 1-GPU box")
    ap.add_argument("--b1-records", type=int, default=300,
                    help="records of the index set the reference-faithful driver baseline (B1: gz + parse inside the timer) reads; 0 = skip")
    ap.add_argument("--no-secondary", action="store_true", help="headline only: skip the C3 / C5 / C4 measurements that follow a default (C2) run")
    ap.add_argument("--secondary-c4-kmers", type=int, default=2_000_000_000, help="k-mers of the C4 secondary workloads (BASELINE.json: 2 B)")
    ap.add_argument("--strong", action="store_true", help="also run the strong-scaling pass (fixed total work, sharded) with one rank and no process group")
    ap.add_argument("--strong-total-log2", type=int, default=0,
                    help="random mode: log2 of the fixed probe set of the strong-scaling pass (default: batch-log2 + 3 = what 8 GPUs answer per step of the weak line)")
    ap.add_argument("--plan-records", action="store_true",
                    help="with --plan-only: also run the record-set logic of the first workload (rank 0 generates and shares through --cache-dir, the other "
                         "ranks stream the file) and report source, CRC-32 and peak RSS per rank")
    ap.add_argument("--probe-ceiling", action="store_true",
                    help="re-measure the gather ceilings on this box (tools/gather_ceiling: scattered 128-B lines and the bare pilot -> block chain on tables of the "
                         "section sizes of the C2 / C5 / C4 images, one atomic append per workgroup) and write gpurun_out/gather_ceiling.json -- copy it to "
                         "profiles/ to make it what roofline.frac_of_gather_chain and the line-request ceiling are read from")
    ap.add_argument("--plan-only", action="store_true",
                    help="resolve workloads and cache paths, check that all ranks agree (gloo when WORLD_SIZE > 1), print the plan and stop before the first GPU call")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # before anything initialises the GPU in this process
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    # Everything but the result line goes to stderr, also what native libraries print (RCCL writes a version banner to stdout with C stdio, flushed
    # when the process ends, i.e. AFTER the JSON line): file descriptor 1 is pointed at stderr and the line is written to a copy of the original.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] ERROR: WORLD_SIZE={world} but --gpus {args.gpus}")
        sys.exit(2)
    plan = plan_of(args)

    if args.plan_only:  # host logic only: what tests/test_dist_cpu.py runs with world_size 2 over gloo
        ok = True
        dist = None
        if world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            if os.environ.get("LPHQ_BENCH_TEST_FAIL_RANK") == str(rank):  # test hook: a rank that fails a check
                ok = False
            agree = plans_agree(dist, torch, torch.device("cpu"), plan)
            ok = all_ranks_ok(dist, torch, torch.device("cpu"), ok and agree)
            if not ok:
                dist.destroy_process_group()
                log(f"[bench r{rank}] ERROR: a rank failed or the ranks resolved different plans (agree: {agree})")
                sys.exit(3)
        records_report = None
        if args.plan_records:
            # the host side of "every rank has the index records": rank 0 generates and shares them through the cache directory, the others stream
            # the file (what to_device does, minus the device) -- per rank: where the bases came from, their CRC-32, the process's peak RSS
            import resource

            job = Job(args, rank, local_rank, world, None, dist, torch, np, None)
            rs = job.records(plan["workloads"][0])
            mine = {"rank": rank, "source": rs.source, "crc32": rs.checksum(), "first_record_crc32": __import__("zlib").crc32(rs[0]),
                    "maxrss_mb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0}
            if dist is not None:
                gathered = [None] * world
                dist.all_gather_object(gathered, mine)
                records_report = gathered
            else:
                records_report = [mine]
        strong_report = None
        if "strong" in plan:  # what THIS rank would take of every workload's fixed job, as it works it out from its own RANK (gathered: the test checks the partition)
            mine = {"rank": rank, "shards": [strong_plan(w, world, strong_total_log2_of(args, w))["shards"][rank] for w in plan["workloads"] if w["sink"] == "codes"]}
            if dist is not None:
                gathered = [None] * world
                dist.all_gather_object(gathered, mine)
                strong_report = gathered
            else:
                strong_report = [mine]
        if dist is not None:
            dist.destroy_process_group()
        if rank == 0:
            line = {"plan": plan, "ranks": world, "ranks_agree": True}
            if strong_report is not None:
                line["strong_shards_by_rank"] = strong_report
            if records_report is not None:
                line["records"] = records_report
            result_out.write(json.dumps(line) + "\n")
            result_out.flush()
        return

    import lphash_amd as L

    assert torch.cuda.is_available(), "bench.py needs a GPU: the query path has no CPU fallback"
    if args.probe_ceiling:
        sys.exit(probe_ceiling(args, torch, np, L, result_out))
    # LPHQ_BENCH_SHARE_DEVICE0=1 (test hook for boxes with ONE GPU): every rank works on cuda:0 and the process group runs over gloo (RCCL refuses two ranks
    # on one device) -- everything but the transport of the collectives is then the N > 1 run: N processes, the records shared through the cache directory, the
    # image broadcast into another process's memory, per-rank batches and oracle checks, the sharded strong-scaling pass, the verdict collectives.
    share0 = os.environ.get("LPHQ_BENCH_SHARE_DEVICE0") == "1"
    if share0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.dist_single:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29531")
        if share0:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    job = Job(args, rank, local_rank, world, dev, dist, torch, np, L)
    if dist is not None and not all_ranks_ok(dist, torch, dev, plans_agree(dist, torch, dev, plan)):
        job.fail(2, "the ranks resolved different plans (arguments / cache paths differ between ranks)")
    os.makedirs(args.cache_dir, exist_ok=True)
    if args.sink not in ("codes", "runs", "gather1", "gather2", "gather4", "gather8") or (args.sink != "codes" and (args.mode != "stream" or args.kind not in ("p", "u"))):
        job.fail(2, "--sink runs / gather<E> goes with --mode stream and an LP-MPHF (--kind p | u)")
    if args.mode != "random" and args.kind in ("pthash", "bbhash"):
        job.fail(2, "--kind pthash / bbhash is measured in random mode only")

    wls = plan["workloads"]
    t_job = time.time()
    out = run_workload(job, wls[0])
    secondary, errors = [], []
    exit_code = 0
    for wl in wls[1:]:
        t0 = time.time()
        try:
            e = run_workload(job, wl)
            if rank == 0:
                e = secondary_entry(e, wl)
                e["wall_s"] = time.time() - t0
                secondary.append(e)
        except SystemExit as ex:
            # a parity failure (every rank reaches job.fail together: the verdict is a collective) ends the job with that code -- but the headline was
            # measured and checked before: rank 0 still prints it, with the failed workload named in secondary_errors
            exit_code = ex.code if isinstance(ex.code, int) and ex.code else 3
            why = getattr(ex, "msg", None) or f"exit code {exit_code}"
            err = {"name": wl["name"], "error": why[:300], "exit_code": exit_code}
            if exit_code == 3:  # the only code that is a verdict on the results; anything else is named for what it is
                err["parity_vs_oracle"] = False
            errors.append(err)
            if job.dist is not None:
                job.dist = None  # (job.fail has torn the process group down)
            dist = None
            break
        except Exception as ex:  # noqa: BLE001 -- the headline must survive e.g. a host without the memory for the 2 B-k-mer set
            if dist is not None:
                raise  # (the other ranks are inside collectives of this workload: no way to skip it on one rank alone)
            log(f"[bench] secondary {wl['name']} failed: {type(ex).__name__}: {ex}")
            errors.append({"name": wl["name"], "error": f"{type(ex).__name__}: {ex}"[:300]})
            torch.cuda.empty_cache()
    if rank == 0:
        if len(wls) > 1:
            out["secondary"] = secondary
            if errors:
                out["secondary_errors"] = errors
        out["bench_wall_s"] = time.time() - t_job
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
        log(summary_table(out))  # the last thing on stderr
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
