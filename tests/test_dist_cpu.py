"""world_size-2 gloo tests of the multi-GPU host logic (lphash_amd/dist.py): shards partition the batch, and the
flat image arrives bit-identical on the non-root rank."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import lphash_amd as L
from lphash_amd.dist import broadcast_image, shard_range, shard_records

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 1000, 2**28 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_records_partitions():
    rng = np.random.default_rng(0)
    lengths = rng.integers(0, 5000, 333)
    for world in (1, 2, 4, 8):
        spans = [shard_records(lengths, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == len(lengths)
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_shard_records_skewed_lengths_stay_within_one_record_of_the_ideal():
    """a record that straddles a shard boundary goes to the side that leaves the smaller imbalance -- [1000, 1, 1, 1] over two ranks is
    (record 0) + (records 1-3), not (nothing) + (everything); every boundary lies within one record of its ideal base count."""
    assert [shard_records([1000, 1, 1, 1], r, 2) for r in range(2)] == [(0, 1), (1, 4)]
    rng = np.random.default_rng(7)
    for trial in range(50):
        lengths = np.where(rng.random(200) < 0.05, rng.integers(50_000, 500_000, 200), rng.integers(0, 300, 200))
        total, cum = int(lengths.sum()), np.concatenate([[0], np.cumsum(lengths)])
        for world in (2, 3, 8):
            spans = [shard_records(lengths, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == len(lengths) and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            for r in range(1, world):
                b = spans[r][0]
                ideal = total * r // world
                left = int(lengths[b - 1]) if b > 0 else 0
                right = int(lengths[b]) if b < len(lengths) else 0
                assert abs(int(cum[b]) - ideal) <= max(left, right), (trial, world, r)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        image = L.Mphf.load(path, kmer_bits=64).image() if rank == 0 else None
        t = broadcast_image(image, dist, torch.device("cpu"), src=0)
        ref = L.Mphf.load(path, kmer_bits=64).image()
        ok = bool(np.array_equal(t.numpy(), ref)) and bytes(t[:8].numpy()) == b"LPQIMG01"
        # shards of a batch are disjoint and cover it: exchange the spans and check
        a, b = shard_range(1000003, rank, world)
        spans = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(spans, torch.tensor([a, b], dtype=torch.int64))
        cover = sorted((int(s[0]), int(s[1])) for s in spans)
        ok = ok and cover[0][0] == 0 and cover[-1][1] == 1000003 and all(cover[i][1] == cover[i + 1][0] for i in range(world - 1))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_image_broadcast_gloo_world2():
    path = os.path.join(ROOT, "tests", "golden", "p_k31_m20_b64.lph")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}


def test_bench_launches_its_own_ranks_when_asked_for_several_gpus():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset): the parent must start the ranks itself through
    torch.distributed.run -- as a child process, before it touches the GPU -- and exit with the children's code.  Without a GPU (here) the
    ranks fail on their `torch.cuda.is_available()` check, so the observable behaviour is: the launch line on stderr, a non-zero exit
    code, no JSON line claiming a 1-GPU result."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--kmers", "100000"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr
    assert r.returncode != 0
    assert '"n_gpus"' not in r.stdout


def _bench_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location("lphq_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run_bench(argv, env_extra=None, timeout=600):
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LPHQ_BENCH_TEST_FAIL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_plan_names_every_baseline_configuration():
    """The default run = the C2 headline followed by C3, C5 and C4 (random, streaming) at BASELINE.json's sizes; any other headline, and the
    profiling form (--no-cpu-baseline), measure the named workload only."""
    import json

    r = _run_bench(["--plan-only", "--steps", "20", "--warmup", "5"])
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.loads(r.stdout.strip().splitlines()[-1])["plan"]
    names = [w["name"] for w in plan["workloads"]]
    # C3-runs / C4-stream-runs / C4-stream-gather1 (round 6): the streaming pass with the result the reference computes before it expands it, and fused with its
    # consumer; the last two: BASELINE.md section 3's secondary inputs
    assert names == ["C2", "C3", "C3-runs", "C5", "C4-random", "C4-stream", "C4-stream-runs", "C4-stream-gather1", "C2-nonmember50", "C3-shuffled"]
    w = {x["name"]: x for x in plan["workloads"]}
    ip = dict(zip(names, plan["index_paths"]))
    assert w["C2-nonmember50"]["nonmember_frac"] == 0.5 and w["C3-shuffled"]["shuffle_records"] is True
    assert [w[n]["sink"] for n in names] == ["codes", "codes", "runs", "codes", "codes", "codes", "runs", "gather1", "codes", "codes"]
    assert ip["C2-nonmember50"] == ip["C2"] == ip["C3-shuffled"] == ip["C3-runs"] and "strong" not in plan  # (one rank, no --strong: the weak line only)
    c4 = w["C4-random"]
    assert (c4["k"], c4["m"], c4["kmers"], c4["kmer_bits"], c4["batch_log2"], c4["steps"], c4["warmup"]) == (63, 20, 2_000_000_000, 128, 28, 20, 5)
    assert w["C3"]["mode"] == "stream" and w["C5"]["kind"] == "u" and w["C4-stream"]["mode"] == "stream"
    assert ip["C2"] == ip["C3"] and ip["C4-random"] == ip["C4-stream"] == ip["C4-stream-runs"] == ip["C4-stream-gather1"]  # one index per (kind, k, m, n)
    assert len(set(plan["index_paths"])) == 3
    for extra in (["--mode", "stream"], ["--k", "63"], ["--no-secondary"], ["--no-cpu-baseline"], ["--kmers", "1000000"], ["--nonmember-frac", "0.5"]):
        r = _run_bench(["--plan-only"] + extra)
        assert r.returncode == 0, r.stderr[-2000:]
        assert len(json.loads(r.stdout.strip().splitlines()[-1])["plan"]["workloads"]) == 1, extra


def test_bench_rank_logic_gloo_world2(tmp_path):
    """bench.py's rank logic up to the first GPU call, two ranks over gloo (`--plan-only`, launched the way the driver launches a scaling run):
    the arguments reach both ranks, both resolve the same workloads and cache paths, rank 0 alone prints the line; a rank that fails a check
    takes the whole job down with a non-zero code instead of leaving the other rank waiting in a collective."""
    import json

    args = ["--gpus", "2", "--plan-only", "--steps", "7", "--warmup", "3", "--cache-dir", str(tmp_path / "cache")]
    r = _run_bench(args)  # no launcher: bench.py starts torch.distributed.run itself
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["ranks"] == 2 and line["ranks_agree"] is True
    assert line["plan"]["gpus"] == 2 and line["plan"]["cache_dir"] == str(tmp_path / "cache")
    assert all(w["steps"] == 7 and w["warmup"] == 3 for w in line["plan"]["workloads"]) and len(line["plan"]["workloads"]) == 10
    assert all(p.startswith(str(tmp_path / "cache")) for p in line["plan"]["index_paths"])
    _check_strong_partition(line, 2)
    r = _run_bench(args, {"LPHQ_BENCH_TEST_FAIL_RANK": "1"})
    assert r.returncode != 0 and '"plan"' not in r.stdout
    assert "a rank failed" in r.stderr
    # WORLD_SIZE that contradicts --gpus: refused before anything else happens
    r = _run_bench(["--gpus", "1", "--plan-only"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2 but --gpus 1" in r.stderr


def _check_strong_partition(line, world):
    """The strong-scaling plan (fixed total work): per workload the ranks' shards -- as EACH RANK worked its own out -- are the plan's, contiguous, without
    overlap, and cover the fixed job: 2^31 probes (random; = 8 GPUs x the weak line's 2^28) or whole passes over the records adding up to >= 2^32 windows
    (streaming; round 6: one pass -- 0.5 ms of one GPU at k = 31 -- cut eight ways measured launch overhead, not scaling).  The workloads that only change what
    the streaming pass leaves behind (runs, payload) have no strong line of their own."""
    plan, by_rank = line["plan"], sorted(line["strong_shards_by_rank"], key=lambda e: e["rank"])
    wls = [w for w in plan["workloads"] if w["sink"] == "codes"]
    assert [e["rank"] for e in by_rank] == list(range(world)) and len(plan["strong"]) == len(wls) and [sp["name"] for sp in plan["strong"]] == [w["name"] for w in wls]
    for wi, (wl, sp) in enumerate(zip(wls, plan["strong"])):
        shards = [e["shards"][wi] for e in by_rank]
        assert shards == sp["shards"] and len(shards) == world
        assert shards[0][0] == 0 and shards[-1][1] == sp["total"] and all(a[1] == b[0] for a, b in zip(shards, shards[1:])) and all(a <= b for a, b in shards)
        if wl["mode"] == "random":
            assert sp["unit"] == "probes" and sp["total"] == 1 << (wl["batch_log2"] + 3)
        else:
            wpr = wl["rec_len"] - wl["k"] + 1
            assert sp["unit"] == "records" and sp["total"] == wl["n_records"] * sp["passes"] and sp["total"] * wpr >= 1 << 32 > (sp["passes"] - 1) * wl["n_records"] * wpr
        sizes = [b - a for a, b in shards]
        assert max(sizes) - min(sizes) <= 1  # equal-length records / probes: the shards differ by at most one unit


def test_bench_strong_plan_partitions_the_fixed_job_gloo_world8(tmp_path):
    """`--gpus 8 --plan-only`: eight ranks over gloo each take their shard of every workload's fixed job (C4: 2^31 probes -> 2^28 per GPU; 20,000 records ->
    2,500 per GPU), and the shards partition it; ragged cases of the two sharding functions beside it."""
    import json

    from lphash_amd.dist import shard_range, shard_records

    r = _run_bench(["--gpus", "8", "--plan-only", "--cache-dir", str(tmp_path / "cache")], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    _check_strong_partition(line, 8)
    by_name = {sp["name"]: sp for sp in line["plan"]["strong"]}
    c4r, c4s, c3 = by_name["C4-random"], by_name["C4-stream"], by_name["C3"]
    assert c4r["shards"][3] == [3 << 28, 4 << 28] and c4s["passes"] == 3 and c4s["shards"][7] == [52_500, 60_000] and c3["passes"] == 43
    # a shard must be LONG against what a step costs besides the kernel: the launch + barrier cost of this pool, measured by the one-GPU --dist-single run of the
    # round (profiles/r06_launch_barrier_cost.json: an empty step between the same barriers), x 10 -- the plan's own estimate from the round-5 single-GPU rates
    cost = json.load(open(os.path.join(ROOT, "profiles", "r06_launch_barrier_cost.json")))
    floor_ms = 10.0 * cost["empty_step_ms"]
    for sp in line["plan"]["strong"]:
        assert len(sp["planned_shard_ms"]) == 8 and min(sp["planned_shard_ms"]) >= max(2.0, floor_ms), (sp["name"], sp["planned_shard_ms"], floor_ms)
    for n, w in ((10, 3), (7, 8), (0, 4), (1 << 31, 6)):
        cuts = [shard_range(n, r_, w) for r_ in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    lens = [5, 0, 100, 3, 3, 3, 50, 1]
    cuts = [shard_records(lens, r_, 3) for r_ in range(3)]
    assert cuts[0][0] == 0 and cuts[-1][1] == len(lens) and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))


def test_bench_record_set_is_shared_not_regenerated_gloo_world8(tmp_path):
    """The first 8-GPU run, minus the GPUs (`--plan-only --plan-records`, eight ranks over gloo, launched as the driver launches a scaling run): rank 0
    generates the synthetic records once and shares them through the cache directory; ranks 1-7 stream that file and never hold the set -- same CRC-32
    on every rank, `shared_file` as their source, and a peak RSS well below rank 0's (which holds the array).  At C4 this is 2 GB of bases: eight
    private copies plus eight per-record copies were 32 GB of host memory and eight generations."""
    import json

    kmers = 160_000_000  # 1600 records x 100030 bases = 160 MB of bases
    args = ["--gpus", "8", "--plan-only", "--plan-records", "--kmers", str(kmers), "--no-secondary", "--cache-dir", str(tmp_path / "cache")]
    r = _run_bench(args, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["ranks"] == 8 and line["ranks_agree"] is True and len(line["plan"]["workloads"]) == 1
    rec = sorted(line["records"], key=lambda e: e["rank"])
    assert [e["rank"] for e in rec] == list(range(8))
    assert rec[0]["source"] == "generated" and all(e["source"] == "shared_file" for e in rec[1:])
    assert len({e["crc32"] for e in rec}) == 1 and len({e["first_record_crc32"] for e in rec}) == 1
    size_mb = 1600 * 100030 / 1e6
    assert rec[0]["maxrss_mb"] > size_mb  # rank 0 holds the array
    for e in rec[1:]:
        assert e["maxrss_mb"] < rec[0]["maxrss_mb"] - 0.6 * size_mb, rec  # the others hold one 64-MB chunk of it at a time
    shared = tmp_path / "cache" / "records_seed1_1600x100030.u8"
    assert shared.exists() and shared.stat().st_size == 1600 * 100030


def _agree_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bench = _bench_module()
        cpu = torch.device("cpu")
        same = bench.plans_agree(dist, torch, cpu, {"cache_dir": "/tmp/x", "k": 31})
        differ = bench.plans_agree(dist, torch, cpu, {"cache_dir": "/tmp/x" if rank == 0 else "/tmp/y", "k": 31})
        ok_all = bench.all_ranks_ok(dist, torch, cpu, True)
        ok_one_bad = bench.all_ranks_ok(dist, torch, cpu, rank != 1)
        q.put((rank, (same, differ, ok_all, ok_one_bad)))
    finally:
        dist.destroy_process_group()


def test_bench_plan_agreement_and_verdict_collectives_gloo_world2():
    """plans_agree / all_ranks_ok: a cache path that differs on one rank, or one rank whose parity check failed, is seen by EVERY rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == {0: (True, False, True, False), 1: (True, False, True, False)}
