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
