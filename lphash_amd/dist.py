"""Multi-GPU plumbing: one process per GPU, index replicated, query batches sharded.

The reference has no distributed code at all; lookups are independent pure functions of (index, k-mer), so
the only exchange is a one-time broadcast of the flat device image (RCCL over xGMI when the tensors are on
GPUs; the same code runs over gloo on CPU tensors in the tests).  There is no data-path collective.
"""
import numpy as np
import torch


def shard_range(n_units, rank, world):
    """Contiguous shard [begin, end) of n_units for `rank`; the shards partition [0, n_units)."""
    base, rem = divmod(int(n_units), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_records(lengths, rank, world):
    """Split records into `world` contiguous groups of roughly equal base count; returns (first, last+1)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    if len(lengths) == 0:
        return 0, 0
    cum = np.cumsum(lengths)
    total = int(cum[-1])
    # shard r starts at the record boundary NEAREST to r / world of the bases (equal-length records split evenly: 1000 records, 2 ranks -> 500 + 500; a long
    # record that straddles the target goes to whichever side leaves the smaller imbalance, so every non-empty shard is within one record of its ideal)
    bounds = []
    for r in range(world):
        target = (total * r) // world
        i = int(np.searchsorted(cum, target, side="right"))  # records [0, i) end at or before the target
        before = int(cum[i - 1]) if i > 0 else 0
        after = int(cum[i]) if i < len(cum) else total
        bounds.append(i if target - before <= after - target else i + 1)
    bounds.append(len(lengths))
    bounds[0] = 0
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds[rank], bounds[rank + 1]


def broadcast_image(image, dist, device, src=0):
    """Broadcast the flat image (numpy uint8 on `src`, None elsewhere) to every rank.

    Returns a uint8 tensor on `device` holding the image on every rank.  With backend "nccl" (= RCCL on ROCm)
    and CUDA tensors this is the single collective of the whole job."""
    rank = dist.get_rank()
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        n[0] = int(image.size)
    dist.broadcast(n, src=src)
    t = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    if rank == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(image)))
    dist.broadcast(t, src=src)
    return t


def replicate_index(cls, path, kmer_bits, dist, local_rank, src=0):
    """Rank `src` parses `path`; every rank ends up with a handle attached to cuda:<local_rank>."""
    device = torch.device("cuda", local_rank)
    rank = dist.get_rank()
    if rank == src:
        h = cls.load(path, kmer_bits=kmer_bits, device=local_rank)
        image = h.image()
    else:
        h, image = None, None
    t = broadcast_image(image, dist, device, src=src)
    if rank != src:
        h = cls.from_device_image(t.data_ptr(), t.numel(), local_rank)
        h._image_tensor = t  # keep the broadcast buffer alive: the handle does not own it
    return h
