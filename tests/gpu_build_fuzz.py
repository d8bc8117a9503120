"""One-off robustness run of index construction on the GPU (not part of the test suite): random (k, m, width, kind), random ragged record sets with
empty / short records and non-ACGT bases, random batch splits and memory limits -- the files written with the super-k-mer extraction + sort (+ classification, for the
in-memory call) and the pilot search on the GPU (lphq_build_stream and lphq_build_device, device 0) must equal, byte for byte, the one the host-only builder writes (itself byte-identical to the oracle's
builder, tests/test_host_cpu.py).   gpurun --timeout 1200 -- python tests/gpu_build_fuzz.py [seconds]
LPHQ_FUZZ_DEVICE=-1 runs the streamed half without a GPU (batch splits and memory limits -- every vector of the construction in sorted pieces on disk,
both PTHash searches in pieces -- against the in-memory build on the host)."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lphash_amd as L  # noqa: E402
from util import dedup_records, genomic_like_records, random_records  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
dev = int(os.environ.get("LPHQ_FUZZ_DEVICE", "0"))
rng = np.random.default_rng(20261003)
tmp = tempfile.mkdtemp()
t_end = time.time() + budget
n_cfg = n_kmers = n_skipped = 0
while time.time() < t_end:
    bits = int(rng.choice([64, 128]))
    k = int(rng.integers(2, bits // 2))
    m = int(rng.integers(max(1, k - 61), min(k, 32) + 1))
    kind = int(rng.integers(0, 2))
    seed = int(rng.integers(1, 1 << 30))
    shape = int(rng.integers(0, 5))
    if shape >= 3 and k >= 8:  # round 5: sets with the shape of genomes -- repeats, diverged copies: most super-k-mers under repeated minimizers (the external
        recs = genomic_like_records(seed, int(rng.integers(5, 120)), k, int(rng.choice([400, 4000, 30_000])))  # classifier's `colliding` list, the fallback PTHash in pieces)
    elif shape == 0:  # a few long records
        recs = random_records(seed, int(rng.integers(1, 6)), 5_000, 120_000)
    elif shape == 1:  # many short ones (tens of records per tile), some shorter than k, some empty
        recs = random_records(seed, int(rng.integers(50, 2000)), 0, 4 * k + 10)
    else:
        recs = random_records(seed, int(rng.integers(3, 40)), 0, 6_000) + random_records(seed + 1, 2, 20_000, 60_000)
    recs = dedup_records(recs, k) if shape < 3 or k < 8 else recs
    if not recs:
        continue
    if rng.integers(0, 2):  # non-ACGT bases, lower case
        i = int(rng.integers(0, len(recs)))
        if len(recs[i]) > 0:
            b = bytearray(recs[i])
            for p in rng.integers(0, len(b), int(rng.integers(1, 8))):
                b[p] = ord("N")
            recs[i] = bytes(b)
        j = int(rng.integers(0, len(recs)))
        recs[j] = recs[j].lower()
    if not any(len(r) >= k for r in recs):
        continue
    host, gpu = os.path.join(tmp, "h.lph"), os.path.join(tmp, "g.lph")
    try:
        sh = L.build(recs, k, m, host, kind=kind, kmer_bits=bits, threads=8)
    except L.LphqError as e:  # duplicate k-mers left by the non-ACGT edits, or a key set PTHash cannot place: not what is tested here
        n_skipped += 1
        continue
    per_batch = int(rng.integers(1, len(recs) + 1))
    limit = int(rng.choice([0, 0, 1 << 17, 1 << 20] if dev >= 0 else [1 << 12, 1 << 15, 1 << 17, 1 << 20]))

    def batches():
        for i in range(0, len(recs), per_batch):
            yield recs[i:i + per_batch]

    sg = L.build_stream(batches, k, m, gpu, kind=kind, kmer_bits=bits, threads=8, device=dev, tmp_dir=os.path.join(tmp, "spill"), max_memory_bytes=limit)
    assert sg == sh, (k, m, bits, kind, seed, shape, per_batch, limit, sg, sh)
    assert open(gpu, "rb").read() == open(host, "rb").read(), (k, m, bits, kind, seed, shape, per_batch, limit)
    assert os.listdir(os.path.join(tmp, "spill")) == [] if limit else True
    if dev >= 0:
        sm = L.build(recs, k, m, gpu, kind=kind, kmer_bits=bits, threads=8, device=dev)  # one batch: classified on the device as well
        assert sm == sh and open(gpu, "rb").read() == open(host, "rb").read(), ("in-memory", k, m, bits, kind, seed, shape)
    n_cfg += 1
    n_kmers += sh["num_kmers"]
print(f"{'gpu' if dev >= 0 else 'host (memory limits)'} build fuzz ok: {n_cfg} random configurations ({n_skipped} skipped: unbuildable key sets), {n_kmers} k-mers, files identical to the host builder's, in {budget:.0f} s")
