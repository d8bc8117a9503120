"""What the GPU box's host really offers: visible CPUs, cgroup CPU quota, load, and how the oracle's packed lookups scale with POSIX threads (scratch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/loadavg"):
    try:
        print(p, open(p).read().strip())
    except Exception as e:
        print(p, "-", type(e).__name__)
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
import lphash_amd as L, orc
from lphash_amd.synth import synthetic_records
_, recs = synthetic_records(seed=1, n_records=100, length=100_030)
L.build(recs, 31, 20, "/tmp/probe.lph", kmer_bits=64, threads=16)
o = orc.Index("/tmp/probe.lph", 0, 64)
big = np.random.default_rng(1).integers(0, 1 << 62, 1 << 25, dtype=np.uint64)
res = np.zeros(len(big), dtype=np.uint64)
for nt in (1, 4, 16, 32, 64, 128, 256):
    n = min(len(big), (1 << 21) * nt)
    t0 = time.perf_counter(); o.query_packed_threads(big[:n], nt, res); dt = time.perf_counter() - t0
    print(f"{nt:4d} threads: {n / dt / 1e6:8.1f} M lookups/s ({n / dt / 1e6 / nt:.2f} per thread)")
