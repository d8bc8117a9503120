"""Time the index build with the CPU and the GPU pilot search on the C2 synthetic set (scratch tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lphash_amd as L
from lphash_amd.synth import synthetic_records
nrec = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
_, recs = synthetic_records(seed=1, n_records=nrec, length=100_000 + k - 1)
bits = 64 if k <= 31 else 128
t = time.time(); sg = L.build(recs, k, 20, "/tmp/b_gpu.lph", kmer_bits=bits, threads=0, device=0); tg = time.time() - t
print(f"gpu-search build: {tg:.2f}s", sg, flush=True)
if "--cpu" in sys.argv:
    t = time.time(); sc = L.build(recs, k, 20, "/tmp/b_cpu.lph", kmer_bits=bits, threads=0); tc = time.time() - t
    print(f"cpu-search build: {tc:.2f}s identical={open('/tmp/b_cpu.lph','rb').read() == open('/tmp/b_gpu.lph','rb').read()}")
