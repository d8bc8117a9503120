"""Debug aid (profiles/r06_transposed_gather_attempt.txt): the register-retention probe (tools/vgpr_alloc_probe.hip, a wave holds v8..v55 and sleeps) on one stream while the payload
sink answers the fallback-rich batch on another, in one process: do the probe's waves lose registers when the waves beside them are the payload sink's?  Run with the shipped library
and with a build of the sink that allocates 56 VGPRs."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lphash_amd as L  # noqa: E402
import orc  # noqa: E402
from real_fixture import CASES, OUT, load_inputs, unitig_records  # noqa: E402
from test_gpu_runs import DeviceBatch  # noqa: E402

so = "/tmp/libvgprprobe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", "-w", os.path.join(ROOT, "tools", "vgpr_alloc_probe.hip"), "-o", so])
probe = C.CDLL(so)
probe.probe_launch.restype = C.c_int
probe.probe_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int]
name = "se_k47_m20_b128_u"
kind, k, m, bits, ust = CASES[name]
recs = unitig_records(load_inputs(), ust)
path = os.path.join(OUT, name + ".lph")
if not os.path.exists(path):
    path = "/tmp/dbg_%s.lph" % name
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8, device=0)
dev = torch.device("cuda", 0)
h = L.MphfAlt.load(path, kmer_bits=bits, device=0)
o = orc.Index(path, kind, bits)
short = [r for r in recs if len(r) < 900]
one = np.concatenate([o(q, False) for q in short])
reps = 100
b = DeviceBatch(torch, dev, short * reps, k, 0)
want = torch.from_numpy(np.tile(one, reps).view(np.int64)).to(dev)
table = torch.arange(h.get_kmer_count(), dtype=torch.int64, device=dev)
out = torch.empty(b.n_out + 16, dtype=torch.int64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for alloc64 in (0, 1):
    hist = torch.zeros(64, dtype=torch.int32, device=dev)
    wrong = launches = 0
    torch.cuda.synchronize()
    for it in range(60):
        for _ in range(6):  # ~6 x 0.1 ms of the payload sink ...
            h.query_records_gather(b.d_seq, b.n_bases, b.d_so.data_ptr(), b.n_records, b.d_oo.data_ptr(), table.data_ptr(), 8, out.data_ptr(), missing=7, stream=s1.cuda_stream)
            launches += 1
        rc = probe.probe_launch(C.c_void_p(s2.cuda_stream), C.c_void_p(hist.data_ptr()), 2000, 12, alloc64)  # ... beside 2000 workgroups of the probe, each holding its registers for ~6 us
        assert rc == 0
        torch.cuda.synchronize()
        wrong += int((out[: b.n_out] != want).sum())
    hh = hist.cpu().numpy()
    print(f"probe allocation {64 if alloc64 else 56}: payload-sink launches beside it {launches}, wrong windows in the last launch of each round (sum) {wrong}; "
          f"probe registers that lost their value: {({f'v{r}': int(c) for r, c in enumerate(hh) if c} or 'none')}", flush=True)
