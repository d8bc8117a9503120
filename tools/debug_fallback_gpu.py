"""Debug aid for profiles/r06_transposed_gather_attempt.txt: a build with the hooks of profiles/attempts/r06_debug_hooks.patch (-DLPHQ_DEBUG_S64 -DLPHQ_DEBUG_WHAT=<expr>) dumps one value per window under a colliding
minimizer (by base position); the dump of launch n is written to LPHQ_DBG_FILE by launch n + 1.  Compared here with the truth for the windows launch n got wrong."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["LPHQ_DBG_FILE"] = "/tmp/lphq_dbg.bin"
import lphash_amd as L  # noqa: E402
import orc  # noqa: E402
from real_fixture import CASES, OUT, load_inputs, unitig_records  # noqa: E402
from test_gpu_runs import DeviceBatch  # noqa: E402
from util import pack_kmers  # noqa: E402

what = sys.argv[1]  # lo | hi | code
name = "se_k47_m20_b128_u"
kind, k, m, bits, ust = CASES[name]
recs = unitig_records(load_inputs(), ust)
path = os.path.join(OUT, name + ".lph")
if not os.path.exists(path):
    path = "/tmp/dbg_%s.lph" % name
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8, device=0)
dev = torch.device("cuda", 0)
h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
o = orc.Index(path, kind, bits)
nk = h.get_kmer_count()
short = [r for r in recs if len(r) < 900]
one = np.concatenate([o(q, False) for q in short])
reps = 100
b = DeviceBatch(torch, dev, short * reps, k, 0)
want = torch.from_numpy(np.tile(one, reps).view(np.int64)).to(dev)
table = torch.arange(nk, dtype=torch.int64, device=dev)
outs = []
for launch in range(2):
    out = torch.full((b.n_out + 16,), 99, dtype=torch.int64, device=dev)
    h.query_records_gather(b.d_seq, b.n_bases, b.d_so.data_ptr(), b.n_records, b.d_oo.data_ptr(), table.data_ptr(), 8, out.data_ptr(), missing=7)
    torch.cuda.synchronize()
    outs.append(out)
out = outs[0]
d = np.fromfile("/tmp/lphq_dbg.bin", dtype=np.uint64)  # launch 0's dump
bad = torch.nonzero(out[: b.n_out] != want).flatten().cpu().numpy()
print("launch 0: windows that differ:", len(bad), "of", b.n_out, "; dump entries set:", int((d != 0).sum()), flush=True)
so, oo = b.seq_off, b.out_off
flat = np.frombuffer(b"".join(short), dtype=np.uint8)
one_bases = len(flat)
okc = badc = 0
shown = 0
outc = out.cpu().numpy()
wantc = want.cpu().numpy()
for w in bad:
    r = int(np.searchsorted(oo, w, side="right") - 1)
    g = int(so[r] + (w - oo[r]))
    gg = g % one_bases
    if what == "code":
        truth = int(wantc[w])
    else:
        pk = pack_kmers(flat[gg:gg + k].tobytes(), k, bits)
        truth = int(pk[0]) if what == "lo" else (int(pk[1]) if bits == 128 else 0)
    same = int(d[g]) == truth
    okc += same
    badc += not same
    if shown < 12:
        shown += 1
        print(f"  window {w} base {g}: dumped {what} {int(d[g]):016x} true {truth:016x} {'==' if same else '!='} | got code {int(outc[w])} want {int(wantc[w])}" + (f" dumped==got {int(d[g]) == int(outc[w])}" if what == 'code' else ''))
print(f"over the {len(bad)} wrong windows: dumped {what} right {okc}, wrong {badc}")
# the windows that were RIGHT: is the dump right there?
import random
good = np.setdiff1d(np.nonzero(d[: b.n_bases] != 0)[0], np.array([int(so[int(np.searchsorted(oo, w, side='right') - 1)] + (w - oo[int(np.searchsorted(oo, w, side='right') - 1)])) for w in bad]))
print("dump entries at windows that came out right:", len(good))
