"""Debug aid: lphq_query_records_gather on the real-genome unitigs of one tests/golden/real case, mismatches against table[codes] located (record, offset, tile).
gpurun -- python tools/debug_gather_gpu.py se_k47_m20_b128_u"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lphash_amd as L  # noqa: E402
from real_fixture import CASES, OUT, load_inputs, unitig_records  # noqa: E402
from test_gpu_runs import DeviceBatch, gpu_codes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "se_k47_m20_b128_u"
kind, k, m, bits, ust = CASES[name]
recs = unitig_records(load_inputs(), ust)
path = os.path.join(OUT, name + ".lph")
if not os.path.exists(path):
    path = "/tmp/dbg_%s.lph" % name
    L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8, device=0)
dev = torch.device("cuda", 0)
sel = sys.argv[2] if len(sys.argv) > 2 else "all"  # the index is built on all unitigs; the batch may hold only the short ones (every tile multi-record) or only the long ones
build_recs = recs
if sel == "short":
    recs = [r for r in recs if len(r) < 900] * 100
elif sel == "long":
    recs = [r for r in recs if len(r) >= 3000]
print(name, sel, len(recs), "records", flush=True)
h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
nk = h.get_kmer_count()
for mis in (3, 0):
    b = DeviceBatch(torch, dev, recs, k, mis)
    d_c = gpu_codes(h, torch, dev, b)
    if mis == 3:  # the codes sink against the oracle, once
        import orc
        o = orc.Index(path, kind, bits)
        uniq = recs[: max(1, len(recs) // (100 if sel == "short" else 1))]
        exp = np.concatenate([o(q, False) for q in uniq if len(q) >= k])
        got = d_c[: len(exp)].cpu().numpy().view(np.uint64)
        print("codes sink == oracle on the first", len(exp), "windows:", bool(np.array_equal(got, exp)), "; whole batch periodic:",
              bool(torch.equal(d_c[: len(exp)], d_c[len(exp): 2 * len(exp)])) if sel == "short" else "-", flush=True)
    for eb, dt in ((1, torch.uint8), (8, torch.int64)):
        g = torch.Generator(device=dev)
        g.manual_seed(2)
        table = torch.randint(0, 120, (nk,), generator=g, device=dev, dtype=torch.int64).to(dt)
        if eb == 8:
            table = torch.arange(nk, dtype=torch.int64, device=dev)  # the element IS the slot: a wrong element names the code the kernel used
        for rep in range(3):
            out = torch.full((b.n_out + 16,), 99, dtype=dt, device=dev)
            h.query_records_gather(b.d_seq, b.n_bases, b.d_so.data_ptr(), b.n_records, b.d_oo.data_ptr(), table.data_ptr(), eb, out.data_ptr(), missing=125)
            torch.cuda.synchronize()
            want = table[d_c.clamp(0, nk - 1)]
            bad = torch.nonzero(out[: b.n_out] != want).flatten().cpu().numpy()
            print(f"{name} misalign {mis} elem {eb} rep {rep}: {len(bad)} of {b.n_out} windows differ", flush=True)
            if len(bad) and eb == 8:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez(os.path.join(ROOT, "gpurun_out", f"gather_bad_{sel}_{mis}_{rep}.npz"), bad=bad, got=out[torch.from_numpy(bad).to(dev)].cpu().numpy(), want=want[torch.from_numpy(bad).to(dev)].cpu().numpy())
            if len(bad):
                oo = b.d_oo.cpu().numpy()
                so = b.d_so.cpu().numpy()
                W = k - m + 1
                tw = (1024 - W + 1) & ~15
                for w in bad[:12]:
                    r = int(np.searchsorted(oo, w, side="right") - 1)
                    pos = int(w - oo[r])
                    gpos = int(so[r]) + pos  # base index of the window in the batch
                    near = d_c[max(0, int(w) - 600): int(w) + 600].cpu().numpy()
                    hit = np.nonzero(near == int(out[w]))[0] if eb == 8 else []
                    print("   got == code of window(s) at distance", [int(x) - min(600, int(w)) for x in hit], end=" |")
                    print(f"   window {w}: record {r} (length {int(so[r + 1] - so[r])}) offset {pos}; base {gpos}: tile {gpos // tw} + {gpos % tw}; got {int(out[w])} want {int(want[w])} code {int(d_c[w])}")
                runs = np.split(bad, np.nonzero(np.diff(bad) != 1)[0] + 1)
                print("   stretches:", [(int(s[0]), len(s)) for s in runs[:20]], "of", len(runs))
