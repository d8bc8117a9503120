import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import lphash_amd as L, orc
from runs_util import *
import test_gpu_runs as G
from real_fixture import CASES, OUT, load_inputs, unitig_records
name = sys.argv[1]
kind, k, m, bits, ust = CASES[name]
z = load_inputs(); recs = unitig_records(z, ust)
path = os.path.join(OUT, name + '.lph')
if not os.path.exists(path):
    path = '/tmp/' + name + '.lph'; L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8, device=0)
h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
dev = torch.device('cuda:0')
b = G.DeviceBatch(torch, dev, recs, k, 3)
d_c = G.gpu_codes(h, torch, dev, b)
exp = d_c.cpu().numpy().view(np.uint64)
for trial in range(3):
    runs, codes, counts, dv = G.gpu_runs(h, torch, dev, b, per_arena=b.n_out // ARENAS + 4096)
    got = expand_runs_numpy(runs, codes, counts, b.n_out, L.INVALID)
    bad = np.nonzero(got != exp)[0]
    d_e, st = G.gpu_expand(h, torch, dev, dv, b.n_out)
    bad2 = np.nonzero(d_e.cpu().numpy().view(np.uint64) != exp)[0]
    print('trial', trial, 'runs', int(counts['runs'].sum()), 'xc', int(counts['codes'].sum()), 'numpy-expand mismatches', len(bad), bad[:8], 'gpu-expand mismatches', len(bad2), bad2[:8], 'st', st)
    for i in list(bad[:3]) + list(bad2[:3]):
        print(' window', i, 'got', got[i - 2:i + 3], 'exp', exp[i - 2:i + 3], 'gpu', d_e[i - 2:i + 3].cpu().numpy().view(np.uint64))
        for a in range(ARENAS):
            n = int(counts[a]['runs']); f, l, kd = run_fields(runs[a][:n])
            for j in np.nonzero((f <= i) & (i < f + l))[0]:
                print('   arena', a, 'run', j, 'first', f[j], 'len', l[j], 'kind', kd[j], 'code', runs[a][j]['code'], 'xc', codes[a][int(runs[a][j]['code']):int(runs[a][j]['code']) + 4] if kd[j] == 2 else '')

# where do the wrong values come from?
runs, codes, counts, dv = G.gpu_runs(h, torch, dev, b, per_arena=b.n_out // ARENAS + 4096)
where = {}  # expected code -> window
slot_of = {}  # window -> (arena, slot)
for a in range(ARENAS):
    n = int(counts[a]['runs']); f, l, kd = run_fields(runs[a][:n])
    for j in np.nonzero(kd == 2)[0]:
        for i in range(int(l[j])):
            slot_of[int(f[j]) + i] = (a, int(runs[a][j]['code']) + i)
for w in slot_of:
    where[int(exp[w])] = w
nbad = 0
for w, (a, s) in sorted(slot_of.items()):
    v = int(codes[a][s])
    if v != int(exp[w]):
        nbad += 1
        src = where.get(v)
        if nbad <= 12:
            print('window', w, 'slot', (a, s), 'holds', v, 'which is the code of window', src, 'whose slot is', slot_of.get(src), 'delta windows', None if src is None else src - w)
print('bad explicit slots', nbad, 'of', len(slot_of))
