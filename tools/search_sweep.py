#!/usr/bin/env python3
"""Sweep of the GPU pilot search's tuning aids on one box (index construction, lphash_amd/csrc/search_rounds.hpp): the same synthetic set is built once per setting, the stage
times come from LPHQ_BUILD_VERBOSE, the file's SHA-256 must not move.   gpurun -- python tools/search_sweep.py [c2|c4]"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
k, m, bits, n_rec = (31, 20, 64, 1000) if which == "c2" else (63, 20, 128, 20_000)
SETTINGS = [{}, {"LPHQ_SEARCH_LOOKAHEAD": "65536"}, {"LPHQ_SEARCH_LOOKAHEAD": "65536", "LPHQ_SEARCH_STEPS": "64"}, {"LPHQ_SEARCH_STEPS": "64"}, {"LPHQ_SEARCH_ROUNDS_PER_SYNC": "64"}]
with tempfile.TemporaryDirectory() as d:
    for st in SETTINGS:
        env = dict(os.environ, LPHQ_BUILD_VERBOSE="1", **st)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "search_sweep_child.py"), ROOT, str(k), str(m), str(bits), str(n_rec), os.path.join(d, "x.lph")], capture_output=True, text=True, env=env, timeout=1200)
        rounds = re.findall(r"gpu pilot search: (\d+) buckets in (\d+) rounds", r.stderr)
        secs = re.findall(r"pilot search \(GPU\)\s+([0-9.]+) s", r.stderr)
        sha = re.findall(r"sha256 (\w+)", r.stdout)
        bs = re.findall(r"build_s ([0-9.]+)", r.stdout)
        print(which, st or "defaults", "| minimizer function:", rounds[:1], secs[:1], "s | whole build", bs, "s | sha256", (sha[0][:16] if sha else r.stderr[-300:]), flush=True)
