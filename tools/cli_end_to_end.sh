#!/bin/bash
# End-to-end `lphash query-p` on the C2 set written as plain FASTA and as .fa.gz (the reference's timers include inflate + parse,
# src/query.cpp:48-56, and so do these): gpurun --timeout 900 -- bash tools/cli_end_to_end.sh
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
python - <<'PY' 2>&1 | tee gpurun_out/cli_end_to_end.txt
import gzip, os, subprocess, sys, time
sys.path.insert(0, ".")
import lphash_amd as L
from lphash_amd.synth import synthetic_records
flat, recs = synthetic_records(1, 1000, 100030)
with open("/tmp/c2.fa", "wb") as f:
    for i, r in enumerate(recs):
        f.write(b">s%d\n" % i); f.write(r); f.write(b"\n")
with gzip.open("/tmp/c2_300.fa.gz", "wb", compresslevel=6) as f:
    for i, r in enumerate(recs[:300]):
        f.write(b">s%d\n" % i); f.write(r); f.write(b"\n")
import struct, zlib
def bgzf(data, block=65280):
    out = bytearray()
    for chunk in [data[i:i + block] for i in range(0, len(data), block)] + [b""]:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = c.compress(chunk) + c.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(payload) + 25) + payload + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)
open("/tmp/c2_300.bgzf.fa.gz", "wb").write(bgzf(b"".join(b">s%d\n" % i + r + b"\n" for i, r in enumerate(recs[:300]))))
import numpy as np
rng = np.random.default_rng(5)
starts = rng.integers(0, len(flat) - 150, 2_000_000)
with open("/tmp/reads150.fq", "wb") as f:  # 2 M reads of 150 bp sampled from the index set, four-line FASTQ
    qual = b"I" * 150
    chunks = []
    for i, st in enumerate(starts):
        chunks.append(b"@read%d\n%s\n+\n%s\n" % (i, flat[st:st + 150].tobytes(), qual))
        if len(chunks) == 100000:
            f.write(b"".join(chunks)); chunks = []
    f.write(b"".join(chunks))
import itertools
with gzip.open("/tmp/reads150_500k.fq.gz", "wb", compresslevel=6) as f, open("/tmp/reads150.fq", "rb") as g:
    f.writelines(itertools.islice(g, 4 * 500_000))  # what a user's input usually is: reads as ONE gzip member (0.5 M reads of 150 bp, ~160 MB of text)
idx = "/tmp/c2_cli.lph"
t = time.time(); L.build(recs, 31, 20, idx, kmer_bits=64, threads=64, device=0); print("index built in %.1f s" % (time.time() - t))
with gzip.open("/tmp/c2_1000.fa.gz", "wb", compresslevel=6) as f:  # the whole index set as one gzip member (100 MB of text)
    for i, r in enumerate(recs):
        f.write(b">s%d\n" % i); f.write(r); f.write(b"\n")
for q in ("/tmp/c2.fa", "/tmp/reads150.fq", "/tmp/c2_300.fa.gz", "/tmp/c2_1000.fa.gz", "/tmp/reads150_500k.fq.gz", "/tmp/c2_300.bgzf.fa.gz"):
    for extra in ([], ["--threads", "8"], ["--threads", "32"], ["--pageable"], ["--batch-mb", "8"], ["--batch-mb", "128"]):
        t = time.time()
        r = subprocess.run(["lphash_amd/lphash", "query-p", "-i", idx, "-q", q, "--kmer-bits", "64"] + extra, capture_output=True, text=True)
        print(os.path.basename(q), extra, "k-mers, ns/k-mer streaming, random:", r.stdout.strip().split(",")[2:], "wall %.2f s" % (time.time() - t), r.stderr[-200:])
# single-member .gz: the serial decoder (round 3) against the span-parallel one (round 4), CLI and decoder alone
for q in ("/tmp/c2_300.fa.gz", "/tmp/c2_1000.fa.gz"):
    r = subprocess.run(["lphash_amd/lphash", "query-p", "-i", idx, "-q", q, "--kmer-bits", "64"], capture_output=True, text=True, env=dict(os.environ, LPHQ_GZ_SERIAL="1"))
    print(os.path.basename(q), "LPHQ_GZ_SERIAL=1 (serial decoder):", r.stdout.strip().split(",")[2:])
    for span in ("131072", "262144", "524288", "1048576"):
        r = subprocess.run(["lphash_amd/lphash", "query-p", "-i", idx, "-q", q, "--kmer-bits", "64"], capture_output=True, text=True, env=dict(os.environ, LPHQ_GZ_SPAN=span))
        print(os.path.basename(q), "span", span, r.stdout.strip().split(",")[2:])
    for args in (["--zlib"], [], ["--parallel", "4"], ["--parallel", "8"], ["--parallel", "16"], ["--parallel", "32"], ["--parallel", "16", "--span", "131072"],
                 ["--parallel", "16", "--span", "262144"], ["--parallel", "16", "--span", "1048576"], ["--parallel", "32", "--span", "131072"]):
        best = None
        for rep in range(3):
            r = subprocess.run(["lphash_amd/lphash", "inflate", "-i", q] + args, capture_output=True, text=True)
            f = r.stdout.strip().split(",")
            best = f if best is None or float(f[2]) < float(best[2]) else best
        print("inflate", os.path.basename(q), args, "bytes, status, s (best of 3), pieces parallel / serial, span:", best, "-> %.2f GB/s" % (int(best[0]) / float(best[2]) / 1e9))
r = subprocess.run(["lphash_amd/lphash", "inflate", "-i", "/tmp/c2_300.fa.gz", "--parallel", "16"], capture_output=True, text=True, env=dict(os.environ, LPHQ_PGZ_TRACE="1"))
print(r.stderr)
PY
