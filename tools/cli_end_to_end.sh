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
idx = "/tmp/c2_cli.lph"
t = time.time(); L.build(recs, 31, 20, idx, kmer_bits=64, threads=64, device=0); print("index built in %.1f s" % (time.time() - t))
for q in ("/tmp/c2.fa", "/tmp/reads150.fq", "/tmp/c2_300.fa.gz", "/tmp/c2_300.bgzf.fa.gz"):
    for extra in ([], ["--threads", "8"], ["--threads", "32"], ["--pageable"], ["--batch-mb", "8"], ["--batch-mb", "128"]):
        t = time.time()
        r = subprocess.run(["lphash_amd/lphash", "query-p", "-i", idx, "-q", q, "--kmer-bits", "64"] + extra, capture_output=True, text=True)
        print(os.path.basename(q), extra, "k-mers, ns/k-mer streaming, random:", r.stdout.strip().split(",")[2:], "wall %.2f s" % (time.time() - t), r.stderr[-200:])
PY
