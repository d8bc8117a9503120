#!/bin/bash
# End-to-end `lphash` CLI on a synthetic 100 Mbase FASTA (plain and gz): build-p, then query-p (CSV = file,mphf,kmers,ns/kmer streaming,ns/kmer random)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from lphash_amd.synth import synthetic_records
flat, recs = synthetic_records(1, 1000, 100030)
with open('/tmp/c2.fa', 'wb') as f:
    for i, r in enumerate(recs):
        f.write(b'>rec%d\n' % i + r + b'\n')
PY
gzip -k -1 -f /tmp/c2.fa
time ./lphash_amd/lphash build-p -i /tmp/c2.fa -k 31 -m 20 -o /tmp/c2_cli.lph --kmer-bits 64 -t 64
time ./lphash_amd/lphash query-p -i /tmp/c2_cli.lph -q /tmp/c2.fa --kmer-bits 64
time ./lphash_amd/lphash query-p -i /tmp/c2_cli.lph -q /tmp/c2.fa.gz --kmer-bits 64
