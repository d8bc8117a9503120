"""Intake of files written by a REAL reference build (tests/golden/ref/README.md): what is checked for every case directory, on the CPU
(loader, oracle, both builders) and on the GPU (kernels through the C ABI).  Test infrastructure: the oracle is the checker here, and it is
itself checked -- against the reference's codes -- whenever a case supplies them."""
import glob
import gzip
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "tests", "golden", "ref")
REFERENCE_DATA = "/root/reference/data/unitigs_stitched"  # the build container only; never read by the -m gpu tests
KIND = {"p": 0, "u": 1}


def cases(base=REF_DIR):
    """Case directories (those holding meta.json and index.lph), sorted."""
    return sorted(d for d in glob.glob(os.path.join(base, "*")) if os.path.isfile(os.path.join(d, "meta.json")) and os.path.isfile(os.path.join(d, "index.lph")))


def read_fastx(path):
    """Records (bytes) of a FASTA (multi-line) or four-line FASTQ file, gzip or plain -- the sequences kseq_read yields for regular files."""
    op = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    recs, cur, mode, skip = [], None, None, 0
    with op(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if skip:
                skip -= 1
                continue
            if mode == "q+":  # the '+' line of a FASTQ record was just read: one quality line follows (four-line FASTQ)
                mode = None
                continue
            if line.startswith(b">"):
                if cur is not None:
                    recs.append(bytes(cur))
                cur, mode = bytearray(), "a"
            elif line.startswith(b"@") and mode != "a":
                if cur is not None:
                    recs.append(bytes(cur))
                cur, mode = bytearray(), "q"
            elif mode == "q" and line.startswith(b"+"):
                mode = "q+"
            elif cur is not None:
                cur += line
    if cur is not None:
        recs.append(bytes(cur))
    return recs


def _first(*paths):
    for p in paths:
        if p and os.path.isfile(p):
            return p
    return None


class Case:
    def __init__(self, d):
        self.dir = d
        self.name = os.path.basename(d)
        self.meta = json.load(open(os.path.join(d, "meta.json")))
        self.kind = KIND[self.meta["kind"]]
        self.kmer_bits, self.k, self.m = int(self.meta["kmer_bits"]), int(self.meta["k"]), int(self.meta["m"])
        self.seed, self.c = int(self.meta.get("seed", 42)), float(self.meta.get("c", 3.0))
        self.index = os.path.join(d, "index.lph")
        self.input = _first(os.path.join(d, "input.fa.gz"), os.path.join(d, "input.fa"), os.path.join(REFERENCE_DATA, os.path.basename(self.meta.get("input", "") or "-")))
        self.queries = _first(os.path.join(d, self.meta.get("queries") or "queries.fa.gz"), os.path.join(d, "queries.fa.gz"), os.path.join(d, "queries.fa"))
        self.codes = {}
        for mode in ("streaming", "random"):
            p = os.path.join(d, f"codes_{mode}.u64")
            if os.path.isfile(p):
                self.codes[mode] = np.fromfile(p, dtype="<u8")


def check_case_cpu(case, L, orc):
    """Loader accepts the file and agrees with meta.json; the oracle answers the file's codes; both builders reproduce its bytes.
    Returns the list of checks that ran (a case may lack the optional pieces)."""
    ran = []
    cls = L.Mphf if case.kind == 0 else L.MphfAlt
    h = cls.load(case.index, kmer_bits=case.kmer_bits)  # every redundancy cross-check of the loader (LPHQ_ERR_FORMAT otherwise)
    assert (h.k, h.m, h.info.mm_seed) == (case.k, case.m, case.seed), "meta.json disagrees with the file"
    nk = h.get_kmer_count()
    h.close()
    ran.append("loader")
    o = orc.Index(case.index, case.kind, case.kmer_bits)
    assert o.nkmers == nk
    if case.queries and case.codes:
        recs = read_fastx(case.queries)
        for mode, want in case.codes.items():
            got = np.concatenate([o(r, mode == "streaming") for r in recs] or [np.zeros(0, dtype=np.uint64)])
            assert np.array_equal(got, want), f"oracle codes differ from the reference's ({mode}): the PTHash recall is wrong somewhere"
            ran.append(f"oracle-codes-{mode}")
    if case.input:
        recs = read_fastx(case.input)
        want = open(case.index, "rb").read()
        tmp = case.index + f".rebuild{os.getpid()}"
        try:
            L.build(recs, case.k, case.m, tmp, kind=case.kind, kmer_bits=case.kmer_bits, mm_seed=case.seed, c=case.c, threads=0)
            assert open(tmp, "rb").read() == want, "lphq_build writes different bytes than the reference build"
            ran.append("builder-bytes")
            orc.build(recs, case.k, case.m, tmp, kind=case.kind, kmer_bits=case.kmer_bits, mm_seed=case.seed, c=case.c)
            assert open(tmp, "rb").read() == want, "the oracle's builder writes different bytes than the reference build"
            ran.append("oracle-builder-bytes")
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return ran


def check_case_gpu(case, L, device=0):
    """The gfx950 kernels through the C ABI (lphq_query_contig: streaming and random) answer the reference's codes."""
    assert case.queries and case.codes, "a case without queries + codes has nothing to check on the GPU"
    cls = L.Mphf if case.kind == 0 else L.MphfAlt
    h = cls.load(case.index, kmer_bits=case.kmer_bits, device=device)
    recs = read_fastx(case.queries)
    ran = []
    for mode, want in case.codes.items():
        got = np.concatenate([h(r, streaming=(mode == "streaming")) for r in recs] or [np.zeros(0, dtype=np.uint64)])
        assert np.array_equal(got, want), f"GPU codes differ from the reference's ({mode})"
        ran.append(f"gpu-codes-{mode}")
    h.close()
    return ran


def write_oracle_case(d, orc, kind="p", k=31, m=20, kmer_bits=64, seed=42, c=3.0, n_records=6, rec_len=3000, rng_seed=5):
    """DRY RUN ONLY: a case directory in the layout of README.md whose index and codes come from the ORACLE (so it pins nothing); written
    under a temporary directory by the tests that exercise this module, never under tests/golden/ref."""
    from util import dedup_records, random_records

    os.makedirs(d, exist_ok=True)
    recs = dedup_records(random_records(rng_seed, n_records, rec_len // 2, rec_len), k)
    with gzip.open(os.path.join(d, "input.fa.gz"), "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">r%d some description\n" % i)
            for a in range(0, len(r), 70):  # multi-line FASTA, as the bundled files are
                f.write(r[a:a + 70] + b"\n")
    orc.build(recs, k, m, os.path.join(d, "index.lph"), kind=KIND[kind], kmer_bits=kmer_bits, mm_seed=seed, c=c)
    queries = recs[:2] + random_records(rng_seed + 1, 2, k + 3, 500) + [recs[0][: k - 1]]
    with gzip.open(os.path.join(d, "queries.fa.gz"), "wb") as f:
        for i, r in enumerate(queries):
            f.write(b">q%d\n%s\n" % (i, r))
    o = orc.Index(os.path.join(d, "index.lph"), KIND[kind], kmer_bits)
    for mode in ("streaming", "random"):
        np.concatenate([o(r, mode == "streaming") for r in queries]).astype("<u8").tofile(os.path.join(d, f"codes_{mode}.u64"))
    json.dump({"kind": kind, "kmer_bits": kmer_bits, "k": k, "m": m, "seed": seed, "c": c, "input": "input.fa.gz", "queries": "queries.fa.gz"},
              open(os.path.join(d, "meta.json"), "w"))
    return d
