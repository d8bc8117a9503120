"""CPU tests of the product's host side: C-ABI surface, loader + cross-checks, flat image, index builder
(byte-identical to the oracle's restatement of the reference build path), error behaviour.  No compute calls:
the query path only exists on the GPU."""
import ctypes as C
import gzip
import json
import os
import re
import subprocess

import numpy as np
import pytest

import lphash_amd as L
import orc
from lphash_amd import _lib
from util import dedup_records, random_records

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "lphq.h")).read()
    declared = set(re.findall(r"\b(lphq_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\b(lphq_[a-z_]+)\b", nm))
    assert declared <= exported, declared - exported
    # ... and nothing else: a drop-in C ABI exports its header (csrc/lphq.map), not kernel stubs or the C++ internals
    names = {ln.split()[-1].split("@")[0] for ln in nm.splitlines() if ln.strip()}
    assert names - {"LPHQ_3"} == declared, sorted(names ^ declared)[:20]


def test_no_torch_types_and_no_oracle_in_the_product():
    """The product must not link/import the oracle or expose torch types (the C ABI is plain pointers/sizes)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lphash_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", ".cuh")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "lph_oracle" not in src and "liblphoracle" not in src and "import orc" not in src, f
    header = open(os.path.join(ROOT, "include", "lphq.h")).read()
    assert "torch" not in header and "at::" not in header
    ldd = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "torch" not in ldd


def golden(name):
    return os.path.join(ROOT, "tests", "golden", name)


@pytest.mark.parametrize("name,cls,bits", [("p_k31_m20_b64", L.Mphf, 64), ("p_k63_m20_b128", L.Mphf, 128), ("u_k31_m20_b64", L.MphfAlt, 64),
                                           ("p_k21_m8_b64_fallback", L.Mphf, 64), ("u_k21_m8_b64_fallback", L.MphfAlt, 64)])
def test_loader_reads_golden_files(name, cls, bits):
    z = np.load(golden(name + ".npz"))
    h = cls.load(golden(name + ".lph"), kmer_bits=bits)
    assert (h.k, h.m, h.kmer_bits) == (int(z["k"]), int(z["m"]), bits)
    assert h.get_kmer_count() == int(z["nkmers"]) and h.get_minimizer_L0() == int(z["distinct_minimizers"])
    assert h.info.fallback_kmers == int(z["fallback_keys"])
    assert h.info.file_bytes == os.path.getsize(golden(name + ".lph")) and h.info.device == -1
    img = h.image()
    assert img.size == h.info.image_bytes and img.size % 256 == 0
    assert bytes(img[:8]) == b"LPQIMG01" and int(np.frombuffer(img[8:16].tobytes(), dtype=np.uint64)[0]) == img.size
    # same bytes through lphq_open_memory
    h2 = cls.from_bytes(open(golden(name + ".lph"), "rb").read(), kmer_bits=bits)
    assert np.array_equal(h2.image(), img)


def test_loader_errors(tmp_path):
    good = open(golden("p_k31_m20_b64.lph"), "rb").read()
    with pytest.raises(L.LphqError) as e:
        L.Mphf.load(str(tmp_path / "missing.lph"), kmer_bits=64)
    assert e.value.code == _lib.ERR_IO
    for bad, why in ((good[:-9], "truncated"), (good + b"\0" * 8, "trailing"), (good[:40], "header only")):
        with pytest.raises(L.LphqError) as e:
            L.Mphf.from_bytes(bad, kmer_bits=64)
        assert e.value.code == _lib.ERR_FORMAT, why
    with pytest.raises(L.LphqError) as e:  # a query-p file opened as query-u
        L.MphfAlt.from_bytes(good, kmer_bits=64)
    assert e.value.code == _lib.ERR_FORMAT
    with pytest.raises(L.LphqError) as e:
        L.Mphf.from_bytes(good, kmer_bits=32)
    assert e.value.code == _lib.ERR_ARG
    k63 = open(golden("p_k63_m20_b128.lph"), "rb").read()
    with pytest.raises(L.LphqError) as e:  # k=63 does not fit a 64-bit kmer_t (constants.hpp:16)
        L.Mphf.from_bytes(k63, kmer_bits=64)
    assert e.value.code == _lib.ERR_FORMAT
    # a flipped redundant field must be caught, not silently mis-hash
    corrupt = bytearray(good)
    corrupt[2 + 8 * 7 + 8 * 3] ^= 0x10  # low byte of m_M inside minimizer_order
    with pytest.raises(L.LphqError) as e:
        L.Mphf.from_bytes(bytes(corrupt), kmer_bits=64)
    assert e.value.code == _lib.ERR_FORMAT


def test_queries_fail_loudly_without_a_device():
    import torch

    h = L.Mphf.load(golden("p_k31_m20_b64.lph"), kmer_bits=64)
    with pytest.raises(L.LphqError) as e:  # not attached: no CPU fallback exists
        h(b"ACGT" * 20, streaming=False)
    assert e.value.code == _lib.ERR_STATE
    if not torch.cuda.is_available():
        with pytest.raises(L.LphqError) as e:
            h.attach(0)
        assert e.value.code == _lib.ERR_NO_DEVICE


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (31, 15, 128, 0), (63, 20, 128, 0), (31, 20, 64, 1), (15, 9, 64, 0), (31, 31, 64, 0),
                                           (21, 8, 64, 0), (21, 8, 64, 1), (47, 17, 128, 1), (33, 5, 128, 0)])
def test_builder_is_byte_identical_to_the_oracle_build(tmp_path, k, m, bits, kind):
    """Product builder (position-parallel, multi-threaded) vs the oracle's line-by-line restatement of
    from_string / classify / build_inverted_index / get_colliding_kmers: same bytes, for any thread count."""
    recs = random_records(1234 + k + m, 14, max(k - 3, 1), 3000)  # includes records shorter than k
    po, pp = str(tmp_path / "o.lph"), str(tmp_path / "p.lph")
    try:
        orc.build(recs, k, m, po, kind=kind, kmer_bits=bits)
    except RuntimeError:
        pytest.skip("duplicate k-mers in this random input")
    for threads in (1, 3, 8):
        L.build(recs, k, m, pp, kind=kind, kmer_bits=bits, threads=threads)
        assert open(po, "rb").read() == open(pp, "rb").read()


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (31, 20, 64, 1), (47, 30, 128, 0)])
def test_builder_parallel_pieces_are_byte_identical_to_the_oracle_build(tmp_path, k, m, bits, kind):
    """Enough minimizers (> 3 x 65,536) that the stages which work on pieces -- classification cut at minimizer boundaries, the type pass of the
    inverted index in 64-aligned pieces -- really run on several, against the oracle's sequential restatement."""
    recs = dedup_records(random_records(4321 + k, 20, 90_000, 140_000), k)
    po, pp = str(tmp_path / "o.lph"), str(tmp_path / "p.lph")
    orc.build(recs, k, m, po, kind=kind, kmer_bits=bits)
    for threads in (8, 3):
        st = L.build(recs, k, m, pp, kind=kind, kmer_bits=bits, threads=threads)
        assert st["distinct_minimizers"] > 3 * 65536
        assert open(po, "rb").read() == open(pp, "rb").read()


def test_builder_follows_the_reference_on_dirty_index_input(tmp_path):
    """Non-ACGT bytes in the INDEX set.  The reference restarts its window at every such byte (include/minimizer.hpp:139-152) -- and, because it looks for the
    minimum of a stretch's first window only when base k + 1 arrives (:61-66), a stretch of EXACTLY k valid bases that a non-ACGT byte closes is counted as a
    k-mer (:60) and pushed into no super-k-mer (the same stretch closed by the record's end is kept, :154-163).  The product's position-parallel builder -- in
    memory, with pieces of records on several threads, streamed in batches under a memory limit -- writes the oracle's file byte for byte on inputs made of such
    stretches (round 5: it used to index that k-mer, found by this differential run); the pure-Python model agrees."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import pymodel
    from util import dirty_index_records

    k, m = 31, 20
    a, b, c, d = random_records(4, 4, k, k)
    body = random_records(5, 3, 200, 400)
    recs = random_records(3, 6, 400, 900) + [a + b"N" + body[0], body[1] + b"N" + b + b"N" + body[2], random_records(8, 1, 77, 77)[0] + b"N" + c, d + b"NN"]
    want_lost = 3  # a, b and d; c ends with its record and stays
    for kind in (0, 1):
        po, pp = str(tmp_path / "o.lph"), str(tmp_path / "p.lph")
        orc.build(recs, k, m, po, kind=kind, kmer_bits=64)
        st = L.build(recs, k, m, pp, kind=kind, kmer_bits=64, threads=4)
        assert open(po, "rb").read() == open(pp, "rb").read()
        o = orc.Index(po, kind, 64)
        assert st["num_kmers"] == o.nkmers == sum(1 for r in recs for i in range(len(r) - k + 1) if b"N" not in r[i:i + k])  # the count includes them
        members = np.concatenate([o(x, False) for x in recs[:6] + [c]])
        assert len(np.unique(members)) == len(members) and int(members.max()) < o.nkmers - want_lost  # the function is over the k-mers it kept: n - 3 slots
        assert pymodel.Model(recs, k, m, kind, 64).serialize() == open(po, "rb").read()
    n = 0
    for seed in range(84):
        kk = [15, 21, 31, 47][seed % 4]
        mm = [7, 20, 13, 17, 5][seed % 5] if kk > 21 else [7, 11, 5][seed % 3]
        bits = 64 if kk <= 31 else 128
        recs = dirty_index_records(seed, kk, mm)
        if recs is None:
            continue
        po, pp, ps = str(tmp_path / "o.lph"), str(tmp_path / "p.lph"), str(tmp_path / "s.lph")
        try:
            orc.build(recs, kk, mm, po, kind=seed % 2, kmer_bits=bits)
        except RuntimeError:
            continue  # (a key set PTHash cannot place)
        L.build(recs, kk, mm, pp, kind=seed % 2, kmer_bits=bits, threads=1 + seed % 3)
        assert open(po, "rb").read() == open(pp, "rb").read(), (seed, kk, mm)
        if seed % 4 == 0:
            L.build_stream(lambda: iter([recs[:2], recs[2:]]), kk, mm, ps, kind=seed % 2, kmer_bits=bits, threads=2, tmp_dir=str(tmp_path / "spill"), max_memory_bytes=1 << 12)
            assert open(ps, "rb").read() == open(po, "rb").read(), (seed, kk, mm, "streamed")
        n += 1
    assert n > 50


def test_builder_argument_errors(tmp_path):
    recs = random_records(1, 2, 100, 200)
    out = str(tmp_path / "x.lph")
    for kw in (dict(k=32, m=20, kmer_bits=64), dict(k=31, m=32), dict(k=31, m=0), dict(k=31, m=20, c=2.0), dict(k=31, m=20, c=11.0),
               dict(k=64, m=20, kmer_bits=128)):
        kw.setdefault("kmer_bits", 64)
        with pytest.raises(L.LphqError) as e:
            L.build(recs, out_path=out, **kw)
        assert e.value.code == _lib.ERR_ARG


def test_cli_usage_and_build(tmp_path):
    """`lphash` CLI: usage/exit codes of src/lphash.cpp:21-35, build-p writes the same bytes as the library call."""
    import gzip

    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Available tools" in r.stderr and "query-p" in r.stderr
    r = subprocess.run([exe, "frobnicate"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unsupported tool 'frobnicate'." in r.stderr
    r = subprocess.run([exe, "query-p", "-i", "x.lph"], capture_output=True, text=True)  # -q missing -> ParseError -> 1
    assert r.returncode == 1
    recs = random_records(77, 9, 40, 3000)
    fa = tmp_path / "in.fa.gz"
    with gzip.open(fa, "wb") as f:
        for i, rec in enumerate(recs):
            f.write(b">rec%d some comment\n" % i)
            for j in range(0, len(rec), 70):  # multi-line FASTA, CRLF on some lines
                f.write(rec[j:j + 70] + (b"\r\n" if (i + j) % 3 == 0 else b"\n"))
    out = tmp_path / "cli.lph"
    r = subprocess.run([exe, "build-p", "-i", str(fa), "-k", "31", "-m", "20", "-o", str(out), "--kmer-bits", "64", "-t", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    fields = r.stdout.strip().split(",")
    assert fields[0] == str(fa) and fields[1:3] == ["31", "20"] and len(fields) == 8
    lib_out = tmp_path / "lib.lph"
    L.build(recs, 31, 20, str(lib_out), kmer_bits=64, threads=2)
    assert out.read_bytes() == lib_out.read_bytes()
    r = subprocess.run([exe, "build-p", "-i", str(fa), "-k", "33", "-m", "20", "--kmer-bits", "64"], capture_output=True, text=True)
    assert r.returncode == 3 and "k cannot be larger than 31" in r.stderr  # OptionError -> 3, src/build.cpp:17-20


def test_loader_fuzz_never_crashes():
    """Random corruptions of valid files must either load (and then describe a self-consistent function: the image builder
    re-derives every table) or be rejected with LPHQ_ERR_FORMAT/UNSUPPORTED -- never crash or hang the host side."""
    rng = np.random.default_rng(123)
    outcomes = {"ok": 0, "rejected": 0}
    for name, cls, bits in (("p_k31_m20_b64", L.Mphf, 64), ("u_k21_m8_b64_fallback", L.MphfAlt, 64), ("p_k63_m20_b128", L.Mphf, 128)):
        good = bytearray(open(golden(name + ".lph"), "rb").read())
        for trial in range(150):
            bad = bytearray(good)
            mode = trial % 3
            if mode == 0:  # flip a few bits
                for _ in range(int(rng.integers(1, 4))):
                    bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:  # overwrite an aligned u64 with a wild value
                off = int(rng.integers(0, len(bad) - 8))
                bad[off:off + 8] = int(rng.integers(0, 2**63)).to_bytes(8, "little")
            else:  # truncate or extend
                cut = int(rng.integers(1, len(bad)))
                bad = bad[:cut] if trial % 2 else bad + bytes(rng.integers(0, 256, 16, dtype=np.uint8))
            try:
                h = cls.from_bytes(bytes(bad), kmer_bits=bits)
                assert h.info.image_bytes > 0
                outcomes["ok"] += 1
            except L.LphqError as e:
                assert e.code in (_lib.ERR_FORMAT, _lib.ERR_UNSUPPORTED, _lib.ERR_ARG), e
                outcomes["rejected"] += 1
    assert outcomes["rejected"] > 100  # the cross-checks do catch most corruptions


def test_fastx_reader_record_semantics(tmp_path):
    """The CLI's reader must yield the records kseq_read yields (external/kseq.h:192-240): multi-line FASTA and FASTQ,
    '>'/'@' headers with comments, blank lines, CRLF, quality lines starting with '@', a last line without newline, gz.
    Observed through `lphash build-p`: its file must equal the library build over the hand-written expected records."""
    import gzip

    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    r = random_records(4242, 12, 80, 400)
    fa = b"garbage before the first header\n>r0 comment here\n" + r[0][:50] + b"\n" + r[0][50:] + b"\n\n>r1\r\n" + r[1] + b"\r\n>empty\n>r2\tx\n"
    fa += r[2][:10] + b"\n" + r[2][10:300].lower() + b"\n" + r[2][300:] + b"\n>r3\n" + r[3]  # no trailing newline
    fq = b"@q4 desc\n" + r[4] + b"\n+\n" + b"@" * len(r[4]) + b"\n@q5\n" + r[5][:40] + b"\n" + r[5][40:] + b"\n+q5\n" + b"I" * 40 + b"\n" + b"@" + b"I" * (len(r[5]) - 41)
    fq += b"\n@q6\n" + r[6] + b"\n+\n" + b"#" * len(r[6]) + b"\n"
    cases = [("a.fa", fa, [r[0], r[1], b"", r[2][:10] + r[2][10:300].lower() + r[2][300:], r[3]]), ("b.fq", fq, [r[4], r[5], r[6]]),
             ("c.fa.gz", gzip.compress(b">x\n" + r[7] + b"\n>y\n" + r[8] + b"\n"), [r[7], r[8]]),
             ("d.fq", b"@t\n" + r[9] + b"\n+\n" + b"I" * (len(r[9]) - 3) + b"\n@u\n" + r[10] + b"\n+\n" + b"I" * len(r[10]) + b"\n", [])]  # truncated quality: kseq stops (-2)
    for name, data, expected in cases:
        p = tmp_path / name
        p.write_bytes(data)
        out = tmp_path / (name + ".lph")
        res = subprocess.run([exe, "build-p", "-i", str(p), "-k", "31", "-m", "20", "-o", str(out), "--kmer-bits", "64", "--cpu"], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        ref = tmp_path / (name + ".ref.lph")
        L.build(expected, 31, 20, str(ref), kmer_bits=64, threads=2)
        assert out.read_bytes() == ref.read_bytes(), name


def test_parallel_fastx_reader_equals_serial(tmp_path):
    """`lphash records`: the multi-threaded batch reader (fastx_batch.hpp) yields exactly the serial kseq-semantics reader's records
    -- same count, bases and checksum -- for any batch size, thread count and piece size; FASTQ-looking input makes it fall back."""
    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    rng = np.random.default_rng(77)

    def fasta(n, lmin, lmax, width, crlf=False, blank=False, lower=False, headers=True):
        out = bytearray()
        for i in range(n):
            out += b">r%d some comment\tx" % i if headers else b">"
            out += b"\r\n" if crlf else b"\n"
            s = bytes(rng.choice(np.frombuffer(b"ACGTacgtNn" if lower else b"ACGT", dtype=np.uint8), int(rng.integers(lmin, lmax + 1))))
            for j in range(0, len(s), width):
                out += s[j:j + width] + (b"\r\n" if crlf else b"\n")
                if blank and rng.integers(0, 5) == 0:
                    out += b"\n"
        return bytes(out)

    r = random_records(5, 8, 30, 300)
    files = {
        "plain.fa": fasta(300, 0, 900, 60),
        "crlf.fa": fasta(120, 0, 500, 70, crlf=True),
        "blank_lower.fa": fasta(200, 1, 400, 33, blank=True, lower=True),
        "tiny_records.fa": fasta(2000, 0, 3, 80),
        "one_line_long.fa": fasta(3, 200000, 300000, 1 << 30),
        "long_multiline.fa": fasta(2, 150000, 200000, 60),           # records far longer than a piece: pieces begin inside records
        "long_multiline_crlf.fa": fasta(2, 50000, 80000, 70, crlf=True),
        "no_trailing_newline.fa": fasta(50, 10, 200, 60)[:-1],
        "header_only_at_end.fa": fasta(20, 10, 200, 60) + b">last",
        "header_with_space_at_end.fa": fasta(20, 10, 200, 60) + b">last one",
        "at_records.fa": b">a\n" + r[0] + b"\n@b\n" + r[1] + b"\n>c\n" + r[2] + b"\n",          # '@' starts a record in kseq
        "gt_inside_lines.fa": b">a x>y\n" + r[0][:20] + b">" + r[0][20:] + b"\n>b\n" + r[1] + b"\n",  # '>' not at a line start
        "cr_only_line.fa": b">a\n\r\n" + r[0] + b"\n>b\n" + r[1] + b"\r\n\r\n>c\n",
        "fastq_after_fasta.fq": fasta(40, 50, 300, 60) + b"@q\n" + r[3] + b"\n+\n" + b">" * len(r[3]) + b"\n>tail\n" + r[4] + b"\n",  # '+' line: must fall back
        "plus_first.fa": b">a\n+\n" + r[0] + b"\n",
        "starts_with_garbage.fa": b"junk\n" + fasta(10, 10, 100, 60),
        "empty.fa": b"",
        "only_gt.fa": b">",
    }
    for name, data in files.items():
        p = tmp_path / name
        p.write_bytes(data)
        ser = subprocess.run([exe, "records", "-i", str(p), "--serial"], capture_output=True, text=True)
        assert ser.returncode == 0, (name, ser.stderr)
        want = ser.stdout.strip().split(",")[:3]
        for threads, batch, piece in ((1, 1 << 30, 1 << 16), (4, 1 << 30, 1), (8, 997, 1), (3, 50000, 64), (16, 1, 1)):
            par = subprocess.run([exe, "records", "-i", str(p), "-t", str(threads), "--batch-bytes", str(batch), "--min-piece", str(piece)],
                                 capture_output=True, text=True)
            assert par.returncode == 0, (name, par.stderr)
            got = par.stdout.strip().split(",")
            assert got[:3] == want, (name, threads, batch, piece, got, want)
            if name in ("plain.fa", "tiny_records.fa", "crlf.fa") and data:
                assert int(got[4]) == int(got[3]) > 0  # every batch came from the multi-threaded path
            if name in ("plus_first.fa", "starts_with_garbage.fa"):
                assert int(got[4]) == 0                # ... and none here


def test_parallel_fastx_reader_fuzz(tmp_path):
    """Random byte soups over the characters that matter to the record state machine ('>', '@', '+', line ends, blanks): whatever the
    serial reader makes of them, the multi-threaded reader must make the same (it either parses pieces or falls back)."""
    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    rng = np.random.default_rng(2024)
    alphabet = np.frombuffer(b"ACGTacgtN\n\n\n\r>@+ \t", dtype=np.uint8)
    weights = {
        "mostly_fasta": [8, 8, 8, 8, 1, 1, 1, 1, 1, 2, 2, 2, .2, .4, .02, .0, .1, .05],
        "few_plus": [8, 8, 8, 8, 1, 1, 1, 1, 1, 2, 2, 2, .2, .4, .05, .01, .1, .05],
        "hostile": [2, 2, 2, 2, 1, 1, 1, 1, 1, 3, 3, 3, 1, 2, 1, 1, 1, 1],
    }
    n_parallel_batches = 0
    for name, w in weights.items():
        pr = np.array(w, dtype=np.float64)
        pr /= pr.sum()
        for trial in range(12):
            body = bytes(rng.choice(alphabet, size=int(rng.integers(1, 6000)), p=pr))
            data = (b">" if trial % 4 else b"") + body
            p = tmp_path / f"{name}_{trial}.fa"
            p.write_bytes(data)
            ser = subprocess.run([exe, "records", "-i", str(p), "--serial"], capture_output=True, text=True)
            assert ser.returncode == 0, (name, trial, ser.stderr)
            want = ser.stdout.strip().split(",")[:3]
            for threads, batch, piece in ((4, 1 << 30, 1), (8, 257, 1), (3, 1500, 16)):
                par = subprocess.run([exe, "records", "-i", str(p), "-t", str(threads), "--batch-bytes", str(batch), "--min-piece", str(piece)],
                                     capture_output=True, text=True)
                assert par.returncode == 0, (name, trial, par.stderr)
                got = par.stdout.strip().split(",")
                assert got[:3] == want, (name, trial, threads, batch, piece, got, want, data[:200])
                n_parallel_batches += int(got[4])
    assert n_parallel_batches > 50  # the multi-threaded path did take part


def test_cli_build_from_long_records_with_threads(tmp_path):
    """`lphash build-p -t 8` on records much longer than a parse piece: the reader's pieces begin inside records, the builder needs
    them contiguous again (FastxBatch::compact) -- the file must equal the library build over the same records."""
    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    recs = random_records(31337, 3, 250_000, 400_000)
    fa = tmp_path / "long.fa"
    with open(fa, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">contig%d\n" % i)
            for j in range(0, len(r), 70):
                f.write(r[j:j + 70] + b"\n")
    out, ref = tmp_path / "cli.lph", tmp_path / "ref.lph"
    res = subprocess.run([exe, "build-p", "-i", str(fa), "-k", "31", "-m", "20", "-o", str(out), "--kmer-bits", "64", "--cpu", "-t", "8"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    L.build(recs, 31, 20, str(ref), kmer_bits=64, threads=4)
    assert out.read_bytes() == ref.read_bytes()


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (31, 15, 64, 1), (63, 20, 128, 0), (21, 8, 64, 0), (33, 7, 128, 1)])
def test_streamed_build_with_a_memory_limit_writes_the_same_file(tmp_path, k, m, bits, kind):
    """lphq_build_stream (the reference's two passes over the input + external_memory_vector, src/partitioned_mphf.cpp:58-130): whatever the batch
    boundaries and the memory limit -- no spill, a few sorted pieces, more than 64 pieces (merged on the way) -- the bytes are those of the in-memory
    build (itself byte-identical to the oracle's builder), the statistics too, and the temporary files are gone.  With a limit every per-minimizer
    vector lives in files too (distinct minimizers, their order under the MPHF, the super-k-mers of repeated minimizers -- most of them at m = 7, 8)
    and both PTHash functions are built from sorted pieces of (bucket, hash) pairs, searched a piece at a time."""
    # (m = 7, 8: millions of k-mers go to the fallback function and its pilot search on one host thread is seconds per build -- fewer builds there, a smaller input at m = 7)
    recs = dedup_records(random_records(99 + k, 24 if m == 7 else 60, max(k - 2, 1), 110_000), k)
    recs = recs[:7] + [b"", b"ACGT"] + recs[7:]  # empty and too-short records inside a batch and at batch boundaries
    want_path, got_path, spill = str(tmp_path / "mem.lph"), str(tmp_path / "str.lph"), tmp_path / "spill"
    want_stats = L.build(recs, k, m, want_path, kind=kind, kmer_bits=bits, threads=4)
    want = open(want_path, "rb").read()
    settings = ((len(recs), 0), (5, 0), (1, 1 << 20), (9, 1 << 22), (len(recs), 1 << 18))
    for per_batch, limit in (settings if m > 8 else settings[3:] if m == 8 else settings[2:3]):
        calls = []

        def batches():
            calls.append(1)
            for i in range(0, len(recs), per_batch):
                yield recs[i:i + per_batch]

        stats = L.build_stream(batches, k, m, got_path, kind=kind, kmer_bits=bits, threads=4, tmp_dir=str(spill), max_memory_bytes=limit)
        assert open(got_path, "rb").read() == want, (per_batch, limit)
        assert stats == want_stats
        assert len(calls) == (2 if stats["colliding_super_kmers"] else 1)  # the second pass collects the k-mers of repeated minimizers, if any
        assert not spill.exists() or os.listdir(spill) == []
    # > 64 pieces at the smallest limit: 32-byte records, pieces of max(4096, limit / 4 / 32) records
    if (k, m) == (31, 20):
        assert want_stats["num_super_kmers"] > 64 * max(4096, (1 << 18) // 4 // 32)
    if (k, m) in ((31, 20), (33, 7)):  # buckets of "unheard-of" size share one file and are ordered in memory: the threshold lowered to 3 keys puts most buckets there
        os.environ["LPHQ_EXT_BIG_BUCKET"] = "3"
        try:
            stats = L.build_stream(lambda: iter([recs]), k, m, got_path, kind=kind, kmer_bits=bits, threads=4, tmp_dir=str(spill), max_memory_bytes=1 << 19)
        finally:
            del os.environ["LPHQ_EXT_BIG_BUCKET"]
        assert open(got_path, "rb").read() == want and stats == want_stats and os.listdir(spill) == []
    if m == 8:  # repeated minimizers: more of their super-k-mers than 64 pieces of 4096, more fallback k-mers than 64 pieces of (bucket, hash) pairs
        assert want_stats["colliding_super_kmers"] > 64 * 4096 and want_stats["fallback_kmers"] > 64 * 4096


def test_concurrent_builds_share_a_temporary_directory(tmp_path):
    """Several builds with a memory limit at once in one process and one temporary directory (the names of the temporary files carry a process-wide serial):
    each writes the file of the in-memory build, and the directory is empty afterwards."""
    import threading

    jobs = [(31, 20, 64, 0, 11), (21, 8, 64, 0, 12), (31, 15, 64, 1, 13), (31, 20, 64, 0, 14)]
    inputs = {j: dedup_records(random_records(j[4], 25, j[0], 60_000), j[0]) for j in jobs}
    want = {}
    for j in jobs:
        k, m, bits, kind, _ = j
        L.build(inputs[j], k, m, str(tmp_path / "want.lph"), kind=kind, kmer_bits=bits, threads=2)
        want[j] = (tmp_path / "want.lph").read_bytes()
    spill, errors = tmp_path / "shared", []

    def work(j, i):
        k, m, bits, kind, _ = j
        try:
            out = str(tmp_path / f"got{i}.lph")
            L.build_stream(lambda: iter([inputs[j][:9], inputs[j][9:]]), k, m, out, kind=kind, kmer_bits=bits, threads=2, tmp_dir=str(spill), max_memory_bytes=1 << 16)
            if open(out, "rb").read() != want[j]:
                errors.append((j, "different file"))
        except Exception as e:  # noqa: BLE001
            errors.append((j, repr(e)))

    threads = [threading.Thread(target=work, args=(j, i)) for i, j in enumerate(jobs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == [] and os.listdir(spill) == []


def test_build_with_a_memory_limit_when_the_disk_refuses_writes(tmp_path):
    """A full disk in the middle of a construction in external memory (RLIMIT_FSIZE in a child process: every write past the limit fails): whichever
    temporary file is hit first -- sorted pieces of super-k-mer records, the file of distinct minimizers, the (bucket, hash) pairs, the files by
    bucket size, the order pieces -- the CLI reports the file and exits non-zero, nothing is left in the temporary directory; with room enough the
    file is the in-memory one."""
    import resource
    import signal

    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    recs = dedup_records(random_records(777, 30, 50_000, 120_000), 31)
    fa = tmp_path / "in.fa"
    with open(fa, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">c%d\n" % i + r + b"\n")
    want = tmp_path / "want.lph"
    r0 = subprocess.run([exe, "build-p", "-i", str(fa), "-k", "31", "-m", "20", "-o", str(want), "--cpu", "-t", "4"], capture_output=True, text=True)
    assert r0.returncode == 0, r0.stderr
    failed = set()
    for limit_kb in (64, 200, 700, 1500, 3000, 6000, 1 << 20):
        def pre():
            signal.signal(signal.SIGXFSZ, signal.SIG_IGN)
            resource.setrlimit(resource.RLIMIT_FSIZE, (limit_kb << 10, limit_kb << 10))

        d, out = tmp_path / f"ext{limit_kb}", tmp_path / f"got{limit_kb}.lph"
        r = subprocess.run([exe, "build-p", "-i", str(fa), "-k", "31", "-m", "20", "-o", str(out), "--cpu", "-t", "4", "-d", str(d), "--max-memory-mb", "2"],
                           capture_output=True, text=True, preexec_fn=pre)
        assert os.listdir(d) == [], (limit_kb, os.listdir(d))
        if r.returncode == 0:
            assert out.read_bytes() == want.read_bytes(), limit_kb
        else:
            assert "temporary file" in r.stderr or "short write" in r.stderr or "cannot write" in r.stderr, (limit_kb, r.stderr[-500:])
            failed.add(r.stderr.strip().rsplit(".", 1)[-1] if "temporary file" in r.stderr else "output")
    assert len(failed) >= 2 and r.returncode == 0, failed  # more than one kind of file was the first to be refused; the last run had room


def test_every_environment_hook_is_documented():
    """Every LPHQ_* variable the product reads (getenv in lphash_amd/csrc, os.environ in the Python mirror) has its line in INTEGRATION.md section 5."""
    import re

    names = set()
    csrc = os.path.join(ROOT, "lphash_amd", "csrc")
    for root, _, files in os.walk(csrc):
        for f in files:
            if f.endswith((".cpp", ".hpp", ".hip", ".cuh", ".h")):
                names |= set(re.findall(r'getenv\("(LPHQ_[A-Z0-9_]+)"\)', open(os.path.join(root, f), errors="replace").read()))
    for f in os.listdir(os.path.join(ROOT, "lphash_amd")):
        if f.endswith(".py"):
            names |= set(re.findall(r'environ(?:\.get)?[\[(]"(LPHQ_[A-Z0-9_]+)"', open(os.path.join(ROOT, "lphash_amd", f)).read()))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert len(names) >= 15 and [n for n in sorted(names) if f"`{n}`" not in doc] == []


def test_streamed_build_errors(tmp_path):
    recs = dedup_records(random_records(5, 12, 200, 4000), 31)
    out = str(tmp_path / "x.lph")

    def failing():
        yield recs[:3]
        raise IOError("disk on fire")

    with pytest.raises(IOError):
        L.build_stream(failing, 31, 20, out, kmer_bits=64)
    # a source that does not replay the same input: the second pass cannot find the windows of the first
    repeated = [recs[0], recs[0][:500]] + recs[1:]  # a repeated stretch -> repeated minimizers -> the second pass has work to do
    state2 = {"n": 0}

    def repeated_then_short():
        state2["n"] += 1
        yield repeated if state2["n"] == 1 else [r[:40] for r in repeated]

    with pytest.raises(L.LphqError) as e:
        L.build_stream(repeated_then_short, 31, 20, out, kmer_bits=64)
    assert e.value.code == _lib.ERR_IO
    stats = _lib.BuildStats()
    rc = _lib.lib.lphq_build_stream(None, None, 31, 20, 42, 3.0, 0, 64, 1, -1, None, 0, out.encode(), C.byref(stats))  # no source
    assert rc == _lib.ERR_ARG
    with pytest.raises(L.LphqError) as e:  # argument checks come before the first batch is asked for
        L.build_stream(lambda: iter([recs]), 33, 20, out, kmer_bits=64)
    assert e.value.code == _lib.ERR_ARG
    with pytest.raises(L.LphqError) as e:  # an unwritable temporary directory
        big = dedup_records(random_records(6, 8, 20_000, 40_000), 31)  # more super-k-mers than the smallest piece (4096 records)
        L.build_stream(lambda: iter([big]), 31, 20, out, kmer_bits=64, tmp_dir="/proc/no/such/dir", max_memory_bytes=1 << 12)
    assert e.value.code == _lib.ERR_IO and "temporary file" in str(e.value)


def test_cli_build_with_max_memory(tmp_path):
    """`lphash build-p|build-u -d tmp --max-memory-mb N` (parser_build.cpp:74-88: -d is created, the limit switches to construction in external
    memory): same file and same CSV line as without a limit; --check (one more pass over the file, batch by batch) passes where a GPU is present."""
    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    recs = dedup_records(random_records(4242, 40, 20, 100_000), 31)
    fa = tmp_path / "in.fa"
    with open(fa, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">c%d\n" % i)
            for j in range(0, len(r), 61):
                f.write(r[j:j + 61] + b"\n")
    for sub in ("build-p", "build-u"):
        a, b, d = tmp_path / "a.lph", tmp_path / "b.lph", tmp_path / "ext"
        base = [exe, sub, "-i", str(fa), "-k", "31", "-m", "20", "--kmer-bits", "64", "--cpu", "-t", "4"]
        r1 = subprocess.run(base + ["-o", str(a)], capture_output=True, text=True)
        r2 = subprocess.run(base + ["-o", str(b), "-d", str(d), "--max-memory-mb", "1", "--verbose"], capture_output=True, text=True)
        assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr, r2.stderr)
        assert a.read_bytes() == b.read_bytes() and r1.stdout == r2.stdout
        assert "Memory limit 1 MB" in r2.stderr and d.is_dir() and os.listdir(d) == []
    r = subprocess.run(base + ["--max-memory", "256"], capture_output=True, text=True)
    assert r.returncode == 3 and "The maximum allowed amount of ram is 255GB" in r.stderr


KSEQ_DUMP = os.path.join(ROOT, "oracle", "_ref", "kseq_dump")


@pytest.mark.skipif(not os.path.exists(KSEQ_DUMP), reason="oracle/_ref/kseq_dump is built only where /root/reference exists (oracle/Makefile)")
def test_readers_against_the_real_kseq(tmp_path):
    """Pin: the reference's OWN parser (external/kseq.h, compiled from /root/reference by oracle/Makefile into oracle/_ref/kseq_dump and
    driven exactly like src/query.cpp:43-56) against the product's serial and multi-threaded readers, on crafted corner cases and on
    byte soups: record count, base count and checksum of every record must agree."""
    import gzip

    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    rng = np.random.default_rng(31)
    r = random_records(6, 8, 30, 300)
    crafted = [
        b">a\n" + r[0] + b"\n>b", b">a\n" + r[0] + b"\n>b x", b">a\n" + r[0] + b"\n>", b">", b">x", b"", b"\n\n", b"junk>a\nAC\n",
        b">a\nAC\r\n\r\nGT\r", b">a\n\r\nAC\n", b">a\nAC\n\r", b">a\nA\r", b">a\r\nAC\r\r\n\nG\n",
        b"@q\n" + r[1] + b"\n+\n" + b"I" * len(r[1]) + b"\n@r\n" + r[2] + b"\n+\n" + b"I" * (len(r[2]) - 1) + b"\n",
        b"@q\n\n+\nIIII\n@r\nAC\n+\nII\n", b"@q\nAC\n+", b"@q\nAC\n+\n", b"@q\nAC\n+\nI\r\nI\n>x\nGG\n", b"@q\nACGT\n+\nII\r\r\n\n>x\nGG\n",
        b">a b\tc\nAC>GT\n+notqual\nAC\n", b">a\n@b\nAC\n@c\n", gzip.compress(b">z\n" + r[3] + b"\n>y\n" + r[4][:50] + b"\n" + r[4][50:] + b"\n"),
    ]
    alphabet = np.frombuffer(b"ACGTacgtN\n\n\n\r>@+ \t", dtype=np.uint8)
    weights = [[8, 8, 8, 8, 1, 1, 1, 1, 1, 2, 2, 2, .2, .4, .02, .0, .1, .05], [2, 2, 2, 2, 1, 1, 1, 1, 1, 3, 3, 3, 1, 2, 1, 1, 1, 1],
               [4, 4, 4, 4, 0, 0, 0, 0, 0, 2, 2, 2, .5, .3, .3, .3, .2, .2], [8, 8, 8, 8, 0, 0, 0, 0, 0, 1, 1, 1, 1.5, .05, 0, 0, 0, 0]]
    soups = []
    for t in range(120):
        pr = np.array(weights[t % 4], dtype=np.float64)
        pr /= pr.sum()
        body = bytes(rng.choice(alphabet, size=int(rng.integers(0, 2500)), p=pr))
        soups.append((b">" if t % 3 else (b"@" if t % 2 else b"")) + body)
    for i, data in enumerate(crafted + soups):
        p = tmp_path / f"f{i}.fx"
        p.write_bytes(data)
        want = subprocess.run([KSEQ_DUMP, str(p)], capture_output=True, text=True).stdout.strip()
        ser = subprocess.run([exe, "records", "-i", str(p), "--serial"], capture_output=True, text=True).stdout.strip().split(",")[:3]
        par = subprocess.run([exe, "records", "-i", str(p), "-t", "4", "--min-piece", "1", "--batch-bytes", str(1 + 37 * i)], capture_output=True,
                             text=True).stdout.strip().split(",")[:3]
        assert ",".join(str(v) for v in orc.file_digest(str(p))) == want, (i, "oracle driver reader", want, data[:120])
        assert ",".join(ser) == want, (i, "serial", ser, want, data[:120])
        assert ",".join(par) == want, (i, "parallel", par, want, data[:120])


@pytest.mark.skipif(not os.path.exists(KSEQ_DUMP), reason="oracle/_ref/kseq_dump is built only where /root/reference exists (oracle/Makefile)")
def test_parallel_fastq_parser_against_the_real_kseq(tmp_path):
    """Plain FASTQ goes through the multi-threaded parser when it is the four-line kind; the records must be kseq's in every case: quality lines that
    begin with '@' or '+', a last record without a line end, and -- via the fall-back to the serial reader -- multi-line records, CRLF, truncated or
    over-long quality strings, empty sequences, junk between records."""
    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    rng = np.random.default_rng(99)
    qual_alphabet = np.frombuffer(b"@+IIIIFFFF#5:<>", dtype=np.uint8)

    def record(i, n, style):
        seq = random_records(int(rng.integers(1 << 30)), 1, n, n)[0]
        qual = bytes(rng.choice(qual_alphabet, size=n))
        if style == "at":
            qual = b"@" + qual[1:]
        if style == "plus":
            qual = b"+" + qual[1:]
        if style == "short":
            qual = qual[:-1]
        if style == "long":
            qual = qual + b"I"
        if style == "multiline" and n > 10:
            return b"@r%d x\n%s\n%s\n+\n%s\n%s\n" % (i, seq[:7], seq[7:], qual[:5], qual[5:])
        if style == "crlf":
            return b"@r%d\r\n%s\r\n+\r\n%s\r\n" % (i, seq, qual)
        if style == "junk":
            return b"@r%d\n%s\n+r%d again\n%s\nstray line\n" % (i, seq, i, qual)
        if style == "empty":
            return b"@r%d\n\n+\n\n" % i
        return b"@r%d some comment\n%s\n+\n%s\n" % (i, seq, qual)

    n_parallel = 0
    for trial in range(60):
        styles = ["ok", "ok", "ok", "at", "plus"] if trial < 30 else ["ok", "at", "plus", "short", "long", "multiline", "crlf", "junk", "empty"]
        recs = [record(i, int(rng.integers(1, 200)), styles[int(rng.integers(len(styles)))]) for i in range(int(rng.integers(1, 120)))]
        data = b"".join(recs)
        if trial % 4 == 0:
            data = data.rstrip(b"\n")  # last quality line without a line end
        p = tmp_path / f"t{trial}.fq"
        p.write_bytes(data)
        want = subprocess.run([KSEQ_DUMP, str(p)], capture_output=True, text=True).stdout.strip()
        for threads, piece, batch in ((4, 1, 700), (7, 1, 100000), (3, 64, 5000)):
            par = subprocess.run([exe, "records", "-i", str(p), "-t", str(threads), "--min-piece", str(piece), "--batch-bytes", str(batch)], capture_output=True,
                                 text=True).stdout.strip().split(",")
            assert ",".join(par[:3]) == want, (trial, threads, piece, batch, par, want, data[:300])
            n_parallel += int(par[4])
    assert n_parallel > 100  # the multi-threaded path did take part


def test_host_sources_under_sanitizers(tmp_path):
    """The GPU-free host sources of the product (loader, image builder, index builder with the CPU pilot search, FASTA readers) compiled
    with -fsanitize=address,undefined (tests/san/san_host.cpp): builds with 1 and 8 threads, loads, and a few hundred corrupted files
    through the loader must raise no report (a report aborts the harness)."""
    import shutil

    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    csrc = os.path.join(ROOT, "lphash_amd", "csrc")
    exe = str(tmp_path / "san_host")
    texe = str(tmp_path / "tsan_host")  # (the ThreadSanitizer build of the last part compiles meanwhile)
    tsan_cc = subprocess.Popen([gxx, "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I" + csrc, os.path.join(ROOT, "tests", "san", "san_host.cpp"),
                                os.path.join(csrc, "builder.cpp"), os.path.join(csrc, "lph_format.cpp"), os.path.join(csrc, "image.cpp"), "-o", texe, "-lz", "-lpthread"],
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cc = subprocess.run([gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + csrc,
                         os.path.join(ROOT, "tests", "san", "san_host.cpp"), os.path.join(csrc, "builder.cpp"), os.path.join(csrc, "lph_format.cpp"),
                         os.path.join(csrc, "image.cpp"), "-o", exe, "-lz", "-lpthread"], capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr:
        tsan_cc.wait()
        pytest.skip("this g++ has no sanitizer runtime")
    assert cc.returncode == 0, cc.stderr[-2000:]
    recs = random_records(808, 30, 300, 6000) + random_records(809, 2, 90_000, 120_000)
    fa = tmp_path / "in.fa"
    with open(fa, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">r%d\n" % i)
            for j in range(0, len(r), 70):
                f.write(r[j:j + 70] + b"\n")
    env = dict(os.environ, UBSAN_OPTIONS="print_stacktrace=1", ASAN_OPTIONS="detect_leaks=1")
    for k, m, kind, bits in ((31, 20, 0, 64), (63, 20, 0, 128), (31, 20, 1, 64), (21, 8, 0, 64)):  # (m = 8: most minimizers repeat; one in-memory build is enough there)
        out = tmp_path / f"o_{k}_{kind}.lph"
        for threads in ((1, 8) if m > 8 else (8,)):
            r = subprocess.run([exe, "build", str(fa), str(k), str(m), str(kind), str(bits), str(threads), str(out)], capture_output=True, text=True, env=env)
            assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
        ref = tmp_path / "ref.lph"
        L.build(recs, k, m, str(ref), kind=kind, kmer_bits=bits, threads=2)
        assert out.read_bytes() == ref.read_bytes()
        # the streamed builder: small batches, every vector of the construction in sorted pieces, spilled and merged; both PTHash searches in pieces
        r = subprocess.run([exe, "build_stream", str(fa), str(k), str(m), str(kind), str(bits), "4", str(out), str(1 << 12), str(tmp_path / "spill")],
                           capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
        assert out.read_bytes() == ref.read_bytes()
        if k == 31 and kind == 0:  # the same through a BGZF copy of the input: the block-parallel inflate into a sliding window of a few (tiny) batches
            bg = tmp_path / "in.bgzf.fa.gz"
            bg.write_bytes(bgzf_compress(fa.read_bytes(), block=5000))
            r = subprocess.run([exe, "build_stream", str(bg), str(k), str(m), str(kind), str(bits), "4", str(out), str(1 << 12), str(tmp_path / "spill")],
                               capture_output=True, text=True, env=env)
            assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
            assert out.read_bytes() == ref.read_bytes()
        for cmd in ((["load", str(out), str(kind), str(bits)], ["fuzz", str(out), str(kind), str(bits), "150", str(k)]) if m > 8 else ()):
            r = subprocess.run([exe] + cmd, capture_output=True, text=True, env=env)
            assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
        if m > 8:  # the device layout's 32-bit limits, each forged into the parsed file: refused with LPHQ_ERR_UNSUPPORTED's text, never mis-hashed
            r = subprocess.run([exe, "limits", str(out), str(kind), str(bits)], capture_output=True, text=True, env=env)
            assert r.returncode == 0 and r.stdout.count(": refused") == 7 and "NOT REFUSED" not in r.stdout, (r.stdout, r.stderr[-2000:])

    # round 6: the host side of the run-encoded result (runs_host.hpp): random runs expanded by the pool == a serial expansion, nothing written outside them
    for n_w, seed, threads in ((200_000, 1, 8), (1, 2, 3), (5_000, 3, 1), (70_000, 4, 16)):
        r = subprocess.run([exe, "runs", str(n_w), str(seed), str(threads)], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "runs ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, (n_w, r.returncode, r.stderr[-3000:])
    # the .gz decoder of the serial input path: a file through the threaded source (odd read sizes), then corrupted copies through the decoder
    import gzip

    gz = tmp_path / "in.fa.gz"
    with gzip.open(gz, "wb", compresslevel=6) as f:
        f.write(fa.read_bytes() * 12)  # several 4-MiB chunks
    r = subprocess.run([exe, "inflate", str(gz)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.split(",")[0] == str(12 * fa.stat().st_size) and r.stdout.strip().endswith(",ok"), (r.stdout, r.stderr[-2000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    small = tmp_path / "small.fa.gz"
    with gzip.open(small, "wb", compresslevel=9) as f:
        f.write(fa.read_bytes()[:300_000])
    r = subprocess.run([exe, "inflate_fuzz", str(small), "400", "7"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    # the same stream through the span-parallel decoder (parallel_gz.hpp): same bytes (FNV of the text), then corrupted copies -- no report, no hang
    want = subprocess.run([exe, "inflate", str(gz)], capture_output=True, text=True, env=env).stdout.split(",")[:2]
    for threads, span in ((4, 40_000), (7, 9_000), (2, 0)):
        r = subprocess.run([exe, "inflate_par", str(gz), str(threads), str(span)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().split(",") == want + ["ok"], (threads, span, r.stdout, r.stderr[-3000:])
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    r = subprocess.run([exe, "inflate_par_fuzz", str(small), "120", "11", "4", "12000", str(tmp_path / "bad.gz")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    # ... and under ThreadSanitizer: the pool, the chain and the emits of the parallel decoder race-free
    if tsan_cc.wait() == 0:  # (a g++ without the TSan runtime: the ASan run above stands alone)
        r = subprocess.run([texe, "runs", "300000", "9", "8"], capture_output=True, text=True, timeout=900)
        assert "ThreadSanitizer" not in r.stderr and r.returncode == 0, r.stderr[-4000:]
        for threads, span in ((4, 40_000), (6, 9_000)):
            r = subprocess.run([texe, "inflate_par", str(gz), str(threads), str(span)], capture_output=True, text=True, timeout=900)
            assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
            assert r.returncode == 0 and r.stdout.strip().split(",") == want + ["ok"], (threads, span, r.stdout, r.stderr[-3000:])


def test_ptbb_build_cli_and_file(tmp_path):
    """`lphash ptbb-build` (the reference's ptbb_build, PTHash half, src/ptbb/ptbb_build.cpp): CSV fields, the file equals the oracle's
    restatement byte for byte, loader round trip; -b (BBHash) is refused; unreadable input -> exit code 2."""
    import gzip

    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    recs = random_records(4242, 9, 40, 3000)
    fa = tmp_path / "in.fa.gz"
    with gzip.open(fa, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">u%d\n%s\n" % (i, r))
    n = sum(len(r) - 30 for r in recs)
    for bits in (64, 128):
        out, ref = tmp_path / f"cli{bits}.pthash", tmp_path / f"orc{bits}.pthash"
        r = subprocess.run([exe, "ptbb-build", "-i", str(fa), "-k", "31", "-p", str(out), "--kmer-bits", str(bits), "--cpu", "-t", "4"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        f = r.stdout.strip().split(",")
        assert len(f) == 7 and f[0] == str(fa) and f[1] == "31" and int(f[2]) == n and int(f[3]) == os.path.getsize(out) * 8 and f[5] == f[6] == ""
        orc.ptbb_build(recs, 31, str(ref), kmer_bits=bits)
        assert out.read_bytes() == ref.read_bytes()
        h = L.PtHash.load(str(out), 31, kmer_bits=bits)
        assert h.get_kmer_count() == n and h.info.kind == L.PTHASH and h.info.m == 31
        with pytest.raises(L.LphqError):
            L.Mphf.load(str(out), kmer_bits=bits)  # a bare PTHash is not an LPHash file
    r = subprocess.run([exe, "ptbb-build", "-i", str(fa), "-k", "31"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == f"{fa},31,{n},,,,"
    # BBHash half: the CSV's last two fields, a file the loader accepts; gamma < 1 is refused
    bb = tmp_path / "cli.bbh"
    r = subprocess.run([exe, "ptbb-build", "-i", str(fa), "-k", "31", "-b", str(bb), "--kmer-bits", "64", "-g", "2", "-t", "3"], capture_output=True, text=True)
    f = r.stdout.strip().split(",")
    assert r.returncode == 0 and len(f) == 7 and int(f[2]) == n and f[3] == f[4] == "" and int(f[5]) == os.path.getsize(bb) * 8, (r.stdout, r.stderr)
    assert L.BbHash.load(str(bb), 31, kmer_bits=64).get_kmer_count() == n
    assert subprocess.run([exe, "ptbb-build", "-i", str(fa), "-k", "31", "-b", str(bb), "-g", "0.5"], capture_output=True).returncode == 3
    assert subprocess.run([exe, "ptbb-build", "-i", str(tmp_path / "nope.fa"), "-k", "31", "-p", "x"], capture_output=True).returncode == 2
    # the reference ships ptbb_build as an executable of its own (CMakeLists.txt:59): a link to the CLI under that name takes the same arguments
    link = os.path.join(ROOT, "lphash_amd", "ptbb_build")
    if os.path.exists(link):
        viaLink = tmp_path / "link.pthash"
        r = subprocess.run([link, "-i", str(fa), "-k", "31", "-p", str(viaLink), "--kmer-bits", "64", "--cpu"], capture_output=True, text=True)
        assert r.returncode == 0 and viaLink.read_bytes() == (tmp_path / "cli64.pthash").read_bytes()


def bgzf_compress(data, block=60000, level=6):
    """BGZF (bgzip / htslib) container: independent gzip members of <= 64 KiB with the 'BC' extra field, plus the empty EOF member."""
    import struct
    import zlib

    out = bytearray()
    for chunk in [data[i:i + block] for i in range(0, len(data), block)] + [b""]:
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = c.compress(chunk) + c.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(payload) + 25)
        out += payload + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)


@pytest.mark.skipif(not os.path.exists(KSEQ_DUMP), reason="oracle/_ref/kseq_dump is built only where /root/reference exists (oracle/Makefile)")
def test_bgzf_input_is_inflated_block_parallel(tmp_path):
    """A BGZF-compressed FASTA (what bgzip writes; zlib reads it as ordinary multi-member gzip, so the reference does too) is inflated block by
    block on several threads and then parsed by the multi-threaded reader: same records as the reference's kseq.h and as the serial reader;
    a damaged block and a plain single-member .gz fall back to the serial reader (which reports what zlib reports)."""
    import gzip

    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    recs = random_records(77, 40, 1, 30000)
    text = b"".join(b">r%d some text\n" % i + b"\n".join(r[j:j + 61] for j in range(0, len(r), 61)) + b"\n" for i, r in enumerate(recs))
    cases = {"multi": bgzf_compress(text), "one_block": bgzf_compress(text[:5000]), "tiny_blocks": bgzf_compress(text[:40000], block=777),
             "fastq": bgzf_compress(b"@q\nACGT\n+\nIIII\n" * 50), "plain_gz": gzip.compress(text[:100000])}
    broken = bytearray(cases["multi"])
    broken[len(broken) // 2] ^= 0x5A
    for name, blob in cases.items():
        p = tmp_path / f"{name}.fa.gz"
        p.write_bytes(blob)
        want = subprocess.run([KSEQ_DUMP, str(p)], capture_output=True, text=True).stdout.strip()
        ser = subprocess.run([exe, "records", "-i", str(p), "--serial"], capture_output=True, text=True).stdout.strip().split(",")
        par = subprocess.run([exe, "records", "-i", str(p), "-t", "6", "--min-piece", "1", "--batch-bytes", "50000"], capture_output=True, text=True).stdout.strip().split(",")
        assert ",".join(ser[:3]) == want and ",".join(par[:3]) == want, (name, ser, par, want)
        assert (int(par[4]) > 0) == (name in ("multi", "one_block", "tiny_blocks", "fastq")), (name, par)  # batches that came from the parallel parser
    p = tmp_path / "broken.fa.gz"
    p.write_bytes(bytes(broken))
    ser = subprocess.run([exe, "records", "-i", str(p), "--serial"], capture_output=True, text=True).stdout.strip().split(",")
    par = subprocess.run([exe, "records", "-i", str(p), "-t", "6"], capture_output=True, text=True).stdout.strip().split(",")
    assert par[:3] == ser[:3] and int(par[4]) == 0  # whatever zlib makes of the damaged stream, both readers agree (one batch: the damage is in its window)
    # small batches: the windows before the damaged block are parsed in parallel, the serial reader takes over at the batch that meets it
    par = subprocess.run([exe, "records", "-i", str(p), "-t", "6", "--batch-bytes", "50000", "--min-piece", "1"], capture_output=True, text=True).stdout.strip().split(",")
    assert par[:3] == ser[:3] and 0 < int(par[4]) < int(par[3])


def test_plain_gzip_input_is_decoded_by_all_threads(tmp_path):
    """An ordinary .gz -- one member, what `gzip` writes and what the reference's own query files are (data/queries/*.fasta.gz, read through
    gzread inside the timer, src/query.cpp:43-58) -- is decoded by all threads (parallel_gz.hpp) into the sliding window the parallel parser works
    on: the records are the serial reader's (= kseq's) for multi-line FASTA, FASTQ, several members, tiny batches (the window slides and stays far
    smaller than the text), spans smaller than a deflate block; a '+' line, text that is no FASTA, a damaged or truncated stream hand the file to the
    serial reader from the right record on."""
    import gzip

    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    recs = random_records(191, 120, 1000, 160_000)
    fasta = b"".join(b">r%d some comment\n" % i + b"\n".join(r[j:j + 80] for j in range(0, len(r), 80)) + b"\n" for i, r in enumerate(recs))
    reads = random_records(192, 30_000, 100, 250)
    fastq = b"".join(b"@q%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads))
    cut = fasta.index(b"\n>", 3_000_000) + 1
    plus = fasta[:cut] + b">odd\nACGT\n+\nIIII\n" + fasta[cut:]  # a quality line in the middle: from that batch on the serial reader
    cases = {
        "fasta": gzip.compress(fasta, 6),
        "fasta_l1": gzip.compress(fasta, 1),
        "fastq": gzip.compress(fastq, 6),
        "members": b"".join(gzip.compress(fasta[i:i + 2_500_000], 6) for i in range(0, len(fasta), 2_500_000)),
        "plus_line": gzip.compress(plus, 6),
        "not_fasta": gzip.compress(b"no header here\n" + fasta, 6),
    }
    damaged = bytearray(cases["fasta"])
    damaged[len(damaged) // 2] ^= 0x24
    cases["damaged"] = bytes(damaged)
    cases["truncated"] = cases["fasta"][: len(cases["fasta"]) * 2 // 3]
    for name, blob in cases.items():
        p = tmp_path / f"{name}.gz"
        p.write_bytes(blob)
        ser = subprocess.run([exe, "records", "-i", str(p), "--serial"], capture_output=True, text=True).stdout.strip().split(",")
        if os.path.exists(KSEQ_DUMP) and name not in ("damaged", "truncated"):
            assert ",".join(ser[:3]) == subprocess.run([KSEQ_DUMP, str(p)], capture_output=True, text=True).stdout.strip(), name
        for threads, span, batch in (("6", "0", 1 << 29), ("6", "30000", 300_000), ("3", "200000", 1_000_000), ("8", "7000", 50_000)):
            par = subprocess.run([exe, "records", "-i", str(p), "-t", threads, "--batch-bytes", str(batch), "--min-piece", "1000"], capture_output=True, text=True,
                                 env=dict(os.environ, LPHQ_GZ_SPAN=span))
            f = par.stdout.strip().split(",")
            assert par.returncode == 0 and f[:3] == ser[:3], (name, threads, span, batch, f, ser)
            n_batches, n_parallel, window, n_pgz = int(f[3]), int(f[4]), int(f[6]), int(f[7])
            if name in ("fasta", "fasta_l1", "fastq", "members"):
                assert n_pgz == n_batches == n_parallel > 0, (name, f)  # every batch: decoded by the pool, parsed by the parallel parser
                if batch <= 1_000_000:
                    assert window < len(fasta) // 2, (name, f)  # the text is never held whole
            elif name == "not_fasta":
                assert n_pgz == 0
            elif batch <= 1_000_000:  # the batches before the '+' line / the damage came from the pool
                assert 0 < n_pgz < n_batches, (name, f)
    # the same file with the parallel decoder switched off is the serial path of old
    par = subprocess.run([exe, "records", "-i", str(tmp_path / "fasta.gz"), "-t", "6"], capture_output=True, text=True, env=dict(os.environ, LPHQ_GZ_SERIAL="1"))
    assert par.stdout.strip().split(",")[7] == "0"


def test_plain_gzip_reader_fuzz_against_the_serial_reader(tmp_path):
    """Random FASTA / FASTQ texts (line widths, CR LF, empty lines and records, a quality line in a FASTA, no final newline, two members) as .gz through
    the span-parallel decoder + parse-while-decoding reader with random thread counts, span sizes, batch sizes and piece sizes: the records are the serial
    reader's every time (the serial reader is the one pinned to kseq.h)."""
    import gzip
    import random

    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    rng = random.Random(20261002)
    gz = tmp_path / "t.gz"
    for it in range(30):
        kind = rng.choice(["fasta", "fasta", "fastq", "crlf", "weird"])
        big = rng.random() < 0.3
        out = []
        for i in range(rng.randint(1, 300)):
            ln = rng.randint(0, 60000 if big else 3000)
            seq = "".join(rng.choice("ACGT") for _ in range(ln))
            if kind == "fastq":
                out.append("@r%d x\n%s\n+\n%s\n" % (i, seq, "I" * ln))
            else:
                w = rng.choice([60, 70, 80, 100, 100000])
                nl = "\r\n" if kind == "crlf" else "\n"
                out.append(">r%d some comment%s%s%s" % (i, nl, nl.join(seq[j:j + w] for j in range(0, ln, w)), nl if (ln > 0 or rng.random() < 0.5) else ""))
                if kind == "weird" and rng.random() < 0.1:
                    out.append("\n\n")
                if kind == "weird" and rng.random() < 0.03:
                    out.append(">q\nACGT\n+\nIIII\n")
        text = "".join(out).encode()
        if rng.random() < 0.2:
            text = text.rstrip(b"\n")
        if rng.random() < 0.2 and len(text) > 1000:
            gz.write_bytes(gzip.compress(text[: len(text) // 2], 6) + gzip.compress(text[len(text) // 2:], 6))
        else:
            gz.write_bytes(gzip.compress(text, rng.choice([1, 6, 9])))
        ser = subprocess.run([exe, "records", "-i", str(gz), "--serial"], capture_output=True, text=True).stdout.strip().split(",")[:3]
        for threads, span, batch in ((4, rng.choice([3000, 20000, 100000]), rng.choice([5000, 100000, 2000000, 1 << 29])), (7, 0, rng.choice([30000, 1 << 29]))):
            env = dict(os.environ, LPHQ_GZ_SPAN=str(span)) if span else dict(os.environ)
            r = subprocess.run([exe, "records", "-i", str(gz), "-t", str(threads), "--batch-bytes", str(batch), "--min-piece", str(rng.choice([1, 1000, 65536]))],
                               capture_output=True, text=True, env=env)
            assert r.returncode == 0 and r.stdout.strip().split(",")[:3] == ser, (it, kind, len(text), threads, span, batch, ser, r.stdout, r.stderr[-500:])


def test_bgzf_window_is_bounded_by_the_batch_size(tmp_path):
    """The inflated text of a BGZF file is held in a window of a few batches, not whole (a bgzip'd FASTQ of hundreds of GB must not need its
    uncompressed size in RAM): 24 MB of text read in 1-MB batches keeps < 6 MB inflated at once, a record larger than the batch grows the window to
    what that record needs, and the records are the serial reader's in every case (FASTA, FASTQ, many threads, any batch size)."""
    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    recs = random_records(91, 300, 1000, 160_000)
    fasta = b"".join(b">r%d\n" % i + b"\n".join(r[j:j + 80] for j in range(0, len(r), 80)) + b"\n" for i, r in enumerate(recs))
    reads = random_records(92, 30_000, 100, 250)
    fastq = b"".join(b"@q%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads))
    big = random_records(93, 1, 6_000_000, 6_000_000)[0]
    one_big = b">small\nACGT\n>big\n" + b"\n".join(big[j:j + 70] for j in range(0, len(big), 70)) + b"\n>after\nGGCC\n"
    for name, text, batch in (("fasta", fasta, 1 << 20), ("fastq", fastq, 1 << 20), ("fastq_tiny_batches", fastq, 3000), ("one_big_record", one_big, 1 << 20)):
        p = tmp_path / f"{name}.gz"
        p.write_bytes(bgzf_compress(text))
        ser = subprocess.run([exe, "records", "-i", str(p), "--serial"], capture_output=True, text=True).stdout.strip().split(",")
        for threads in ("1", "7"):
            par = subprocess.run([exe, "records", "-i", str(p), "-t", threads, "--batch-bytes", str(batch), "--min-piece", "1000"], capture_output=True, text=True)
            f = par.stdout.strip().split(",")
            assert par.returncode == 0 and f[:3] == ser[:3], (name, threads, f, ser)
            assert int(f[4]) == int(f[3]) > 0  # every batch came from the parallel path
            window = int(f[6])
            if name == "one_big_record":
                assert len(one_big) <= window < 4 * len(one_big)
            else:
                assert 0 < window <= 6 * max(batch, 1 << 17) and window < len(text)


BBHASH_REF = os.path.join(ROOT, "oracle", "_ref", "bbhash_ref")


@pytest.mark.skipif(not os.path.exists(BBHASH_REF), reason="oracle/_ref/bbhash_ref (the real BooPHF) is built only where /root/reference exists")
def test_bbhash_files_round_trip_through_the_real_boophf(tmp_path):
    """The reference's own BooPHF (oracle/_ref/bbhash_ref) LOADS the files lphq_build_bbhash writes and answers with a bijection; files it writes
    load here (fresh inputs, several gammas and widths); corrupted files are rejected by the loader's cross-checks, not mis-hashed."""
    for k, bits, gamma, seed in ((31, 64, 1.0, 1), (27, 64, 3.0, 2), (47, 128, 1.5, 3)):
        recs = [r for r in random_records(600 + seed, 12, k, 4000)]
        fa = tmp_path / "i.fa"
        with open(fa, "wb") as f:
            for i, r in enumerate(recs):
                f.write(b">r%d\n%s\n" % (i, r))
        n = sum(len(r) - k + 1 for r in recs)
        ours, theirs, codes = tmp_path / "ours.bbh", tmp_path / "theirs.bbh", tmp_path / "c.u64"
        st = L.build_bbhash(recs, k, str(ours), kmer_bits=bits, gamma=gamma, threads=4)
        assert st["num_kmers"] == n
        r = subprocess.run([BBHASH_REF, "query", str(ours), str(k), str(bits), str(fa), str(codes)], capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0 and int(r.stdout) == n, r.stderr
        c = np.fromfile(codes, dtype=np.uint64)
        assert len(np.unique(c)) == n and int(c.max()) == n - 1  # ptbb_build --check, by the reference's own lookup()
        subprocess.run([BBHASH_REF, "build", str(fa), str(k), str(bits), str(gamma), str(theirs)], check=True, capture_output=True, cwd=tmp_path)
        h = L.BbHash.load(str(theirs), k, kmer_bits=bits)
        assert h.get_kmer_count() == n
        blob = bytearray(theirs.read_bytes())
        for pos in (9, 40, len(blob) // 2, len(blob) - 3):
            bad = bytearray(blob)
            bad[pos] ^= 0x10
            p = tmp_path / "bad.bbh"
            p.write_bytes(bytes(bad))
            try:
                hb = L.BbHash.load(str(p), k, kmer_bits=bits)
                assert hb.get_kmer_count() == n  # a flip inside an exact-map key or value is not detectable; nothing else may pass
                assert pos == len(blob) - 3
            except L.LphqError as e:
                assert e.code == L._lib.ERR_FORMAT
        for cut in (3, 20, len(blob) - 1):
            p = tmp_path / "cut.bbh"
            p.write_bytes(bytes(blob[:cut]))
            with pytest.raises(L.LphqError):
                L.BbHash.load(str(p), k, kmer_bits=bits)


def _gzip_member(data, level=6, strategy=0, name=None, comment=None, extra=None, hcrc=False):
    """One gzip member (RFC 1952) with the header fields asked for; strategy: zlib.Z_DEFAULT_STRATEGY / Z_FIXED / Z_HUFFMAN_ONLY / Z_RLE."""
    import struct
    import zlib

    flg = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0) | (2 if hcrc else 0)
    head = b"\x1f\x8b\x08" + bytes([flg]) + b"\x00\x00\x00\x00\x00\xff"
    if extra is not None:
        head += struct.pack("<H", len(extra)) + extra
    if name is not None:
        head += name + b"\x00"
    if comment is not None:
        head += comment + b"\x00"
    if hcrc:
        head += struct.pack("<H", zlib.crc32(head) & 0xFFFF)
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return head + c.compress(data) + c.flush() + struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF)


def test_fast_inflate_equals_zlib(tmp_path):
    """The .gz decoder of the serial input path (fast_inflate.hpp / gz_source.hpp, `lphash inflate`) against zlib: stored, fixed and dynamic blocks,
    Huffman-only and run-length streams (long codes, distance-1 matches), every header flag, several members (tiny ones, ones spanning several 4-MiB
    chunks of the decoder's ring), trailing garbage; truncated and corrupted files end the data early with an error and never crash."""
    import zlib

    exe = os.path.join(ROOT, "lphash_amd", "lphash")
    rng = np.random.default_rng(77)
    dna = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=9_000_000))
    fasta = b">r\n" + b"\n".join(dna[i:i + 70] for i in range(0, 3_000_000, 70)) + b"\n"
    repetitive = (b"ACGTTGCA" * 5000 + b"A" * 70000 + dna[:1000]) * 40
    noise = rng.integers(0, 256, 700_000, dtype=np.uint8).tobytes()
    text = b"".join(b"@read%d some text\n%s\n+\n%s\n" % (i, dna[i * 50:i * 50 + 150], b"I" * 150) for i in range(20000))
    cases = {
        "empty": _gzip_member(b""),
        "one_byte": _gzip_member(b"A"),
        "stored": _gzip_member(noise, level=0),
        "incompressible": _gzip_member(noise, level=6),
        "fixed": _gzip_member(fasta[:200_000], strategy=zlib.Z_FIXED),
        "huffman_only": _gzip_member(text, strategy=zlib.Z_HUFFMAN_ONLY),
        "rle": _gzip_member(repetitive, strategy=zlib.Z_RLE),
        "level1": _gzip_member(fasta, level=1),
        "level9_big": _gzip_member(dna + repetitive, level=9),  # > 2 chunks, matches across the chunk borders
        "fastq": _gzip_member(text, level=6),
        "flags": _gzip_member(fasta[:50_000], name=b"x.fa", comment=b"made by a test", extra=b"AB\x02\x00xy", hcrc=True),
        "members": b"".join(_gzip_member(fasta[i:i + 77_777], level=int(rng.integers(0, 10))) for i in range(0, 1_500_000, 77_777)),
        "members_then_garbage": _gzip_member(text[:100_000]) + _gzip_member(b"") + _gzip_member(text[100_000:300_000]) + b"\x00garbage after the last member",
        # hundreds of small members: nothing for the pool to do (it stops itself after 64 pieces decoded by the caller's thread)
        "many_members": b"".join(_gzip_member(fasta[i:i + 9_000], level=6) for i in range(0, 2_700_000, 9_000)),
        # a run of one letter longer than any span's cap of text (the N stretches of a chromosome; placeholders never die out inside it)
        "poly_n": _gzip_member(fasta[:600_000] + b"N" * 40_000_000 + fasta[600_000:1_200_000], level=6),
    }
    plain = {"members_then_garbage": text[:300_000], "members": b"".join(fasta[i:i + 77_777] for i in range(0, 1_500_000, 77_777)), "many_members": fasta[:2_700_000]}
    for name, blob in cases.items():
        f, o = tmp_path / f"{name}.gz", tmp_path / f"{name}.out"
        f.write_bytes(blob)
        want = plain.get(name)
        if want is None:
            d = zlib.decompressobj(31)
            want = d.decompress(blob)
        r = subprocess.run([exe, "inflate", "-i", str(f), "-o", str(o)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.split(",")[:2] == [str(len(want)), "ok"], (name, r.stdout, r.stderr)
        assert o.read_bytes() == want, name
        # the same stream decoded by several threads (parallel_gz.hpp): spans of the compressed file from guessed block starts, with placeholders for
        # the unknown history; spans far smaller than a block, spans of many blocks, more spans than threads and fewer
        for threads, span in ((2, 0), (3, 20_000), (8, 65_536), (8, 5_000), (5, 300_000)):
            r = subprocess.run([exe, "inflate", "-i", str(f), "-o", str(o), "--parallel", str(threads)] + (["--span", str(span)] if span else []),
                               capture_output=True, text=True)
            if r.returncode == 2:  # a file of less than two spans is left to the serial decoder
                assert len(blob) < 2 * max(span, 128 << 10), (name, threads, span)
                continue
            assert r.returncode == 0 and r.stdout.split(",")[:2] == [str(len(want)), "ok"], (name, threads, span, r.stdout, r.stderr)
            assert o.read_bytes() == want, (name, threads, span)
            n_par, n_ser = int(r.stdout.split(",")[3]), int(r.stdout.split(",")[4])
            if name in ("level1", "level9_big", "huffman_only") and span >= 20_000:  # dynamic blocks throughout: the pool does the work
                assert n_par >= 3 * n_ser, (name, threads, span, r.stdout)
    # the reader on top of it yields the records kseq / zlib yield (`lphash records --serial` with and without LPHQ_GZ_ZLIB)
    for name in ("level1", "fastq", "members"):
        a = subprocess.run([exe, "records", "-i", str(tmp_path / f"{name}.gz"), "--serial"], capture_output=True, text=True)
        b = subprocess.run([exe, "records", "-i", str(tmp_path / f"{name}.gz"), "--serial"], capture_output=True, text=True, env=dict(os.environ, LPHQ_GZ_ZLIB="1"))
        assert a.returncode == 0 and a.stdout.split(",")[:3] == b.stdout.split(",")[:3] and int(a.stdout.split(",")[0]) > 0, (name, a.stdout, b.stdout)
    # damage: a prefix of the data comes out, the call says "error", nothing crashes
    good = cases["level1"]
    want = fasta
    for trial in range(40):
        bad = bytearray(good)
        if trial % 2 == 0:
            bad = bad[: int(rng.integers(1, len(bad)))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                bad[int(rng.integers(10, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        f, o = tmp_path / "bad.gz", tmp_path / "bad.out"
        f.write_bytes(bytes(bad))
        for par in ([], ["--parallel", "4", "--span", "30000"]):
            r = subprocess.run([exe, "inflate", "-i", str(f), "-o", str(o)] + par, capture_output=True, text=True)
            if r.returncode == 2:  # not even a header left (or, for the parallel decoder, less than two spans)
                continue
            assert r.returncode in (0, 3), (trial, par, r.returncode, r.stderr)
            got = o.read_bytes()
            if r.returncode == 0:  # the damage was harmless (e.g. inside the trailing bytes that nothing reads): then the data must be right
                assert got == want, (trial, par)
            elif trial % 2 == 0:  # truncated: everything decoded is a prefix of the truth
                assert want.startswith(got), (trial, par)


def test_bbhash_of_no_keys_loads_its_own_file(tmp_path):
    """A function over no key (every record shorter than k): the level sizes written (one 64-bit word each) are the ones the loader recomputes --
    the size formula of BooPHF.hpp:1027-1043 divides by the number of keys and is not evaluated for zero."""
    path = str(tmp_path / "empty.bbh")
    st = L.build_bbhash([b"ACG", b""], 5, path, kmer_bits=64, gamma=1.0, threads=2)
    assert st["num_kmers"] == 0
    h = L.BbHash.load(path, 5, kmer_bits=64)
    assert h.get_kmer_count() == 0
    h.close()


def test_abi_version_and_both_count_valid_entry_points_are_exported():
    """ABI 2 added lphq_count_valid_on_stream and kept the four-argument lphq_count_valid of ABI 1 (a caller compiled against the first header must
    keep working: an entry point never changes its signature)."""
    assert L.lib.lphq_abi_version() == 4
    hdr = open(os.path.join(ROOT, "include", "lphq.h")).read()
    assert "#define LPHQ_ABI_VERSION 4" in hdr
    assert "int lphq_count_valid(const lphq_index* idx, const uint64_t* d_codes, size_t n, uint64_t* count);" in hdr
    assert "int lphq_count_valid_on_stream(const lphq_index* idx, const uint64_t* d_codes, size_t n, void* stream, uint64_t* count);" in hdr


# ---- files written by a real reference build (tests/golden/ref/README.md): skipped until one is supplied ----
def test_reference_built_files_pin_loader_oracle_and_builders():
    """For every case directory under tests/golden/ref: the loader accepts the reference's file, the oracle reproduces the reference's codes,
    both builders reproduce the file's bytes.  This is the test that turns "parity unpinned" into "pinned"."""
    import ref_intake

    found = ref_intake.cases()
    if not found:
        pytest.skip("no reference-built file supplied (tests/golden/ref/README.md has the recipe)")
    for d in found:
        ran = ref_intake.check_case_cpu(ref_intake.Case(d), L, orc)
        assert "loader" in ran, d


@pytest.mark.parametrize("kind,k,m,bits", [("p", 31, 20, 64), ("u", 31, 20, 64), ("p", 63, 17, 128)])
def test_reference_intake_dry_run_cpu(tmp_path, kind, k, m, bits):
    """The intake machinery itself, on a case the ORACLE wrote into a temporary directory in exactly the layout of README.md (gzip multi-line
    FASTA input, query file, .u64 code dumps, meta.json): every check runs and passes; a corrupted code dump and a file of the wrong kind fail."""
    import ref_intake

    d = ref_intake.write_oracle_case(str(tmp_path / "case"), orc, kind=kind, k=k, m=m, kmer_bits=bits)
    assert ref_intake.cases(str(tmp_path)) == [d]
    ran = ref_intake.check_case_cpu(ref_intake.Case(d), L, orc)
    assert ran == ["loader", "oracle-codes-streaming", "oracle-codes-random", "builder-bytes", "oracle-builder-bytes"]
    codes = np.fromfile(os.path.join(d, "codes_random.u64"), dtype="<u8")
    codes[len(codes) // 2] ^= 1
    codes.tofile(os.path.join(d, "codes_random.u64"))
    with pytest.raises(AssertionError, match="oracle codes differ"):
        ref_intake.check_case_cpu(ref_intake.Case(d), L, orc)
    meta = json.load(open(os.path.join(d, "meta.json")))
    meta["kind"] = "u" if kind == "p" else "p"
    json.dump(meta, open(os.path.join(d, "meta.json"), "w"))
    with pytest.raises(L.LphqError):  # the loader's cross-checks refuse a file read as the other class
        ref_intake.check_case_cpu(ref_intake.Case(d), L, orc)


def test_reference_intake_fastx_reader_matches_the_product_reader(tmp_path):
    """ref_intake.read_fastx (what feeds the builders in the intake) yields the records the product's reader -- pinned to the real kseq.h -- yields."""
    import ref_intake

    fa = tmp_path / "x.fa.gz"
    with gzip.open(fa, "wb") as f:
        f.write(b">a desc\nACGT\nACG\n>b\n\n>c\nTTTT\r\nGG\n")
    assert ref_intake.read_fastx(str(fa)) == [b"ACGTACG", b"", b"TTTTGG"]
    fq = tmp_path / "x.fq"
    fq.write_bytes(b"@r1\nACGTN\n+\n@IIII\n@r2\nGGCC\n+r2\n>>>>\n")
    assert ref_intake.read_fastx(str(fq)) == [b"ACGTN", b"GGCC"]
    n, bases, _ = orc.file_digest(str(fa))
    assert (n, bases) == (3, 13)


def test_integration_stub_is_built_and_needs_a_gpu(tmp_path):
    """examples/query_example{64,128}: the reference-side binding of INTEGRATION.md section 2 (examples/gpu_mphf.hpp + the reference's query<MPHF>
    driver restated), compiled with the plain host compiler against include/lphq.h by `make`.  Without a GPU (here) it must fail loudly -- there
    is no CPU path behind the C ABI -- and keep the reference's exit codes for usage errors."""
    for bits in (64, 128):
        exe = os.path.join(ROOT, "examples", f"query_example{bits}")
        assert os.path.exists(exe), "make -C lphash_amd/csrc builds the examples"
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 1 and "usage" in r.stderr
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU test runs the binding")
    path = os.path.join(ROOT, "tests", "golden", "p_k31_m20_b64.lph")
    fa = tmp_path / "q.fa"
    fa.write_bytes(b">q\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    r = subprocess.run([os.path.join(ROOT, "examples", "query_example64"), "query-p", "-i", path, "-q", str(fa)], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout == "" and "device" in r.stderr.lower()
