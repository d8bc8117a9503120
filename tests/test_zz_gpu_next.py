"""GPU tests of the SURVEY section 8(f) "next" components (index construction on the GPU, opt-in kernels).

Kept in a file that sorts after test_gpu_parity.py so that `pytest -x` reaches every query-path test (section 8 a-e) first."""
import os

import pytest

pytestmark = pytest.mark.gpu

import lphash_amd as L  # noqa: E402
import orc  # noqa: E402  (the checker)
from lphash_amd.synth import synthetic_records  # noqa: E402
from util import dedup_records  # noqa: E402


def torch_dev():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a device; the product has no CPU fallback"
    return torch, torch.device("cuda", 0)


@pytest.mark.parametrize("k,m,bits,kind,nrec,rlen", [(31, 20, 64, 0, 40, 30_000), (63, 20, 128, 0, 30, 40_000), (31, 20, 64, 1, 40, 30_000), (21, 8, 64, 0, 20, 20_000),
                                                     (31, 15, 128, 0, 50, 60_000)])
def test_gpu_pilot_search_build_is_byte_identical(tmp_path, k, m, bits, kind, nrec, rlen):
    """lphq_build_device (PTHash pilot search on the GPU: speculative window + in-order commit) must write exactly the
    file the sequential first-fit search writes -- and exactly the file the oracle's restatement of the reference's construction
    (src/partitioned_mphf.cpp:33-268, src/unpartitioned_mphf.cpp:23-170, include/minimizer.hpp:11-319) writes."""
    _, recs = synthetic_records(seed=11 + k, n_records=nrec, length=rlen)
    recs = dedup_records(recs, k) if k < 25 else recs
    pc, pg = str(tmp_path / "cpu.lph"), str(tmp_path / "gpu.lph")
    sc = L.build(recs, k, m, pc, kind=kind, kmer_bits=bits, threads=8)
    sg = L.build(recs, k, m, pg, kind=kind, kmer_bits=bits, threads=8, device=0)
    assert sc == sg
    assert open(pc, "rb").read() == open(pg, "rb").read()
    po = str(tmp_path / "oracle.lph")
    orc.build(recs, k, m, po, kind=kind, kmer_bits=bits)
    assert open(pg, "rb").read() == open(po, "rb").read(), "GPU-built file differs from the oracle's"


@pytest.mark.parametrize("k,m,bits,kind", [(31, 20, 64, 0), (63, 20, 128, 0), (31, 15, 64, 1), (15, 9, 64, 0), (47, 32, 128, 0), (33, 5, 128, 0)])
def test_gpu_super_kmer_extraction_build_is_byte_identical(tmp_path, k, m, bits, kind):
    """Parts 1-2 of construction on the GPU (k_extract_runs + radix sort by minimizer, extract.hip) must lead to the file the host extraction
    leads to: ragged batches with empty / short records, non-ACGT bases, many short records per tile and records spanning many tiles; in one
    batch, in several batches, and with a memory limit that spills sorted pieces."""
    import numpy as np
    from util import random_records

    recs = dedup_records(random_records(77 + k, 30, k - 1, 40_000) + random_records(78 + k, 400, 0, 3 * k) + random_records(79 + k, 3, 150_000, 300_000), k)
    noisy = bytearray(recs[0])
    for p in np.random.default_rng(k).integers(0, len(noisy), 9):
        noisy[p] = ord("N")
    recs[0] = bytes(noisy)
    recs.insert(3, b"")
    pc, pg = str(tmp_path / "cpu.lph"), str(tmp_path / "gpu.lph")
    os.environ["LPHQ_BUILD_HOST_EXTRACT"] = "1"
    try:
        sc = L.build(recs, k, m, pc, kind=kind, kmer_bits=bits, threads=8, device=0)  # host extraction, GPU pilot search
    finally:
        del os.environ["LPHQ_BUILD_HOST_EXTRACT"]
    want = open(pc, "rb").read()
    sg = L.build(recs, k, m, pg, kind=kind, kmer_bits=bits, threads=8, device=0)
    assert sg == sc and open(pg, "rb").read() == want
    po = str(tmp_path / "oracle.lph")
    orc.build(recs, k, m, po, kind=kind, kmer_bits=bits)  # the oracle's restatement of the reference's construction, directly
    assert want == open(po, "rb").read(), "GPU-built file (extraction + sort + pilot search on the device) differs from the oracle's"
    os.environ["LPHQ_EXTRACT_INITIAL_CAP"] = "1000"  # the first launch runs out of room: the kernel only counts on, the second launch has the exact size
    try:
        sg = L.build(recs, k, m, pg, kind=kind, kmer_bits=bits, threads=8, device=0)
    finally:
        del os.environ["LPHQ_EXTRACT_INITIAL_CAP"]
    assert sg == sc and open(pg, "rb").read() == want
    for per_batch, limit in ((50, 0), (7, 1 << 20), (len(recs), 1 << 19)):
        def batches():
            for i in range(0, len(recs), per_batch):
                yield recs[i:i + per_batch]

        ss = L.build_stream(batches, k, m, pg, kind=kind, kmer_bits=bits, threads=8, device=0, tmp_dir=str(tmp_path / "spill"), max_memory_bytes=limit)
        assert ss == sc and open(pg, "rb").read() == want, (per_batch, limit)


def test_gpu_extraction_follows_the_reference_on_dirty_index_input(tmp_path):
    """k_extract_runs on index sets made of stretches of k - 1, k, k + 1, m, ... valid bases between non-ACGT bytes: a stretch of EXACTLY k bases closed by such a
    byte is counted as a k-mer and pushed into no super-k-mer by the reference (include/minimizer.hpp:60-66,139-152); the GPU stages -- one batch (classified on
    the device), several batches, pieces under a memory limit -- write the oracle's file, k-mer count included."""
    from util import dirty_index_records, random_records

    torch_dev()
    n = 0
    for seed in range(60):
        k = [15, 21, 31, 47, 63][seed % 5]
        m = [7, 20, 13, 17, 5][seed % 5] if k > 21 else [7, 11, 5][seed % 3]
        bits = 64 if k <= 31 else 128
        recs = dirty_index_records(1000 + seed, k, m, n_records=9)
        if recs is None:
            continue
        recs = recs + random_records(seed, 2, 3000, 9000)  # (several tiles)
        po, pg = str(tmp_path / "o.lph"), str(tmp_path / "g.lph")
        try:
            orc.build(recs, k, m, po, kind=seed % 2, kmer_bits=bits)
        except RuntimeError:
            continue
        want = open(po, "rb").read()
        st = L.build(recs, k, m, pg, kind=seed % 2, kmer_bits=bits, threads=4, device=0)
        assert open(pg, "rb").read() == want, (seed, k, m)
        assert st["num_kmers"] == orc.Index(po, seed % 2, bits).nkmers
        L.build_stream(lambda: iter([recs[:3], recs[3:7], recs[7:]]), k, m, pg, kind=seed % 2, kmer_bits=bits, threads=4, device=0, tmp_dir=str(tmp_path / "spill"),
                       max_memory_bytes=(1 << 14) if seed % 2 else 0)
        assert open(pg, "rb").read() == want, (seed, k, m, "streamed")
        n += 1
    assert n > 40


def test_gpu_pilot_search_giving_up_falls_back_to_the_cpu_search(tmp_path, capfd):
    """A GPU pilot search that runs out of its wall-clock limit (here: a millisecond, checked after every round) hands the buckets to the sequential CPU
    search -- loudly, and with the same file: in memory (the whole search again) and with a memory limit (piece by piece, from the table of taken positions
    the pieces before left)."""
    from util import random_records

    torch_dev()
    recs = dedup_records(random_records(2024, 40, 2000, 60_000), 31)
    want_path, got = str(tmp_path / "cpu.lph"), str(tmp_path / "got.lph")
    want_stats = L.build(recs, 31, 20, want_path, kmer_bits=64, threads=8)
    want = open(want_path, "rb").read()
    os.environ["LPHQ_SEARCH_MAX_SECONDS"] = "0.001"
    os.environ["LPHQ_SEARCH_ROUNDS_PER_SYNC"] = "1"
    try:
        capfd.readouterr()
        s1 = L.build(recs, 31, 20, got, kmer_bits=64, threads=8, device=0)
        err1 = capfd.readouterr().err
        assert s1 == want_stats and open(got, "rb").read() == want
        s2 = L.build_stream(lambda: iter([recs[:11], recs[11:]]), 31, 20, got, kmer_bits=64, threads=8, device=0, tmp_dir=str(tmp_path / "spill"), max_memory_bytes=1 << 20)
        err2 = capfd.readouterr().err
        assert s2 == want_stats and open(got, "rb").read() == want
    finally:
        del os.environ["LPHQ_SEARCH_MAX_SECONDS"], os.environ["LPHQ_SEARCH_ROUNDS_PER_SYNC"]
    assert "falling back to the sequential CPU search" in err1 and err2.count("falling back to the sequential CPU search") >= 2, (err1[-300:], err2[-600:])


@pytest.mark.parametrize("k,bits", [(31, 64), (31, 128), (63, 128)])
def test_pthash_competitor_harness(tmp_path, k, bits):
    """the PTHash half of the reference's ptbb_build / ptbb_query on the GPU.  Codes == the oracle's restatement of
    pthash_mphf(kmer) (src/ptbb/ptbb_query.cpp:47) for packed batches (members and non-members) and for records (one code per window of
    k valid bases, as ptbb_file_itr yields them); bijection onto [0, n); `lphash ptbb-query` prints the reference's CSV."""
    import subprocess

    import numpy as np
    import orc
    from util import pack_kmers, random_records

    torch, dev = torch_dev()
    _, recs = synthetic_records(seed=5 + k, n_records=12, length=20_000)
    path = str(tmp_path / "p.pthash")
    sg = L.build_pthash(recs, k, path, kmer_bits=bits, threads=8, device=0)  # GPU pilot search
    ref = str(tmp_path / "o.pthash")
    orc.ptbb_build(recs, k, ref, kmer_bits=bits)
    assert open(path, "rb").read() == open(ref, "rb").read()
    o = orc.PtHash(path, k, bits)
    h = L.PtHash.load(path, k, kmer_bits=bits, device=0)
    n = sum(len(r) - k + 1 for r in recs)
    assert sg["num_kmers"] == n == h.get_kmer_count() == o.num_keys
    allc = []
    for r in recs[:4]:
        q = bytearray(r)
        q[100] = q[5000] = ord("N")
        for rec in (r, bytes(q)):
            got = h(rec)
            assert np.array_equal(got, o(rec))
        allc.append(h(r))
    seq = recs[5][:3000] + random_records(3, 1, 2000, 2000)[0]
    pos = np.arange(0, len(seq) - k + 1, dtype=np.int64)
    pk = pack_kmers(seq, pos, k, bits)
    d_k = torch.from_numpy(pk.view(np.int64)).to(dev)
    d_o = torch.empty(len(pos), dtype=torch.int64, device=dev)
    h.query_packed(d_k.data_ptr(), len(pos), d_o.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_o.cpu().numpy().view(np.uint64), o.query_packed(pk))
    every = np.concatenate([h(r) for r in recs])
    assert len(every) == n and len(np.unique(every)) == n and int(every.max()) == n - 1
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lphash_amd", "lphash")
    fa = tmp_path / "q.fa"
    with open(fa, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">r%d\n%s\n" % (i, r))
    r = subprocess.run([exe, "ptbb-query", "-q", str(fa), "-k", str(k), "-p", path, "--kmer-bits", str(bits)], capture_output=True, text=True)
    f5 = r.stdout.strip().split(",")
    assert r.returncode == 0 and len(f5) == 6 and f5[0] == str(fa) and f5[1] == str(k) and f5[2] == path and float(f5[3]) > 0 and f5[4] == f5[5] == "", (r.stdout, r.stderr)
    r = subprocess.run([exe, "ptbb-build", "-i", str(fa), "-k", str(k), "-p", str(tmp_path / "c.pthash"), "--kmer-bits", str(bits), "--check"], capture_output=True, text=True)
    assert r.returncode == 0 and "EVERYTHING OK" in r.stderr, r.stderr


def _bbhash_cases():
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bbhash")
    return sorted(os.path.splitext(f)[0] for f in os.listdir(d) if f.endswith(".npz"))


@pytest.mark.parametrize("name", _bbhash_cases())
def test_bbhash_competitor_against_the_real_boophf(tmp_path, name):
    """The BBHash half of the competitor harness on the GPU, PINNED against the reference's own code: tests/golden/bbhash/*.bbh were written by the real
    BooPHF (tests/golden/make_bbhash_golden.py) and the expected codes are its lookup().  Where the prebuilt oracle/_ref/bbhash_ref travelled to this
    box, a larger fresh set is compared with it too."""
    import subprocess

    import numpy as np
    from util import pack_kmers

    torch, dev = torch_dev()
    here = os.path.dirname(os.path.abspath(__file__))
    z = np.load(os.path.join(here, "golden", "bbhash", name + ".npz"))
    k, bits = int(z["k"]), int(z["kmer_bits"])
    h = L.BbHash.load(os.path.join(here, "golden", "bbhash", name + ".bbh"), k, kmer_bits=bits, device=0)
    queries = [bytes(z["queries"][z["q_off"][i]:z["q_off"][i + 1]]) for i in range(len(z["q_off"]) - 1)]
    want, at = z["codes"], 0
    for q in queries:
        got = h(q)  # lphq_query_contig: one code per window of k valid bases
        assert np.array_equal(got, want[at:at + len(got)])
        at += len(got)
    assert at == len(want)
    seq = queries[0]
    pos = np.arange(0, len(seq) - k + 1, dtype=np.int64)
    pk = pack_kmers(seq, pos, k, bits)
    d_k = torch.from_numpy(pk.view(np.int64)).to(dev)
    d_o = torch.empty(len(pos), dtype=torch.int64, device=dev)
    h.query_packed(d_k.data_ptr(), len(pos), d_o.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_o.cpu().numpy().view(np.uint64), want[: len(pos)])
    # a larger function from the product's builder: bijection on the GPU; and the real BooPHF agrees code for code where its binary is here
    if k >= 20:
        _, recs = synthetic_records(seed=21 + k, n_records=10, length=30_000)
    else:  # 4^k is too small for 300 K distinct random k-mers: the (duplicate-free) records of the fixture
        recs = [bytes(z["records"][z["rec_off"][i]:z["rec_off"][i + 1]]) for i in range(len(z["rec_off"]) - 1)]
    ours = str(tmp_path / "ours.bbh")
    st = L.build_bbhash(recs, k, ours, kmer_bits=bits, gamma=float(z["gamma"]), threads=8)
    h2 = L.BbHash.load(ours, k, kmer_bits=bits, device=0)
    every = np.concatenate([h2(r) for r in recs])
    n = st["num_kmers"]
    assert len(every) == n and len(np.unique(every)) == n and int(every.max()) == n - 1
    ref = os.path.join(os.path.dirname(here), "oracle", "_ref", "bbhash_ref")
    if os.path.exists(ref):
        fa, codes = tmp_path / "i.fa", tmp_path / "c.u64"
        with open(fa, "wb") as f:
            for i, r in enumerate(recs):
                f.write(b">r%d\n%s\n" % (i, r))
        r = subprocess.run([ref, "query", ours, str(k), str(bits), str(fa), str(codes)], capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0, r.stderr
        assert np.array_equal(np.fromfile(codes, dtype=np.uint64), every)
    exe = os.path.join(os.path.dirname(here), "lphash_amd", "lphash")
    fa2 = tmp_path / "q.fa"
    with open(fa2, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">r%d\n%s\n" % (i, r))
    r = subprocess.run([exe, "ptbb-query", "-q", str(fa2), "-k", str(k), "-b", ours, "--kmer-bits", str(bits)], capture_output=True, text=True)
    f6 = r.stdout.strip().split(",")
    assert r.returncode == 0 and len(f6) == 6 and f6[2] == f6[3] == "" and f6[4] == ours and float(f6[5]) > 0, (r.stdout, r.stderr)
    r = subprocess.run([exe, "ptbb-build", "-i", str(fa2), "-k", str(k), "-b", str(tmp_path / "c.bbh"), "--kmer-bits", str(bits), "--check", "-t", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "EVERYTHING OK" in r.stderr, r.stderr


@pytest.mark.parametrize("mode", ["random", "stream"])
def test_bench_collective_path_with_one_rank(tmp_path, mode):
    """bench.py's N > 1 code (RCCL process group, image broadcast, handle over the received image, replica check, max-over-ranks timing) with the
    one rank a 1-GPU box has: the JSON line says what the collective saw and that the replica answers like the oracle."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--dist-single", "--mode", mode, "--kmers", "2000000", "--batch-log2", "20", "--steps", "2", "--warmup", "1",
           "--no-hbm-probe", "--b1-records", "0", "--cache-dir", str(tmp_path)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["rccl_ranks"] == 1 and line["replicas_agree"] is True and line["parity_vs_oracle"] is True
    assert line["image_broadcast_ms"] > 0 and line["n_gpus"] == 1 and line["roofline"]["frac"] > 0
    # what a scaling run reports per rank: the kernel time of every rank (a straggler must be visible, not only the max), how many ranks
    # compared their own timed batch with the oracle, and the CPU baseline of rank 0 at any N
    assert len(line["kernel_ms_per_rank"]) == 1 and line["kernel_ms_min"] == line["kernel_ms_max"] == line["kernel_ms_per_rank"][0] > 0
    assert line["parity_ranks_checked"] == 1 and line["cpu_baseline"]["value"] > 0 and "secondary" not in line
    # the strong-scaling pass (a fixed job sharded over the ranks) with the one rank: its shard is the whole job -- 2^(20+3) probes of the common seeded set,
    # or 64 whole passes over the 20 records (the fixed job of a streaming workload is whole passes adding up to 2^32 windows, at most 64 of them: round 6);
    # checked against the oracle, streaming also as passes x a permutation of [0, n) by the all-reduced checksums
    st = line["strong"]
    assert line["scaling"] == "weak" and st["scaling"] == "strong" and st["n_gpus"] == 1 and st["parity_vs_oracle"] is True and st["value"] > 0
    if mode == "random":
        assert st["total_units_per_step"] == 1 << 23 and st["shards"] == [[0, 1 << 23]] and st["all_shards_cover_0_n"] is None
    else:
        assert st["passes"] == 64 and st["total_units_per_step"] == 64 * 2_000_000 and st["shards"] == [[0, 64 * 20]] and st["all_shards_cover_0_n"] is True
        assert st["empty_step_ms"] > 0 and len(st["planned_shard_ms"]) == 1
    assert line["roofline"]["bound"] in ("fabric_line_requests", "hbm", "valu_issue") and line["roofline"]["frac_from_ms_per_step"] > 0
    assert line["index_sha256"] and line["index_digest_pinned"] is None  # (2 M k-mers: not one of the pinned full-size sets)


@pytest.mark.parametrize("mode", ["random", "stream"])
def test_bench_two_ranks_on_one_gpu(tmp_path, mode):
    """`bench.py --gpus 2` as TWO PROCESSES on the one GPU of the box (LPHQ_BENCH_SHARE_DEVICE0: both ranks on cuda:0, collectives over gloo because RCCL refuses two
    ranks on one device): everything of the N > 1 run but the transport -- bench.py launches its own ranks, rank 0 generates the records and shares them through the
    cache directory, rank 1 reads that file and receives the image by broadcast into ITS process, each rank times its own batch and checks it against the oracle, the
    verdicts are all-reduced, the strong-scaling pass cuts the fixed job in two different shards (and, streaming, proves by the all-reduced checksums that the two
    halves together are a permutation of [0, n)), rank 0 prints one line."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", mode, "--kmers", "3000000", "--batch-log2", "20", "--steps", "3", "--warmup", "1",
           "--no-hbm-probe", "--b1-records", "0", "--cache-dir", str(tmp_path)]
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LPHQ_BENCH_SHARE_DEVICE0"] = "1"
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["collective_backend"] == "gloo" and line["scaling"] == "weak"
    assert line["parity_vs_oracle"] is True and line["parity_ranks_checked"] == 2 and line["replicas_agree"] is True
    assert len(line["kernel_ms_per_rank"]) == 2 and all(x > 0 for x in line["kernel_ms_per_rank"]) and line["image_broadcast_ms"] > 0
    units = line["config"]["units_per_step_per_gpu"]
    assert abs(line["value"] - 2 * units * line["steps"] / (line["ms_per_step"] * line["steps"] * 1e-3)) / line["value"] < 1e-6  # whole-job value = both ranks' units / max time
    st = line["strong"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 2 and st["parity_vs_oracle"] is True and len(st["kernel_ms_per_rank"]) == 2
    if mode == "random":
        assert st["total_units_per_step"] == 1 << 23 and st["shards"] == [[0, 1 << 22], [1 << 22, 1 << 23]]
    else:
        assert st["passes"] == 64 and st["total_units_per_step"] == 64 * 3_000_000 and st["shards"] == [[0, 960], [960, 1920]] and st["all_shards_cover_0_n"] is True
    assert "shared_file" in r.stderr or "records_seed1_30x" in r.stderr or os.path.exists(os.path.join(str(tmp_path), "records_seed1_30x100030.u8"))


@pytest.mark.parametrize("name", ["p_k31_m20_b64", "u_k31_m20_b64", "p_k63_m20_b128", "p_k21_m8_b64_fallback"])
def test_integration_stub_query_driver(tmp_path, name):
    """The reference-side binding of INTEGRATION.md as compiled code: examples/query_example (gpu_mphf<> over include/lphq.h under the reference's
    query<MPHF> driver, built with g++) on a committed fixture -- the 5-field CSV of src/query.cpp:83-86 with the oracle's k-mer count, and the
    codes of both passes equal to the oracle's."""
    import subprocess

    import numpy as np

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(root, "tests", "golden", name + ".npz"))
    kind, bits = int(z["kind"]), int(z["kmer_bits"])
    lph = os.path.join(root, "tests", "golden", name + ".lph")
    queries = [bytes(z["queries"][z["q_off"][i]:z["q_off"][i + 1]]) for i in range(len(z["q_off"]) - 1)]
    queries = [q for q in queries if q]  # (a FASTA file cannot carry the fixture's empty record as a record of its own)
    fa = tmp_path / "q.fa"
    with open(fa, "wb") as f:
        for i, q in enumerate(queries):
            f.write(b">q%d\n%s\n" % (i, q))
    o = orc.Index(lph, kind, bits)
    want_s = np.concatenate([o(q, True) for q in queries])
    want_r = np.concatenate([o(q, False) for q in queries])
    exe = os.path.join(root, "examples", f"query_example{bits}")
    import gzip

    gz = tmp_path / "q.fa.gz"
    gz.write_bytes(gzip.compress(fa.read_bytes(), 6))
    # the batched loop (gpu_mphf::submit / collect over lphq_session_*, the default), the reference's literal per-record loop, and the batched loop on a .gz
    for q, extra in ((fa, []), (fa, ["--per-record"]), (gz, [])):
        for f in ("codes.streaming.u64", "codes.random.u64"):
            if (tmp_path / f).exists():
                (tmp_path / f).unlink()
        r = subprocess.run([exe, "query-p" if kind == 0 else "query-u", "-i", lph, "-q", str(q), "--dump", str(tmp_path / "codes")] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        f5 = r.stdout.strip().split(",")
        assert len(f5) == 5 and f5[0] == str(q) and f5[1] == lph and int(f5[2]) == len(want_s) and float(f5[3]) > 0 and float(f5[4]) > 0, (extra, f5)
        assert np.array_equal(np.fromfile(str(tmp_path / "codes.streaming.u64"), dtype="<u8"), want_s), extra
        assert np.array_equal(np.fromfile(str(tmp_path / "codes.random.u64"), dtype="<u8"), want_r), extra
    # ABI 4: many small batches over THREE lanes (one session over three replica slots of the one GPU: lphq_session_open_devices) -- the codes come back in the
    # order of the file --, and the streaming pass collected as runs (--runs: the same k-mer count without a code ever expanded)
    from util import random_records

    big = queries + random_records(77, 150, 50, 4000) + queries
    fb = tmp_path / "big.fa"
    with open(fb, "wb") as f:
        for i, q in enumerate(big):
            f.write(b">b%d\n%s\n" % (i, q))
    big_s = np.concatenate([o(q, True) for q in big if len(q) >= int(z["k"])] or [np.zeros(0, np.uint64)])
    for extra in (["--devices", "0,0,0", "--batch-bases", "9000"], ["--batch-bases", "9000"]):
        for f in ("codes.streaming.u64", "codes.random.u64"):
            if (tmp_path / f).exists():
                (tmp_path / f).unlink()
        r = subprocess.run([exe, "query-p" if kind == 0 else "query-u", "-i", lph, "-q", str(fb), "--dump", str(tmp_path / "codes")] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and int(r.stdout.strip().split(",")[2]) == len(big_s), (extra, r.stdout, r.stderr)
        assert np.array_equal(np.fromfile(str(tmp_path / "codes.streaming.u64"), dtype="<u8"), big_s), extra
        assert np.array_equal(np.fromfile(str(tmp_path / "codes.random.u64"), dtype="<u8"), np.concatenate([o(q, False) for q in big if len(q) >= int(z["k"])])), extra
    for extra in (["--runs"], ["--runs", "--devices", "0,0,0", "--batch-bases", "9000"]):
        r = subprocess.run([exe, "query-p" if kind == 0 else "query-u", "-i", lph, "-q", str(fb)] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and int(r.stdout.strip().split(",")[2]) == len(big_s), (extra, r.stdout, r.stderr)
    # exit code 2 for a query file that cannot be opened (src/query.cpp:43-46)
    r = subprocess.run([exe, "query-p" if kind == 0 else "query-u", "-i", lph, "-q", str(tmp_path / "missing.fa")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "Unable to open the input file" in r.stderr


@pytest.mark.parametrize("tool,k,m,bits", [("build-p", 31, 20, 64), ("build-u", 31, 15, 64), ("build-p", 63, 20, 128)])
def test_integration_stub_build_driver(tmp_path, tool, k, m, bits):
    """The build side of the same binding (gpu_mphf<>::build = lphq_build_stream over a record source, replacing MPHF::build + essentials::save of
    src/build.cpp:24-28): the file it writes is the oracle's, byte for byte."""
    import subprocess

    from util import random_records

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    recs = dedup_records(random_records(500 + k, 12, k - 1, 20_000), k)
    fa = tmp_path / "in.fa"
    with open(fa, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">r%d\n%s\n" % (i, r))
    out = tmp_path / "x.lph"
    r = subprocess.run([os.path.join(root, "examples", f"query_example{bits}"), tool, "-i", str(fa), "-k", str(k), "-m", str(m), "-o", str(out), "-t", "4", "-d", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    po = str(tmp_path / "o.lph")
    orc.build(recs, k, m, po, kind=0 if tool == "build-p" else 1, kmer_bits=bits)
    assert open(out, "rb").read() == open(po, "rb").read()
