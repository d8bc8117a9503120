"""Run by tests/test_gpu_parity.py::test_pipelined_stream_kernel in a child process with LPHQ_STREAM_PIPELINE=1 (the launcher reads
the switch once per process): the software-pipelined streaming kernel (lphash_amd/csrc/kernels_stream_pipe.cuh) against the CPU
oracle and against the default streaming kernel's contract, over ragged multi-record batches, every alignment class and all four
(kind, kmer width) instantiation families.  Prints PIPE_OK on success."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lphash_amd as L  # noqa: E402
import orc  # noqa: E402
from util import dedup_records, random_records  # noqa: E402


def main():
    assert os.environ.get("LPHQ_STREAM_PIPELINE") == "1"
    import torch
    dev = torch.device("cuda:0")
    tmp = tempfile.mkdtemp(prefix="lphq_pipe_")
    for k, m, bits, kind in [(31, 20, 64, 0), (63, 20, 128, 0), (31, 20, 64, 1), (21, 8, 64, 0), (31, 31, 64, 0), (9, 7, 64, 1), (47, 17, 128, 1)]:
        recs = random_records(199 + k, 60, 400, 30000) if k > 12 else dedup_records(random_records(5, 300, 12, 24), k)
        assert len(recs) >= 6
        path = os.path.join(tmp, f"t_{k}_{m}_{kind}.lph")
        try:
            L.build(recs, k, m, path, kind=kind, kmer_bits=bits, threads=8)
        except RuntimeError as e:
            if "no pilot below" in str(e):
                continue
            raise
        o = orc.Index(path, kind, bits)
        h = (L.Mphf if kind == 0 else L.MphfAlt).load(path, kmer_bits=bits, device=0)
        batch = [b"", recs[0], recs[1][: k - 1], b"", recs[2][:k], recs[3]] + recs[4:] + random_records(5, 30, 1, 200) + [b"", b"ACGTNNNNACGT" * 40]
        batch += random_records(11, 150, 0, 5) + [recs[0][:k + 2]] + random_records(12, 60, k, k + 20)  # > 64 and ~20 records per tile
        lens = np.array([len(r) for r in batch], dtype=np.int64)
        seq_off = np.concatenate([[0], np.cumsum(lens)])
        nwin = np.maximum(lens - k + 1, 0)
        out_off = np.concatenate([[0], np.cumsum(nwin)])
        flat = np.frombuffer(b"".join(batch), dtype=np.uint8)
        d_so, d_oo = torch.from_numpy(seq_off).to(dev), torch.from_numpy(out_off).to(dev)
        for mis in (0, 3, 15):
            buf = torch.zeros(len(flat) + 64, dtype=torch.uint8, device=dev)
            buf[mis:mis + len(flat)] = torch.from_numpy(flat.copy()).to(dev)
            d_out = torch.full((int(out_off[-1]) + 1,), -7, dtype=torch.int64, device=dev)
            for _ in range(2):  # twice: persistent workgroups must leave nothing behind
                h.query_records(buf.data_ptr() + mis, len(flat), d_so.data_ptr(), len(batch), d_out.data_ptr(), d_oo.data_ptr(), True)
            torch.cuda.synchronize()
            res = d_out.cpu().numpy().view(np.uint64)
            assert res[-1] == np.uint64(2**64 - 7)
            for i, r in enumerate(batch):
                got = res[out_off[i]:out_off[i + 1]]
                exp = o(r, False)
                bad = np.array([c not in b"ACGTUacgtu" for c in r], dtype=np.int64)
                csum = np.concatenate([[0], np.cumsum(bad)])
                valid = (csum[k:] - csum[:-k]) == 0 if len(r) >= k else np.zeros(0, dtype=bool)
                assert np.array_equal(got[valid], exp[valid]), (k, m, bits, kind, mis, i)
                assert np.all(got[~valid] == np.uint64(L.INVALID)), (k, m, bits, kind, mis, i)
    print("PIPE_OK")


if __name__ == "__main__":
    main()
