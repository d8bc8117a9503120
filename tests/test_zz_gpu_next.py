"""GPU tests of the SURVEY section 8(f) "next" components (index construction on the GPU, opt-in kernels).

Kept in a file that sorts after test_gpu_parity.py so that `pytest -x` reaches every query-path test (section 8 a-e) first."""
import os

import pytest

pytestmark = pytest.mark.gpu

import lphash_amd as L  # noqa: E402
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
    file the sequential first-fit search writes."""
    _, recs = synthetic_records(seed=11 + k, n_records=nrec, length=rlen)
    recs = dedup_records(recs, k) if k < 25 else recs
    pc, pg = str(tmp_path / "cpu.lph"), str(tmp_path / "gpu.lph")
    sc = L.build(recs, k, m, pc, kind=kind, kmer_bits=bits, threads=8)
    sg = L.build(recs, k, m, pg, kind=kind, kmer_bits=bits, threads=8, device=0)
    assert sc == sg
    assert open(pc, "rb").read() == open(pg, "rb").read()


def test_pipelined_stream_kernel():
    """The opt-in software-pipelined streaming kernel (LPHQ_STREAM_PIPELINE=1, kernels_stream_pipe.cuh) gives the oracle's codes.
    Runs tests/pipe_stream_check.py in a child process because the launcher reads the switch once per process.  (Kept last in this
    file: the kernel was written while GPU access was closed, so this is its first run on hardware.)"""
    import subprocess
    import sys
    torch_dev()
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pipe_stream_check.py")
    # 1 workgroup per CU: every workgroup walks several tiles of the ~1 M-base batches (the double-buffered loop); 5 = the default
    for wgs, early in (("1", "1"), ("5", "0")):
        env = dict(os.environ, LPHQ_STREAM_PIPELINE="1", LPHQ_STREAM_PIPE_WGS=wgs, LPHQ_STREAM_PIPE_EARLY=early)
        r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "PIPE_OK" in r.stdout, (wgs, early, r.stdout[-2000:] + r.stderr[-4000:])
