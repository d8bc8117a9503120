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
