"""Readers of tests/golden/real/ (fixtures derived from the reference's bundled real data by tests/golden/make_real_golden.py) -- test side only."""
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real")
HEAD = 32_768  # codes per query set kept in full (the rest is pinned by SHA-256)

# name -> (kind, k, m, kmer_bits, unitig set).  The k=31 unitigs are a valid index set for every k >= 31 (no repeated 31-mer => no repeated k-mer).
CASES = {
    "se_k31_m20_b64_p": (0, 31, 20, 64, "ust_k31"),    # BASELINE.json configs[0] (C1) as configured; the .lph is committed
    "se_k31_m15_b64_p": (0, 31, 15, 64, "ust_k31"),    # README.md:175  build-p -k 31 -m 15
    "se_k31_m20_b64_u": (1, 31, 20, 64, "ust_k31"),    # the unpartitioned function over the same set (C5's kernel on real skew)
    "se_k63_m17_b128_p": (0, 63, 17, 128, "ust_k63"),  # README.md:183  build-p -k 63 -m 17
    "se_k63_m20_b128_p": (0, 63, 20, 128, "ust_k63"),  # C4's (k, m)
    "se_k47_m20_b128_u": (1, 47, 20, 128, "ust_k31"),  # README.md:191  build-u -k 47 -m 20
}

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_CODE = np.zeros(256, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
_VALID = np.zeros(256, dtype=bool)
for _c in b"ACGTUacgtu":  # constants::seq_nt4_table (src/constants.cpp:5-13): everything else maps to 4
    _VALID[_c] = True


def pack_2bit(seq: bytes):
    c = _CODE[np.frombuffer(seq, dtype=np.uint8)]
    c = np.concatenate([c, np.zeros((-len(c)) % 4, dtype=np.uint8)]).reshape(-1, 4)
    return (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8)


def unpack_2bit(packed, n):
    c = np.stack([(packed >> s) & 3 for s in (0, 2, 4, 6)], axis=1).reshape(-1)[:n]
    return _ACGT[c]


def unitig_records(z, which):
    lens = z[which + "_len"].astype(np.int64)
    flat = unpack_2bit(z[which + "_2bit"], int(lens.sum()))
    off = np.concatenate([[0], np.cumsum(lens)])
    return [flat[off[i]:off[i + 1]].tobytes() for i in range(len(lens))]


def query_sets(z):
    """name -> list of records (bytes)"""
    off = z["q_fastq_off"]
    fq = z["q_fastq"]
    return {"salmonella": [z["q_salmonella"].tobytes()],
            "ecoli": [z["q_ecoli_a"].tobytes(), z["q_ecoli_b"].tobytes()],
            "fastq": [fq[off[i]:off[i + 1]].tobytes() for i in range(len(off) - 1)]}


def valid_windows(rec: bytes, k):
    """bool per window: all k bases are ones seq_nt4_table maps below 4"""
    if len(rec) < k:
        return np.zeros(0, dtype=bool)
    bad = ~_VALID[np.frombuffer(rec, dtype=np.uint8)]
    cs = np.concatenate([[0], np.cumsum(bad)])
    return (cs[k:] - cs[:-k]) == 0


def load_inputs():
    return np.load(os.path.join(OUT, "inputs.npz"))
