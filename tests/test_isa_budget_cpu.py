"""Compile-time budget of the gfx950 kernels (runs without a GPU: hipcc cross-compiles).

Occupancy decides how much latency the query kernels can hide, and a silent register spill (scratch) or an LDS increase would
change it without failing any parity test.  This test compiles lphash_amd/csrc/kernels.hip and search.hip with
-Rpass-analysis=kernel-resource-usage and checks every kernel against the budget DESIGN.md / profiles/ assume."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lphash_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# kernel -> (max VGPRs, max LDS bytes); no scratch and no VGPR spill anywhere
BUDGET = {
    "k_query_packed": (64, 0),                # 8 waves / SIMD
    "k_query_tiles_stream": (64, 40 * 1024),
    # 128 threads, 8 positions per lane.  Round 4: 64 VGPRs and <= 10 KB of LDS for BOTH variants = 16 workgroups (32 waves, the CU's cap) per CU, no scratch
    # (one copy of the lookup code for the common and the overflow rounds: the duplicated copy was what cost the registers).
    "k_query_tiles_stream8": (64, 10240),
    "k_query_tiles_random": (64, 10 * 1024),  # + the 8-KiB queue of k-mer window starts (read batches); 16 workgroups per CU still fit
    "k_pack_kmers": (64, 0),
    "k_count_valid": (64, 0),
    "k_count_bad_bases": (64, 0),
    "k_gather_payload": (64, 0),
    "k_expand_runs": (64, 8 * 1024),   # lphq_expand_runs: 256 runs (code, where, inclusive length sum) per workgroup in LDS
    "k_compact_runs": (64, 1024),      # sessions: arenas -> one dense array (the 32 prefix sums per workgroup in LDS)
    "k_zero_run_counts": (64, 0),
    "k_extract_runs": (64, 40 * 1024),  # the streaming kernel's phases A-C + run starts (index construction, extract.hip)
    "k_assemble_runs": (64, 0),
    "k_group_flags": (64, 0),
    "k_group_scatter": (64, 0),
    "k_search": (64, 1024),
    "k_claim": (64, 0),
    "k_validate": (64, 0),
    "k_commit": (64, 0),
    "k_advance": (64, 0),
}
# SGPR spills go to VGPR lanes, not to memory.  k_query_packed takes the 1.2-KB header by value (measured 2 % faster than reading the header from the
# image); since round 3 the fallback function's descriptor is read from the image in the rare path that needs it and nothing spills.
SGPR_SPILLS = {"k_query_packed": 0}


_CACHE = {}
_ASM_DIR = None


def _compile(src, want_asm, sink):
    """one device compile of `src` with kernel-resource-usage remarks -> (usage per kernel, assembly text or None)"""
    import tempfile

    asm_path = None
    if want_asm:
        asm_path = os.path.join(_asm_dir(), f"{src}.{sink}.s")
    out_args = ["--cuda-device-only", "-S", "-o", asm_path] if asm_path else ["-c", "-o", os.devnull]
    if src == "stream8.hip":
        out_args = ["-DLPHQ_S8_KIND=0", f"-DLPHQ_S8_SINK={sink}"] + out_args  # (the partitioned half; the unpartitioned one is the same kernels with a smaller lookup)
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I.", "-Rpass-analysis=kernel-resource-usage", src] + out_args,
                       cwd=CSRC, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out, (open(asm_path).read() if asm_path else None)


def _asm_dir():
    global _ASM_DIR
    if _ASM_DIR is None:
        import tempfile

        _ASM_DIR = tempfile.mkdtemp(prefix="lphq_isa_")
    return _ASM_DIR


# every compile the tests of this module ask for: started together on first use (eight CPUs; one after the other they were 75 s of the CPU suite)
_WANTED = [("kernels.hip", True, 0), ("stream8.hip", True, 0), ("stream8.hip", False, 1), ("stream8.hip", False, 2), ("search.hip", False, 0), ("extract.hip", False, 0)]


def resource_usage(src, want_asm=False, sink=0):
    """kernel-resource-usage remarks of the device compile of `src` (cached per test session; all compiles of the module run concurrently on first use).
    want_asm: also return the gfx950 assembly.  sink: stream8.hip's -DLPHQ_S8_SINK (0 codes, 1 runs, 2 payload: one translation unit each)."""
    if not _CACHE:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=6) as ex:
            futs = {key: ex.submit(_compile, *key) for key in _WANTED}
        for key, f in futs.items():
            _CACHE[key] = f.result()
    key = (src, want_asm, sink)
    if key not in _CACHE:
        alt = (src, True, sink)  # (a compile that kept its assembly serves a caller that only wants the remarks)
        if alt in _CACHE:
            return _CACHE[alt] if want_asm else _CACHE[alt][0]
        _CACHE[key] = _compile(*key)
    return _CACHE[key] if want_asm else _CACHE[key][0]


@pytest.fixture(scope="module")
def kernels_asm():
    """gfx950 assembly of kernels.hip (one device-only compile, shared with the budget test of the same file)."""
    return resource_usage("kernels.hip", want_asm=True)


@pytest.fixture(scope="module")
def stream8_asm():
    """... and of stream8.hip (the eight-per-lane streaming kernel's translation unit)."""
    return resource_usage("stream8.hip", want_asm=True)


def _kernel_body(asm, mangled_fragment):
    """text of the first kernel whose symbol contains the fragment"""
    m = re.search(r"^(_Z\S*" + re.escape(mangled_fragment) + r"\S*):", asm, re.M)
    assert m, mangled_fragment
    end = asm.index(".Lfunc_end", m.end())
    return asm[m.start():end]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_stream8_variants_fit_their_occupancy_targets(stream8_asm):
    """The eight-positions-per-lane streaming kernel: every instantiation -- wide windows (P == 6: C4's W = 44) and narrow ones (P == 4: C3's W = 12) --
    within 64 VGPRs and 10240 B of LDS, i.e. 16 workgroups of two waves per CU (the hardware's 32-wave cap), and without a byte of scratch; its leader
    numbering is the DPP scan of gfx950/wave_scan.cuh and its plain-tile leader mask the v_cmp + v_addc pair of gfx950/shift_in.cuh; the wide variant exists
    once per (W - 8) % 8 and, with that remainder a compile-time constant, carries fewer scalar branches than the narrow one."""
    usage, asm = stream8_asm
    seen = 0
    for name, u in usage.items():
        if "k_query_tiles_stream8" not in name:
            continue
        seen += 1
        assert u["VGPRs"] <= 64 and u.get("LDS Size", 0) <= 10240 and u.get("ScratchSize", 0) == 0 and u.get("VGPRs Spill", 0) == 0, (name, u)
        assert u.get("Occupancy", 8) == 8, (name, u)
    assert seen == 2 * (8 + 1)  # 64- and 128-bit k-mers x (eight wide-window instantiations, one per (W - 8) % 8, + the narrow-window one)
    for frag in ("k_query_tiles_stream8ILi0ELb1ELi6ELi128ELb0ELi4E", "k_query_tiles_stream8ILi0ELb0ELi4ELi128ELb1ELin1E"):
        body = _kernel_body(asm, frag)
        for ctrl in ("row_shr:1", "row_shr:2", "row_shr:4", "row_shr:8", "row_bcast:15", "row_bcast:31"):
            assert ctrl in body, (frag, ctrl)
        assert len(re.findall(r"v_cmp_ne_u32[^\n]*\n\s*v_addc_co_u32", body)) >= 8, frag
        assert "scratch_" not in body


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("sink,name", [(1, "runs"), (2, "payload")])
def test_stream8_sink_variants_keep_the_occupancy(sink, name):
    """The other two sinks of the streaming kernel (round 6: lphq_query_records_runs, lphq_query_records_gather): the wide-window variants stay within 64 VGPRs
    and 10240 B of LDS without scratch (16 workgroups per CU, like the codes sink); the narrow-window ones may take a few registers more (the runs sink: 68 VGPRs,
    7 waves per SIMD -- forcing 64 spills) but never scratch."""
    usage = resource_usage("stream8.hip", sink=sink)
    seen = 0
    for kname, u in usage.items():
        if "k_query_tiles_stream8" not in kname:
            continue
        seen += 1
        assert u.get("ScratchSize", 0) == 0 and u.get("VGPRs Spill", 0) == 0 and u.get("LDS Size", 0) <= 10240, (kname, u)
        wide = "ELi6ELi128ELb0E" in kname
        assert u["VGPRs"] <= (64 if wide else 72), (kname, u)
        assert u.get("Occupancy", 8) >= (8 if wide else 7), (kname, u)
    assert seen == 2 * (8 + 1), name


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_minimizer_scan_uses_the_exec_masked_take_min(kernels_asm):
    """gfx950/take_min.cuh is inline assembly the compiler cannot see into: the random kernel at k = 63 (C4) must contain its sequence -- compare, EXEC
    narrowed to the lanes that take the new value, one 64-bit and one 32-bit move, EXEC restored, one wait state (the scan is a loop: the sequence appears once per copy of its body)."""
    _, asm = kernels_asm
    body = _kernel_body(asm, "k_query_packedILi0ELb1ELi6E")
    seq = re.findall(r"v_cmp_lt_u64[^\n]*\n\s*s_and_saveexec_b64[^\n]*\n\s*v_mov_b64[^\n]*\n\s*v_mov_b32[^\n]*\n\s*s_mov_b64 exec,[^\n]*\n\s*s_nop 0", body)
    assert len(seq) >= 1, len(seq)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["kernels.hip", "stream8.hip", "search.hip", "extract.hip"])
def test_kernels_fit_their_register_and_lds_budget(src):
    usage = resource_usage(src)
    assert usage, "no kernel-resource-usage remarks in the compiler output"
    seen = set()
    for name, u in usage.items():
        if "rocprim" in name:  # the library sort's kernels (extract.hip)
            continue
        m = re.search(r"(\d+)k_", name)  # Itanium mangling: <length><identifier>
        assert m, name
        key = name[m.end(1):m.end(1) + int(m.group(1))]
        assert key in BUDGET, f"kernel {key} has no budget entry"
        seen.add(key)
        vg, lds = BUDGET[key]
        assert u.get("ScratchSize", 0) == 0 and u.get("VGPRs Spill", 0) == 0, (name, u)
        assert u.get("SGPRs Spill", 0) <= SGPR_SPILLS.get(key, 0), (name, u)
        assert u["VGPRs"] + u.get("AGPRs", 0) <= vg, (name, u)
        assert u.get("LDS Size", 0) <= lds, (name, u)
    where = {"k_query_tiles_stream8": "stream8.hip", "k_search": "search.hip", "k_claim": "search.hip", "k_validate": "search.hip", "k_commit": "search.hip", "k_advance": "search.hip",
             "k_extract_runs": "extract.hip", "k_assemble_runs": "extract.hip", "k_group_flags": "extract.hip", "k_group_scatter": "extract.hip"}
    expected = {k for k in BUDGET if where.get(k, "kernels.hip") == src}
    assert seen == expected, (seen, expected)


def test_the_committed_isa_listing_is_of_this_tree():
    """profiles/isa_rest.json (tools/isa_rest_listing.py --json: the VALU instructions of k_query_packed by step, what bench.py's floor_instr_per_unit cites) was
    taken on the device sources in this tree -- it is regenerated offline (hipcc, no GPU), so there is no reason for it to lag -- and adds up: the listed total is
    within 1 % of the counter pass of the same tree where profiles/pmc_traffic.json has one."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    from bench import device_source_sha

    ij = json.load(open(os.path.join(ROOT, "profiles", "isa_rest.json")))
    assert ij["tree"] == device_source_sha(), "device sources changed: run `python tools/isa_rest_listing.py --json > profiles/isa_rest.json` (and refresh profiles/r05_isa_rest.txt)"
    tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for tkey, pkey in (("0,false,4", "random|p|k31|m20|n100000000|b28"), ("1,false,4", "random|u|k31|m20|n100000000|b28"), ("0,true,6", "random|p|k63|m20|n2000000000|b28")):
        e = ij[tkey]
        assert e["scan_instr_per_unit"] == 29 * e["W"] and abs(e["listed_instr_per_unit"] - e["scan_instr_per_unit"] - e["rest_instr_per_unit_listed"]) < 0.2
        if tj.get(pkey, {}).get("tree") == ij["tree"]:
            assert abs(e["listed_instr_per_unit"] / tj[pkey]["valu_lane_instr_per_unit"] - 1.0) < 0.01, (tkey, e["listed_instr_per_unit"], tj[pkey]["valu_lane_instr_per_unit"])
