#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the raw counter files of tools/round4_counters.sh.   usage: pmc_to_json.py <tag> <dir with pmc_<tag>_<wl>.txt>

Per workload: bytes per unit = (FETCH_SIZE x 2 + WRITE_SIZE) x 1024 / units (the gfx950 corrections of MI355X_MICROARCH.md's HBM / rocprofv3 section:
FETCH_SIZE counts 128-byte requests in units of 64 B there), VALU lane-instructions per unit = SQ_INSTS_VALU x 64 / units, L2 misses per unit, and the
sha of the device sources the library was built from (bench.py withholds an entry whose sha is not the tree's)."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import device_source_sha  # noqa: E402

tag, d = sys.argv[1], sys.argv[2]
# the streaming kernel's last template argument is its sink (stream8_device.cuh: 0 codes, 1 runs, 2 payload); a run of a sink workload also launches the codes
# kernel once (its check pass), so the pattern names the sink
S8 = r"k_query_tiles_stream8<[^>]*, %d>"
WL = {  # raw file suffix -> (key of bench.py, units per launch, kernel name pattern)
    "c2": ("random|p|k31|m20|n100000000|b28", 1 << 28, "k_query_packed"),
    "c3": ("stream|p|k31|m20|n100000000|b0", 100_000_000, S8 % 0),
    "c5": ("random|u|k31|m20|n100000000|b28", 1 << 28, "k_query_packed"),
    "c4r": ("random|p|k63|m20|n2000000000|b28", 1 << 28, "k_query_packed"),
    "c4s": ("stream|p|k63|m20|n2000000000|b0", 2_000_000_000, S8 % 0),
    "c3runs": ("stream|p|k31|m20|n100000000|b0|runs", 100_000_000, S8 % 1),
    "c4sruns": ("stream|p|k63|m20|n2000000000|b0|runs", 2_000_000_000, S8 % 1),
    "c4sg1": ("stream|p|k63|m20|n2000000000|b0|gather1", 2_000_000_000, S8 % 2),
    "c2nm": ("random|p|k31|m20|n100000000|b28|nonmember0.5", 1 << 28, "k_query_packed"),       # BASELINE.md section 3's secondary inputs
    "c3sh": ("stream|p|k31|m20|n100000000|b0|shuffled", 100_000_000, S8 % 0),
}
out = {"_comment": "HBM/fabric bytes per unit ((FETCH_SIZE x 2 + WRITE_SIZE) x 1024 / units, per MI355X_MICROARCH.md's HBM section) and VALU lane-instructions per "
                   "unit (SQ_INSTS_VALU x 64 / units: a wave64 instruction = 64 lane-instructions whatever the live lanes) from separate rocprofv3 --pmc passes "
                   f"(tools/round4_counters.sh): profiles/{tag}_pmc_<workload>_raw.txt; `tree` = sha of the device sources the passes were taken on "
                   "(bench.py: device_source_sha) -- an entry of another tree is withheld from the bench line"}
sha = device_source_sha()
for suf, (key, units, frag) in WL.items():
    path = os.path.join(d, f"pmc_{tag}_{suf}.txt")
    if not os.path.exists(path):
        continue
    c = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+n=\s*\d+\s+avg=\s*([0-9.]+)\s+(.*)", line)
        if m and re.search(frag, m.group(3)):
            c[m.group(1)] = float(m.group(2))
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    e = {"bytes_per_unit": round((c["FETCH_SIZE"] * 2 + c["WRITE_SIZE"]) * 1024 / units, 2), "source": f"profiles/{tag}_pmc_{suf}_raw.txt", "tree": sha}
    if "SQ_INSTS_VALU" in c:
        e["valu_lane_instr_per_unit"] = round(c["SQ_INSTS_VALU"] * 64 / units, 1)
        e["valu_source"] = f"profiles/{tag}_pmc_{suf}_raw.txt (SQ_INSTS_VALU)"
    if "TCC_MISS_sum" in c:
        e["l2_misses_per_unit"] = round(c["TCC_MISS_sum"] / units, 3)
    stats = os.path.join(d, f"{tag}_{suf}_kernel_stats.txt")  # rocprofv3 --kernel-trace --stats of the same command (tools/pmc_passes.sh): average duration of the kernel
    if os.path.exists(stats):
        for line in open(stats):
            f = line.split()
            if len(f) > 12 and re.search(frag, line) and f[0].isdigit():
                e["kernel_us_rocprof"] = float(f[2])
                e["kernel_us_rocprof_calls"] = int(f[0])
                e["kernel_stats_source"] = f"profiles/{tag}_{suf}_kernel_stats.txt"
                break
    out[key] = e
print(json.dumps(out, indent=1))
