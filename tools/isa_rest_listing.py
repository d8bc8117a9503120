#!/usr/bin/env python3
"""ISA listing of k_query_packed by SOURCE STEP (offline: hipcc + the assembly it emits; no GPU).

Compiles lphash_amd/csrc/kernels.hip for gfx950 with -gline-tables-only (line tables do not change the code: the instruction stream is compared with the
plain -O3 compile and must be identical), takes one instantiation of lphq::k_query_packed<KIND, K128, P>, splits it into basic blocks and attributes
every instruction to the device function its .loc line lies in (inlined callees keep their own lines).  Output: per basic block the loop depth and the
VALU / SALU / VMEM / SMEM counts by source function, and the totals per function for (a) the m-mer scan loop, (b) the straight-line code of a lookup.

    python tools/isa_rest_listing.py 0,true,6      # C4: partitioned, 128-bit k-mers, P = 6
    python tools/isa_rest_listing.py 0,false,4     # C2
    python tools/isa_rest_listing.py 1,false,4     # C5
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lphash_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def compile_asm(path, debug):
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I.", "--cuda-device-only", "-S", "-o", path, "kernels.hip"]
    if debug:
        cmd.insert(1, "-gline-tables-only")
    subprocess.run(cmd, cwd=CSRC, check=True, capture_output=True)


def function_body(asm, mangled):
    out, on = [], False
    for ln in asm:
        if ln.startswith(mangled + ":"):
            on = True
            continue
        if on and ln.startswith(".Lfunc_end"):
            break
        if on:
            out.append(ln.rstrip("\n"))
    return out


def klass(mn):
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if mn.startswith("ds_"):
        return "LDS"
    if mn.startswith(("s_load", "s_buffer_load")):
        return "SMEM"
    if mn.startswith("v_"):
        return "VALU"
    if mn.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep", "s_setprio")):
        return "WAIT"
    if mn.startswith(("s_cbranch", "s_branch")):
        return "BRANCH"
    if mn.startswith("s_"):
        return "SALU"
    return "OTHER"


def source_functions():
    """(file basename, first line, last line, name) of every function defined in the device sources (brace matching from its definition line)."""
    out = []
    for sub in ("", "gfx950"):
        d = os.path.join(CSRC, sub)
        for f in sorted(os.listdir(d)):
            if not f.endswith((".cuh", ".hip")):
                continue
            lines = open(os.path.join(d, f)).read().split("\n")
            i = 0
            while i < len(lines):
                m = re.search(r"(?:__device__|__global__)[^;{]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", re.sub(r"__launch_bounds__\([^)]*\)", "", lines[i]))
                if m and "(" in lines[i]:
                    j, depth, seen = i, 0, False
                    while j < len(lines):
                        depth += lines[j].count("{") - lines[j].count("}")
                        seen = seen or "{" in lines[j]
                        if seen and depth <= 0:
                            break
                        if not seen and ";" in lines[j]:
                            break
                        j += 1
                    if seen:
                        out.append((f, i + 1, j + 1, m.group(1)))
                        i += 1  # (nested lambdas / the next definition: keep scanning inside)
                        continue
                i += 1
    return out


# How often a wave runs each basic block per lookup (one lane = one lookup, so "per lookup" = per wave-iteration).  Derived by reading the listing (--dump);
# `valu` is the block's static VALU count the weight was derived for: a compiler or source change that moves it makes the tool stop instead of mis-adding.
#   W            the m-mer loop: one trip per m-mer of the k-mer (k - m + 1)
#   p_tail       1 - alpha^64: a wave holds a lane whose table position is a free slot (alpha = 0.94: ~0.98)
#   p_fb         1 - (1 - f)^64: a wave holds a k-mer of a colliding minimizer (f = fallback k-mers / k-mers of the index)
MODEL = {
    "0,true,6": {"config": "C4 random: k=63 m=20 (W = 44), partitioned, 128-bit k-mers, 32 minimizers per block", "W": 44, "f_fallback": 0.0014, "steps": [
        ("k-mer load, grid-stride loop, code store", [(".LBB4_2", 3, 1), (".LBB4_3", 2, 1)]),
        ("minimizer_of outside the hash loop: left-align the 128-bit k-mer (5), set up (5), the 64-bit window of each of the 4 groups of 13 m-mers (1 + 5 + 5 + 2: a move, "
         "and a 128-bit funnel shift while the window straddles the two words), cut the minimizer out at its position (12: both arms of the >= 64 / < 64 shift run under EXEC)",
         [(".LBB4_16", 5, 1), (".LBB4_17", 1, 0), (".LBB4_18", 1, 0), (".LBB4_6", 5, 1), (".LBB4_8", 2, "groups:1+1+1+2"), (".LBB4_11", 4, "groups:0+1+1+0"), (".LBB4_19", 12, 1)]),  # (_17 / _18: the uniform arms for k <= 32 in a 128-bit k-mer and for an empty scan)
        ("m-mer loop: cut 2 + MurmurHash2_64 24 + leftmost minimum 3", [(".LBB4_15", 29, "W")]),
        ("PTHash position of the minimizer: Murmur 24, bucket (skew test, exact mod by the dense / sparse bucket count: div_step 19) and codeword fetch + field extract (phf_load_code 30), "
         "hashed-pilot fetch (phf_load_hp 14), x ^ hp mod table size (div_step 19 + mod_u64 9), tail test 4, 2 of minimizer_of scheduled here", [(".LBB4_25", 121, 1)]),
        ("tail table entry of the lanes on a free slot", [(".LBB4_27", 3, "p_tail")]),
        ("fused 128-B block: six 16-byte loads, type / size / position planes by masked popcounts, the four rank bases (uni_decode)", [(".LBB4_28", 76, 1), (".LBB4_31", 20, 1)]),
        ("closed form by type (ctx_partitioned, lookup_ctx, finish_hval): base +- p", [(".LBB4_34", 6, 1), (".LBB4_43", 3, 0), (".LBB4_45", 1, "p_fb"), (".LBB4_47", 1, 1), (".LBB4_48", 2, 1), (".LBB4_49", 2, 1)]),
        ("fallback function for k-mers of colliding minimizers: Murmur of both k-mer words 44 + a second PTHash position", [(".LBB4_38", 145, "p_fb"), (".LBB4_42", 1, "p_fb"), (".LBB4_44", 1, 0)]),
    ]},
    "0,false,4": {"config": "C2 random: k=31 m=20 (W = 12), partitioned, 64-bit k-mers, 64 minimizers per block", "W": 12, "f_fallback": 0.000075, "steps": [
        ("k-mer load, grid-stride loop, code store; window set-up of minimizer_of (4)", [(".LBB7_2", 3, 1), (".LBB7_3", 5, 1)]),
        ("m-mer loop: cut 2 + MurmurHash2_64 24 + leftmost minimum 3", [(".LBB7_5", 29, "W")]),
        ("PTHash position of the minimizer (as above) + 6 of minimizer_of (cut the minimizer out at its position)", [(".LBB7_6", 125, 1)]),
        ("tail table entry of the lanes on a free slot", [(".LBB7_8", 3, "p_tail")]),
        ("fused 128-B block, 64 minimizers: loads, planes by masked popcounts (low_mask 10, popcounts 9), rank bases", [(".LBB7_9", 91, 1), (".LBB7_12", 17, 1)]),
        ("closed form by type", [(".LBB7_15", 6, 1), (".LBB7_24", 3, 0), (".LBB7_26", 1, "p_fb"), (".LBB7_28", 1, 1), (".LBB7_29", 2, 1), (".LBB7_30", 2, 1)]),
        ("fallback function (64-bit k-mer: one Murmur)", [(".LBB7_19", 121, "p_fb"), (".LBB7_23", 1, "p_fb"), (".LBB7_25", 1, 0)]),
    ]},
    "1,false,4": {"config": "C5 random: k=31 m=20 (W = 12), unpartitioned, 64-bit k-mers, pair blocks", "W": 12, "f_fallback": 0.000075, "steps": [
        ("k-mer load, grid-stride loop, code store; window set-up of minimizer_of (4)", [(".LBB11_2", 3, 1), (".LBB11_3", 5, 1)]),
        ("m-mer loop: cut 2 + MurmurHash2_64 24 + leftmost minimum 3", [(".LBB11_5", 29, "W")]),
        ("PTHash position of the minimizer + 6 of minimizer_of", [(".LBB11_6", 126, 1)]),
        ("tail table entry of the lanes on a free slot", [(".LBB11_8", 3, "p_tail")]),
        ("64-B pair block: prefix + size / position planes (pair_decode 35), closed form (ctx_unpartitioned 8)", [(".LBB11_9", 54, 1)]),
        ("closed form", [(".LBB11_18", 3, 0), (".LBB11_20", 1, "p_fb"), (".LBB11_22", 1, 1), (".LBB11_23", 2, 1), (".LBB11_24", 2, 1)]),
        ("fallback function", [(".LBB11_13", 122, "p_fb"), (".LBB11_17", 1, "p_fb"), (".LBB11_19", 1, 0)]),
    ]},
}


def weight_of(w, model):
    if isinstance(w, (int, float)):
        return float(w)
    if w == "W":
        return float(model["W"])
    if w == "p_tail":
        return 1.0 - 0.94 ** 64
    if w == "p_fb":
        return 1.0 - (1.0 - model["f_fallback"]) ** 64
    if w.startswith("groups:"):
        return float(sum(int(x) for x in w[7:].split("+")))
    raise ValueError(w)


def dynamic_account(targs, blocks):
    """VALU instructions per lookup by step from the static block counts and MODEL's trip counts; returns (rows, total, scan, rest)."""
    model = MODEL[targs]
    static = {b["label"]: sum(1 for _, k_, _, _ in b["ins"] if k_ == "VALU") for b in blocks}
    rows, total, used = [], 0.0, set()
    for name, parts in model["steps"]:
        dyn = 0.0
        for label, valu, w in parts:
            assert static.get(label) == valu, f"{targs} {label}: the listing has {static.get(label)} VALU, the model was derived for {valu}: re-derive MODEL (--dump)"
            dyn += valu * weight_of(w, model)
            used.add(label)
        rows.append((name, dyn))
        total += dyn
    skipped = {lb: v for lb, v in static.items() if lb not in used and v and lb != "entry"}
    assert not skipped, f"{targs}: blocks with VALU instructions that the model does not place: {skipped}"
    scan = 29.0 * model["W"]
    return rows, total, scan, total - scan


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--json":  # all three kernels: profiles/isa_rest.json (what bench.py's roofline_valu cites)
        import json

        sys.path.insert(0, ROOT)
        from bench import device_source_sha

        out = {"_comment": "VALU instructions per lookup of lphq::k_query_packed by step, from the compiled gfx950 listing (static counts per basic block x trip counts, "
                           "tools/isa_rest_listing.py; the readable form is profiles/r05_isa_rest.txt); `tree` = sha of the device sources the listing was taken on",
               "tree": device_source_sha()}
        for targs in MODEL:
            blocks = listing(targs)[0]
            rows, total, scan, rest = dynamic_account(targs, blocks)
            out[targs] = {"config": MODEL[targs]["config"], "W": MODEL[targs]["W"], "listed_instr_per_unit": round(total, 1), "scan_instr_per_unit": scan,
                          "hash_floor_instr_per_unit": 24 * MODEL[targs]["W"], "rest_instr_per_unit_listed": round(rest, 1),
                          "floor_instr_per_unit_listed": round(24 * MODEL[targs]["W"] + rest, 1), "steps": [[n, round(v, 1)] for n, v in rows]}
        print(json.dumps(out, indent=1))
        return
    targs = sys.argv[1] if len(sys.argv) > 1 else "0,true,6"
    blocks, mangled = listing(targs)
    report(targs, blocks, mangled)


def listing(targs):
    kind, k128, p = targs.split(",")
    mangled = f"_ZN4lphq14k_query_packedILi{kind}ELb{1 if k128 == 'true' else 0}ELi{p}EEEv17lphq_image_headerPKhPKmmPm"
    with tempfile.TemporaryDirectory() as d:
        compile_asm(os.path.join(d, "g.s"), True)
        compile_asm(os.path.join(d, "n.s"), False)
        g = open(os.path.join(d, "g.s")).readlines()
        n = open(os.path.join(d, "n.s")).readlines()
    files = {}
    for ln in g:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    body = function_body(g, mangled)
    plain = function_body(n, mangled)
    ins = lambda b: [x.split(";")[0].strip() for x in b if x.startswith("\t") and not x.strip().startswith((".", ";")) and x.split(";")[0].strip()]
    assert ins(body) == ins(plain), "line tables changed the instruction stream"
    funcs = source_functions()

    def owner(fname, line):
        best = None
        for f, a, b, name in funcs:
            if f == fname and a <= line <= b and (best is None or a >= best[0]):
                best = (a, name)
        return best[1] if best else f"{fname}:{line}"

    blocks, cur, loc = [], {"label": "entry", "depth": 0, "ins": []}, ("?", 0)
    after_label = False
    for ln in body:
        m = re.match(r"(\.LBB\d+_\d+):\s*;?(.*)", ln)
        if m:
            blocks.append(cur)
            note = m.group(2)
            dm = re.search(r"Depth=(\d+)", note)
            cur = {"label": m.group(1), "depth": int(dm.group(1)) if dm else 0, "ins": [], "note": note.strip()}
            after_label = True
            continue
        if after_label and re.match(r"\s*;", ln):  # the assembler's loop annotation goes on over comment lines: "Parent Loop ... Depth=2", "=> This Inner Loop Header: Depth=3"
            dm = re.search(r"Depth=(\d+)", ln)
            if dm and "Child Loop" not in ln:
                cur["depth"] = int(dm.group(1))
            cur["note"] += " | " + ln.strip("; \t")
            continue
        after_label = False
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        if ln.startswith("\t") and not ln.strip().startswith((".", ";")):
            text = ln.split(";")[0].strip()
            if text:
                cur["ins"].append((text, klass(text.split()[0]), owner(*loc), loc))
    blocks.append(cur)
    return blocks, mangled


def report(targs, blocks, mangled):
    print(f"k_query_packed<{targs}>  ({mangled})")
    print(f"{'block':10} {'depth':>5} {'VALU':>5} {'SALU':>5} {'VMEM':>5} {'SMEM':>5}  by source function (VALU)")
    tot = collections.Counter()
    per_depth = collections.defaultdict(collections.Counter)
    for b in blocks:
        c = collections.Counter(k_ for _, k_, _, _ in b["ins"])
        by = collections.Counter(o for _, k_, o, _ in b["ins"] if k_ == "VALU")
        tot.update(c)
        for o, v in by.items():
            per_depth[b["depth"]][o] += v
        print(f"{b['label']:10} {b['depth']:>5} {c['VALU']:>5} {c['SALU']:>5} {c['VMEM']:>5} {c['SMEM']:>5}  " + ", ".join(f"{o} {v}" for o, v in by.most_common()))
    print("static totals:", dict(tot))
    for dpt in sorted(per_depth):
        print(f"-- VALU by source function, loop depth {dpt}: total {sum(per_depth[dpt].values())}")
        for o, v in per_depth[dpt].most_common():
            print(f"   {v:5d}  {o}")
    if targs in MODEL:
        rows, total, scan, rest = dynamic_account(targs, blocks)
        print(f"-- VALU instructions per lookup by step ({MODEL[targs]['config']})")
        for name, v in rows:
            print(f"   {v:8.1f}  {name}")
        print(f"   {total:8.1f}  total;  scan 29 W = {scan:.0f}, hash floor 24 W = {24 * MODEL[targs]['W']}, rest = {rest:.1f}, floor = 24 W + rest = {24 * MODEL[targs]['W'] + rest:.1f}")
    if "--dump" in sys.argv:
        for b in blocks:
            print(f"\n{b['label']} depth {b['depth']} {b.get('note', '')}")
            for text, k_, o, lc in b["ins"]:
                print(f"    {k_:6} {text:70} ; {o} ({lc[0]}:{lc[1]})")


if __name__ == "__main__":
    main()
