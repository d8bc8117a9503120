#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace the way `--stats` prints it, plus resources.

usage: python tools/rocpd_summary.py <results.db> [<out.txt>]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), "
        "max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["KERNEL_DISPATCH stats (durations in us; from rocprofv3 --kernel-trace --stats, rocpd sqlite)",
             f"{'calls':>6} {'total_us':>12} {'avg_us':>12} {'min_us':>12} {'max_us':>12} {'pct':>6} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7} {'scratch':>7} {'grid_x':>10} {'wg_x':>5}  name"]
    for name, n, tot, avg, mn, mx, vg, ag, sg, lds, scr, gx, wx in rows:
        short = name if len(name) < 160 else name[:157] + "..."
        lines.append(f"{n:>6} {tot/1e3:>12.1f} {avg/1e3:>12.1f} {mn/1e3:>12.1f} {mx/1e3:>12.1f} {100*tot/total:>6.2f} {vg:>5} {ag:>5} {sg:>5} {lds:>7} {scr:>7} {gx:>10} {wx:>5}  {short}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
