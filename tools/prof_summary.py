#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace as text: per-kernel calls / total / mean / min / max.

usage: python tools/prof_summary.py gpurun_out/<dir>/<name>_results.db > profiles/<name>.txt
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main(path: str):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start), d.grid_size_x, d.workgroup_size_x, d.group_segment_size, d.private_segment_size "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"{'kernel':110s} {'calls':>6s} {'total_us':>12s} {'mean_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} {'grid':>10s} {'wg':>5s} {'lds':>7s} {'scratch':>7s}")
    for n, calls, tot, avg, mn, mx, grid, wg, lds, scr in rows:
        print(f"{short(n):110s} {calls:6d} {tot/1e3:12.1f} {avg/1e3:10.3f} {mn/1e3:10.3f} {mx/1e3:10.3f} {100*tot/total:6.2f} {grid:10d} {wg:5d} {lds:7d} {scr:7d}")


if __name__ == "__main__":
    main(sys.argv[1])
