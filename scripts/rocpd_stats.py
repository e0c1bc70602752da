#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME` writes
DIR/NAME_results.db): calls, total / average / min / max duration, share of GPU time -- the table
`rocprofv3 --stats` prints, as CSV on stdout (what is committed under profiles/).

    python scripts/rocpd_stats.py gpurun_out/prof/x_results.db [--skip-first N] > profiles/rNN_x_kernel_stats.csv
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = name.replace("void ", "").replace("q3::", "")
    return name.strip()


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0
    cur = db.cursor()
    rows = cur.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                       "on d.kernel_id = s.id order by d.start").fetchall()
    rows = rows[skip:]
    agg = {}
    for name, st, en in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = en - st
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'"{name}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{a[2]},{a[3]},{100.0 * a[1] / total:.2f}')


if __name__ == "__main__":
    main()
