#!/usr/bin/env python
"""Per-kernel statistics (calls, total, average, min, max; percentage of the summed kernel time) from a rocprofv3
`--kernel-trace --stats` rocpd database, written as CSV: the summary that is committed under profiles/.

    python tools/kernel_stats.py gpurun_out/prof_x/x_results.db [out.csv] [--skip-first N]
"""
import collections
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skip = 0
    if "--skip-first" in sys.argv:
        skip = int(sys.argv[sys.argv.index("--skip-first") + 1])
        args = [a for a in args if a != str(skip)]
    db = args[0]
    out = open(args[1], "w") if len(args) > 1 else sys.stdout
    c = sqlite3.connect(db)
    names = {r[0]: r[1] for r in c.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
    rows = c.execute("select kernel_id, start, end from rocpd_kernel_dispatch order by start").fetchall()
    per = collections.defaultdict(list)
    for k, s, e in rows:
        per[names.get(k, str(k))].append(e - s)
    if skip:
        per = {k: v[min(skip, len(v) - 1):] for k, v in per.items()}
    total = sum(sum(v) for v in per.values())
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"', file=out)
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        name = k.replace("(anonymous namespace)::", "").replace('"', "'")
        print(f'"{name}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100.0 * sum(v) / total:.2f},{min(v)},{max(v)}', file=out)


if __name__ == "__main__":
    main()
