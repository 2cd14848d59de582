#!/usr/bin/env python
"""HBM-class kernels (memory read + fusion, SURVEY §8 rows a4 + a8) out of a rocprofv3 --kernel-trace database of `bench.py`:
all launches, and the back-to-back launches of bench.py's `hbm_class_probe` (a launch whose predecessor on the timeline is the
same kernel), which is what `roofline_hbm` in the bench line prices.

    python tools/hbm_class_profile.py gpurun_out/prof_x/x_results.db [out.json]
"""
import json
import sqlite3
import sys

import numpy as np


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    out = {}
    for key in ("gather_pool_kernel", "project_fuse_kernel", "mw_obs_snapshot_kernel", "mw_obs_kernel", "normalize_dirty_f16_kernel"):
        mine = lambda n: key in n and (key != "mw_obs_kernel" or "snapshot" not in n)
        b2b, prev = [], None
        for n, s, e in rows:
            if mine(n) and prev is not None and mine(prev):
                b2b.append((e - s) / 1e3)
            prev = n
        allv = [(e - s) / 1e3 for n, s, e in rows if mine(n)]
        if not allv:
            continue
        out[key] = {"launches": len(allv), "avg_us": round(float(np.mean(allv)), 2), "min_us": round(float(np.min(allv)), 2),
                    "back_to_back_launches": len(b2b),
                    "back_to_back_median_us": round(float(np.median(b2b)), 2) if b2b else None,
                    "back_to_back_avg_us": round(float(np.mean(b2b)), 2) if b2b else None}
    text = json.dumps({"source": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-variants",
                       "note": "avg_us mixes the launches inside the frames (sharing the chip with the look-ahead trunk and the "
                               "previous frame's detection pass) with the probe's back-to-back launches; the probe figures are the "
                               "ones `roofline_hbm` is computed from",
                       "kernels": out}, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    else:
        print(text)


if __name__ == "__main__":
    main()
