#!/usr/bin/env python
"""Per-launch-shape statistics of ONE kernel from a rocprofv3 `--kernel-trace` rocpd database: the dispatches are grouped by grid
size (the shape of the launch), so that the average of the dominant shape is not blended with the other shapes of the same kernel.

    python tools/kernel_shape_stats.py gpurun_out/prof_x/x_results.db 'conv_igemm_kernel<64, 64, 32' [out.json] [--skip-first N]

For the mask-head 3x3 implicit GEMM (M = rois*196, N = 256, K = 2304) the tile count of a launch is grid / 256 threads; the
algorithmic FLOPs per launch follow from it (`--mask-gemm`: adds flops / achieved TFLOP/s per shape, M taken from the grid)."""
import collections
import json
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skip = 0
    if "--skip-first" in sys.argv:
        skip = int(sys.argv[sys.argv.index("--skip-first") + 1])
        args = [a for a in args if a != str(skip)]
    db, pat = args[0], args[1]
    c = sqlite3.connect(db)
    names = {r[0]: r[1] for r in c.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
    want = {k for k, n in names.items() if pat.replace(" ", "") in n.replace(" ", "")}
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    gx = "grid_size_x" if "grid_size_x" in cols else "grid_x"
    wx = "workgroup_size_x" if "workgroup_size_x" in cols else "workgroup_x"
    gy = gx.replace("_x", "_y")
    rows = c.execute(f"select kernel_id, start, end, {gx}, {gy}, {wx} from rocpd_kernel_dispatch order by start").fetchall()
    per = collections.defaultdict(list)
    for k, s, e, g, g2, w in rows:
        if k in want:
            per[(names[k], int(g), int(g2), int(w))].append(e - s)
    out = []
    for (name, g, g2, w), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        v = v[min(skip, len(v) - 1):] if skip else v
        sv = sorted(v)
        d = {"kernel": name, "grid_threads": [g, g2], "workgroup": w, "workgroups": g // max(w, 1) * max(g2, 1), "launches": len(v),
             "avg_us": round(sum(v) / len(v) / 1e3, 2), "median_us": round(sv[len(sv) // 2] / 1e3, 2), "min_us": round(sv[0] / 1e3, 2),
             "max_us": round(sv[-1] / 1e3, 2), "total_ms": round(sum(v) / 1e6, 3)}
        out.append(d)
    res = {"db": db, "pattern": pat, "shapes": out}
    txt = json.dumps(res, indent=1)
    if len(args) > 2:
        with open(args[2], "w") as fh:
            fh.write(txt + "\n")
    else:
        print(txt)


if __name__ == "__main__":
    main()
