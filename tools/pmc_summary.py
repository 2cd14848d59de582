#!/usr/bin/env python
"""Per-kernel average of ONE counter from a rocprofv3 `--pmc X --output-format csv` pass (counter_collection.csv), as CSV.

    python tools/pmc_summary.py gpurun_out/pmc_r2_fetch FETCH_SIZE [out.csv]

FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE counts a wide coalesced streaming read (16 B per lane) at half its bytes
(MI355X_MICROARCH.md, HBM section): the `x2` column applies that correction; WRITE_SIZE is exact for 16-byte streaming stores."""
import collections
import csv
import glob
import os
import sys


def main():
    root, counter = sys.argv[1], sys.argv[2]
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                acc[(name, int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
    print("counter,kernel,grid_threads,dispatches,avg_counter_KB,avg_MB,avg_MB_x2_wide_read_correction", file=out)
    for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        a = sum(v) / len(v)
        print(f'{counter},"{name[:90]}",{grid},{len(v)},{a:.1f},{a / 1024:.2f},{2 * a / 1024:.2f}', file=out)


if __name__ == "__main__":
    main()
