#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 PMC passes of `bench.py` (FETCH_SIZE and WRITE_SIZE, separate passes as
MI355X_MICROARCH.md's HBM section prescribes; counters are KB; FETCH_SIZE x 2 = the gfx950 correction for wide 16-byte-per-lane
coalesced reads) -> the two JSON summaries committed under profiles/:

    python tools/traffic_json.py gpurun_out/pmc_r3_fetch gpurun_out/pmc_r3_write profiles/r03

writes <prefix>_dominant_kernel_traffic.json (mask-head 3x3 implicit GEMM, by launch shape) and <prefix>_hbm_class_traffic.json
(memory read + fusion, memory write, un-projection kernels)."""
import collections
import csv
import glob
import json
import os
import sys


def load(root, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                acc[(row["Kernel_Name"], int(row["Grid_Size"]))].append(float(row["Counter_Value"]) * 1024.0)
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    prefix = sys.argv[3]
    avg = lambda v: sum(v) / len(v)
    # ---- dominant kernel: conv_igemm_kernel<64,64,32,false,false>, by grid (= launch capacity: detection pass / proposal pass)
    shapes = {}
    for (name, grid), v in fetch.items():
        if "conv_igemm_kernel<64, 64, 32, false, false" not in name.replace("(anonymous namespace)::", ""):
            continue
        w = write.get((name, grid), [0.0])
        shapes[grid] = {"grid_threads": grid, "workgroups": grid // 256, "dispatches": [len(v), len(w)], "fetch_bytes_raw": round(avg(v), 1),
                        "fetch_bytes_corrected": round(2 * avg(v), 1), "write_bytes": round(avg(w), 1),
                        "traffic_bytes_per_launch": round(2 * avg(v) + avg(w), 1)}
    det = shapes.get(3676 * 256)
    prop = shapes.get(1568 * 256)
    out = {"kernel": "conv_igemm_kernel<64,64,32> mask_fcn 3x3 (M = rois * 196, N = 256, K = 2304)",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `bench.py --steps 6 --warmup 3 --no-variants "
                     "--no-cpu-baseline --no-kernel-events`; counters are KB; FETCH_SIZE x2 = the gfx950 correction for wide (16 B per "
                     "lane) coalesced reads of MI355X_MICROARCH.md's HBM section; the launches run in the default multi-stream schedule",
           "detection_pass_launch": det, "proposal_pass_launch": prop,
           "note": "the grid is the launch's capacity (300 / 128 ROIs); the ROIs really computed are the frame's distinct detection "
                   "boxes / memory instances, here taken from the launches' own WRITE_SIZE (rois_mean = write bytes / (196 * 256 * 4)): "
                   "algorithmic bytes per launch = rois * 196 * 256 * 4 * 2 + 256 * 2304 * 4"}
    for key in ("detection_pass_launch", "proposal_pass_launch"):
        if out[key]:
            # the ROIs these launches really computed: every ROI writes 196 x 256 fp32 outputs, so the WRITE_SIZE counter gives the
            # mean count of the profiled frames themselves (the short profiling run is not the bench's 60-frame average)
            rois = round(out[key]["write_bytes"] / (196 * 256 * 4), 1)
            alg = rois * 196 * 256 * 4 * 2 + 256 * 2304 * 4
            out[key]["rois_mean"] = rois
            out[key]["algorithmic_bytes_per_launch"] = round(alg, 1)
            out[key]["traffic_over_algorithmic"] = round(out[key]["traffic_bytes_per_launch"] / alg, 2)
    if det:
        out["traffic_bytes_per_launch"] = det["traffic_bytes_per_launch"]
    with open(prefix + "_dominant_kernel_traffic.json", "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    # ---- HBM class
    cls = {}
    for key in ("gather_pool_kernel", "project_fuse_kernel", "normalize_dirty_f16_kernel", "mw_cover_kernel", "mw_scatter_kernel",
                "mw_commit_kernel", "unproject_kernel"):
        fv = [x for (n, g), v in fetch.items() if key in n for x in v]
        wv = [x for (n, g), v in write.items() if key in n for x in v]
        if fv:
            cls[key] = {"dispatches": [len(fv), len(wv)], "fetch_bytes_raw": round(avg(fv), 1), "fetch_bytes_x2": round(2 * avg(fv), 1),
                        "write_bytes": round(avg(wv), 1) if wv else 0.0}
    tot = lambda keys: sum(cls[k]["fetch_bytes_x2"] + cls[k]["write_bytes"] for k in keys if k in cls)
    read = tot(("gather_pool_kernel", "project_fuse_kernel", "normalize_dirty_f16_kernel"))
    out2 = {"class": "memory read + fusion (a4 + a8), memory write (a16-a19), un-projection (a1 + a2); config B (640x640, N = 40 000)",
            "source": out["source"], "kernels": cls,
            "traffic_bytes_total_x2_reads": round(read, 1), "traffic_bytes_memory_write": round(tot(("mw_cover_kernel", "mw_scatter_kernel", "mw_commit_kernel")), 1),
            "algorithmic_bytes_survey_8d": 102494464,
            "note": "traffic_bytes_total_x2_reads = gather + project/fuse + the stand-alone incremental normalise on the frame's rows "
                    "(the product's write-through snapshot does that work inside mw_commit_kernel)"}
    with open(prefix + "_hbm_class_traffic.json", "w") as fh:
        json.dump(out2, fh, indent=1)
        fh.write("\n")
    print(json.dumps({"dominant": out.get("traffic_bytes_per_launch"), "hbm_read": out2["traffic_bytes_total_x2_reads"]}))


if __name__ == "__main__":
    main()
