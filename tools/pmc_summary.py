#!/usr/bin/env python3
"""Collapse rocprofv3 CSV output to per-kernel summaries of the apd:: kernels (small enough to commit).

usage: pmc_summary.py <rocprof_out_dir> <summary.csv>
"""
import collections
import csv
import glob
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    rows_out = []
    for path in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(list)
        meta = {}
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"]
                if "apd::" not in k:
                    continue
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
                meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"], r["SGPR_Count"])
        for (k, c), v in sorted(agg.items()):
            rows_out.append([k, c, len(v), sum(v) / len(v), min(v), max(v)] + list(meta[k]))
    for path in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
        agg = collections.defaultdict(list)
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"]
                if "apd::" not in k:
                    continue
                agg[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for k, v in sorted(agg.items()):
            rows_out.append([k, "duration_ns", len(v), sum(v) / len(v), min(v), max(v), "", "", "", "", "", ""])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean", "min", "max", "grid", "workgroup", "lds", "scratch", "vgpr", "sgpr"])
        w.writerows(rows_out)
    print("wrote", dst, len(rows_out), "rows")


if __name__ == "__main__":
    main()
