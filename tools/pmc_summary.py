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
            tail = v[2:] if len(v) > 2 else v  # drop the warm-up iteration's two launches (bench --warmup 1)
            rows_out.append([k, c, len(v), sum(v) / len(v), min(v), max(v)] + list(meta[k]) + [sum(tail) / len(tail)] +
                            [" ".join("%.4g" % x for x in v[:16]) if "k67" in k or "k910" in k else ""])
    for path in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
        agg = collections.defaultdict(list)
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"]
                if "apd::" not in k:
                    continue
                agg[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for k, v in sorted(agg.items()):
            tail = v[2:] if len(v) > 2 else v
            rows_out.append([k, "duration_ns", len(v), sum(v) / len(v), min(v), max(v), "", "", "", "", "", "", sum(tail) / len(tail),
                             " ".join("%.4g" % x for x in v[:16]) if "k67" in k or "k910" in k else ""])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean", "min", "max", "grid", "workgroup", "lds", "scratch", "vgpr", "sgpr", "mean_after_warmup", "per_dispatch"])
        w.writerows(rows_out)
    print("wrote", dst, len(rows_out), "rows")


def traffic_json(fetch_csv, write_csv, workload, dst, sq_csv=None, kernel_substr="k67"):
    """FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so the read
    side is doubled (MI355X_MICROARCH.md, HBM section).  Steady state = minimum over the dispatches (the
    first launches run on random planes and scatter more)."""
    import json

    def pick(path, counter):
        with open(path) as f:
            for r in csv.DictReader(f):
                if kernel_substr in r["kernel"] and r["counter"] == counter:
                    return float(r.get("mean_after_warmup") or r["mean"]), float(r["min"]), int(r["dispatches"])
        return None
    fe, wr = pick(fetch_csv, "FETCH_SIZE"), pick(write_csv, "WRITE_SIZE")
    rec = {"workload": workload, "kernel": kernel_substr, "dispatches": fe[2],
           "fetch_bytes_mean": fe[0] * 1024 * 2, "fetch_bytes_min": fe[1] * 1024 * 2,
           "write_bytes_mean": wr[0] * 1024, "write_bytes_min": wr[1] * 1024,
           "hbm_bytes_per_launch": fe[0] * 1024 * 2 + wr[0] * 1024,
           "note": "FETCH_SIZE x2 (gfx950), WRITE_SIZE uncorrected; mean over the timed-region dispatches (warm-up launches dropped); separate --pmc passes"}
    if sq_csv and os.path.exists(sq_csv):  # VALU issue accounting of the same kernel (its own --pmc pass)
        iv, du = pick(sq_csv, "SQ_INSTS_VALU"), pick(sq_csv, "duration_ns")
        if iv and du:
            rec["valu_insts_per_launch"] = iv[0]
            rec["sq_pass_launch_ns"] = du[0]
            # 1024 SIMDs; a wave64 VALU instruction occupies its SIMD's 16-lane pipe for 4 cycles (2.4 GHz)
            rec["valu_pipe_busy_frac"] = iv[0] * 4.0 / (1024 * 2.4 * du[0])
    with open(dst, "w") as f:
        json.dump(rec, f, indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    if sys.argv[1] == "--traffic":
        traffic_json(*sys.argv[2:7])
    else:
        main()
