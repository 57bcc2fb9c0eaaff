#!/usr/bin/env python3
"""Counter profile of one bench.py command line, reduced to what bench.py's `roofline` object needs (runs on the GPU box).

    python tools/profile_bench.py <out_dir> [bench.py flags, e.g. --steps 20 --warmup 5 --workload ...]

Five rocprofv3 runs of the same command (bench.py <flags> --no-cpu-baseline), counters in their own passes as the guide
prescribes (kernel trace only; never combined with other trace domains):
    1. --kernel-trace --stats                                   launch durations, kernel_stats
    2. --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VMEM_RD
    3. --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum   (L1 tag look-ups, L1 -> L2 requests, L2 hits / misses)
    3b. --pmc SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64   (the launch's VALU instructions by class)
    4. --pmc FETCH_SIZE                                          (KiB; x2 on gfx950, MI355X_MICROARCH.md "HBM")
    5. --pmc WRITE_SIZE                                          (KiB)
For the sweep kernels (k67*, k910*) only the launches of the TIMED region are reduced: bench.py launches each of them twice
per iteration (black, red), warm-up first, so the timed launches are the last 2 x steps dispatches of the kernel.

Writes into <out_dir>:
    pmc_bench_<workload>_s<steps>_w<warmup>.json     what bench.py reads (per-launch means + the per-dispatch values)
    pmc_bench_<workload>_s<steps>_w<warmup>.csv      per-dispatch rows of every pass for the sweep kernels (committed evidence)
    kernel_stats_<workload>_s<steps>_w<warmup>.csv   rocprofv3 --stats table of pass 1
    bench_<workload>_s<steps>_w<warmup>_pass<k>.json the bench line of every pass
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ("trace", ["--kernel-trace", "--stats"]),
    ("sq", ["--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
            "SQ_WAIT_ANY", "SQ_WAVES", "SQ_INSTS_VMEM_RD"]),
    ("tcp", ["--kernel-trace", "--pmc", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"]),
    # the launch's VALU instructions by class, whole kernel (every basic block at its real execution count): what bench.py prices with the
    # measured issue costs for `valu_busy_estimate` (round 6; rounds 2-5 priced the static mix of ONE basic block)
    ("mix", ["--kernel-trace", "--pmc", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_TRANS_F32",
             "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64"]),
    ("fetch", ["--kernel-trace", "--pmc", "FETCH_SIZE"]),
    ("write", ["--kernel-trace", "--pmc", "WRITE_SIZE"]),
]
# optional diagnostic groups, selected with APD_PROFILE_PASSES=icache,lds,sq2 (then only these run and the summary is written
# as pmc_extra_<tag>.json with the mean of every counter over the timed launches)
EXTRA_PASSES = {
    "tcp": ["--kernel-trace", "--pmc", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"],
    "sq": ["--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVES",
           "SQ_INSTS_VMEM_RD"],
    "icache": ["--kernel-trace", "--pmc", "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE", "SQ_IFETCH",
               "SQ_IFETCH_LEVEL", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"],
    "lds": ["--kernel-trace", "--pmc", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS",
            "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL"],
    "sq2": ["--kernel-trace", "--pmc", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC",
            "SQ_ACTIVE_INST_ANY", "SQ_INSTS_BRANCH", "SQ_INST_CYCLES_SALU"],
    "sq3": ["--kernel-trace", "--pmc", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32",
            "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_INT32", "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_VMEM"],
}
SWEEP_KERNELS = {"k67": "k67", "k910": "k910_update_weak"}
# APD_PROFILE_PASS_KEY=<key of bench.PASS_WORKLOADS>: the whole-pass sub-line of that key instead of a sweep.  The command becomes
# bench.py --steps 1 --warmup 0 --only-workloads <sweep key of the workload> --only-workloads <key>; K14 / K15 launch once per pass,
# so the timed launches are the last `passes` dispatches of each; output pmc_pass_<workload>_p<passes>.json / .csv.
PASS_KERNELS = {"k14": "k14w_depth_to_weak", "k15": "k15w_local_refine"}


def flag(flags, name, default):
    return type(default)(flags[flags.index(name) + 1]) if name in flags else default


def main():
    out_dir, flags = sys.argv[1], sys.argv[2:]
    steps, warmup = flag(flags, "--steps", 6), flag(flags, "--warmup", 1)
    workload = flag(flags, "--workload", "eth3d_office_fullres_8src")
    tag = "%s_s%d_w%d" % (workload, steps, warmup)
    pass_key = os.environ.get("APD_PROFILE_PASS_KEY")
    kernels_of_interest, tail = SWEEP_KERNELS, ["--no-cpu-baseline", "--no-workloads"]
    n_timed = 2 * steps
    if pass_key:
        sys.path.insert(0, ROOT)
        import bench
        _, workload, n_timed, _, pass_kind = [e for e in bench.PASS_WORKLOADS if e[0] == pass_key][0]
        sweep_key = [e[0] for e in bench.SUB_WORKLOADS if e[1] == workload and e[5] == 1 and e[2] == bench.PASS_ITERATIONS][0]
        flags = ["--steps", "1", "--warmup", "0"]   # the headline before the sub-lines: one iteration of the default workload
        tail = ["--no-cpu-baseline", "--only-workloads", sweep_key, "--only-workloads", pass_key]
        kernels_of_interest = PASS_KERNELS
        tag = "%s_p%d%s" % (workload, n_timed, "" if pass_kind == "photometric" else "_" + pass_kind)
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    per_dispatch = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel key -> counter -> [values in launch order]
    meta = {}
    rows_csv = []
    extra = [e for e in os.environ.get("APD_PROFILE_PASSES", "").split(",") if e]
    passes = [(e, EXTRA_PASSES[e]) for e in extra] if extra else PASSES
    for name, prof_flags in passes:
        raw = os.path.join("/tmp", "apd_prof_%s_%s" % (tag, name))
        subprocess.call(["rm", "-rf", raw])
        cmd = ["rocprofv3"] + prof_flags + ["--output-format", "csv", "-d", raw, "-o", name, "--", sys.executable,
                                           os.path.join(ROOT, "bench.py")] + flags + tail
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, timeout=300)  # a counter name rocprofv3 does not know can hang it
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            sys.stderr.write("pass %s failed (rc %d):\n%s\n" % (name, r.returncode, r.stderr[-2000:]))
            return 1
        with open(os.path.join(out_dir, "bench_%s_pass_%s.json" % (tag, name)), "w") as f:
            f.write(line[-1] + "\n")
        # launch durations of this pass (every pass has the kernel trace)
        for path in glob.glob(os.path.join(raw, "**", "*kernel_trace.csv"), recursive=True):
            rows = sorted(csv.DictReader(open(path)), key=lambda r_: int(r_["Start_Timestamp"]))
            for r_ in rows:
                for key, sub in kernels_of_interest.items():
                    if sub in r_["Kernel_Name"]:
                        per_dispatch[key]["duration_ns@" + name].append(float(r_["End_Timestamp"]) - float(r_["Start_Timestamp"]))
        for path in glob.glob(os.path.join(raw, "**", "*counter_collection.csv"), recursive=True):
            rows = list(csv.DictReader(open(path)))
            rows.sort(key=lambda r_: int(r_.get("Dispatch_Id", 0)))
            for r_ in rows:
                for key, sub in kernels_of_interest.items():
                    if sub in r_["Kernel_Name"]:
                        per_dispatch[key][r_["Counter_Name"]].append(float(r_["Counter_Value"]))
                        meta[key] = {"kernel_name": r_["Kernel_Name"], "grid": r_["Grid_Size"], "workgroup": r_["Workgroup_Size"],
                                     "lds_bytes": r_["LDS_Block_Size"], "scratch_bytes_per_lane": r_["Scratch_Size"],
                                     # rocprofv3's VGPR_Count / SGPR_Count columns are granule values of the dispatch packet, NOT the kernel's
                                     # allocation: the allocation, spills and occupancy are in profiles/<round>/kernel_resources.txt (tools/kres.py)
                                     "rocprof_vgpr_field": r_["VGPR_Count"], "rocprof_sgpr_field": r_["SGPR_Count"]}
        if name == "trace":
            for path in glob.glob(os.path.join(raw, "**", "*kernel_stats.csv"), recursive=True):
                with open(path) as f, open(os.path.join(out_dir, "kernel_stats_%s.csv" % tag), "w") as g:
                    g.write(f.read())
        subprocess.call(["rm", "-rf", raw])

    out = {"config": {"workload": workload, "steps": steps, "warmup": warmup, "flags": flags, "seed": flag(flags, "--seed", 12345),
                      "options": [flags[i + 1] for i, f_ in enumerate(flags) if f_ == "--opt"]},
           "method": "rocprofv3 --kernel-trace + one --pmc pass per counter group; timed launches = last 2*steps dispatches of the kernel; "
                     "FETCH_SIZE in KiB doubled (gfx950 counts 128-B requests as 64 B), WRITE_SIZE in KiB as reported",
           "kernels": {}}
    if pass_key:
        out["config"].update({"pass_key": pass_key, "pass_kind": pass_kind, "passes": n_timed, "steps": None, "warmup": None})
        out["method"] = ("rocprofv3 --kernel-trace + one --pmc pass per counter group over bench.py's whole-pass sub-line; timed launches = the "
                         "last `passes` dispatches of K14 / K15 (one launch per apd_run); FETCH_SIZE in KiB doubled, WRITE_SIZE in KiB as reported")
    for key, counters in per_dispatch.items():
        k = dict(meta.get(key, {}))
        k["launches_timed"] = n_timed
        k["per_dispatch_timed"] = {}
        for cname, vals in sorted(counters.items()):
            timed = vals[-n_timed:]
            k["per_dispatch_timed"][cname] = timed
            for i, v in enumerate(timed):
                rows_csv.append([key, cname, i, v])
        # Timed launch j of ANY command line of this workload (same seed, same options) is launch j of this one: bench.py
        # re-initialises the pass after the warm-up, so its timed region is always iterations 0..steps-1, black then red,
        # and no kernel reads max_iterations.  bench.py therefore serves shorter runs from a prefix of these values
        # (and says so: roofline.pmc_profile_steps).

        def mean(cname):
            v = k["per_dispatch_timed"].get(cname)
            return sum(v) / len(v) if v else None
        k["launch_ms"] = None if mean("duration_ns@trace") is None else mean("duration_ns@trace") / 1e6
        k["launch_ms_sq_pass"] = None if mean("duration_ns@sq") is None else mean("duration_ns@sq") / 1e6
        k["valu_insts_per_launch"] = mean("SQ_INSTS_VALU")
        k["active_inst_valu_per_launch"] = mean("SQ_ACTIVE_INST_VALU")
        k["wave_cycles_per_launch"] = mean("SQ_WAVE_CYCLES")
        k["wait_inst_any_per_launch"] = mean("SQ_WAIT_INST_ANY")
        k["wait_any_per_launch"] = mean("SQ_WAIT_ANY")
        k["vmem_rd_insts_per_launch"] = mean("SQ_INSTS_VMEM_RD")
        k["valu_class_insts_per_launch"] = {c[len("SQ_INSTS_VALU_"):]: mean(c) for c in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32",
                                                                                       "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32",
                                                                                       "SQ_INSTS_VALU_INT64") if mean(c) is not None}
        k["tcp_tag_accesses_per_launch"] = mean("TCP_TOTAL_CACHE_ACCESSES_sum")
        k["tcp_to_l2_read_requests_per_launch"] = mean("TCP_TCC_READ_REQ_sum")
        k["l2_hits_per_launch"], k["l2_misses_per_launch"] = mean("TCC_HIT_sum"), mean("TCC_MISS_sum")
        fe, wr = mean("FETCH_SIZE"), mean("WRITE_SIZE")
        k["fetch_bytes_per_launch"] = None if fe is None else fe * 1024 * 2
        k["write_bytes_per_launch"] = None if wr is None else wr * 1024
        k["hbm_bytes_per_launch"] = None if fe is None or wr is None else fe * 1024 * 2 + wr * 1024
        k["mean_over_timed_launches"] = {c: mean(c) for c in sorted(k["per_dispatch_timed"])}
        out["kernels"][key] = k
    if extra:
        with open(os.path.join(out_dir, "pmc_extra_%s_%s.json" % ("_".join(extra), tag)), "w") as f:
            json.dump(out, f, indent=1)
        for key, k in out["kernels"].items():
            print(key, json.dumps(k["mean_over_timed_launches"]))
        return 0
    stem = "pmc_pass_%s" if pass_key else "pmc_bench_%s"
    with open(os.path.join(out_dir, (stem + ".json") % tag), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(out_dir, (stem + ".csv") % tag), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "timed_launch_index", "value"])
        w.writerows(rows_csv)
    for key, k in out["kernels"].items():
        print("%s: %s launches, %.3f ms/launch, VALU %.4g/launch, fetch %.4g B, write %.4g B, scratch %s B/lane" % (
            key, k["launches_timed"], k["launch_ms"] or 0, k["valu_insts_per_launch"] or 0, k["fetch_bytes_per_launch"] or 0,
            k["write_bytes_per_launch"] or 0, k.get("scratch_bytes_per_lane")))
    return 0


if __name__ == "__main__":
    sys.exit(main())
