#!/usr/bin/env python3
"""Issue-cycle accounting of the NCC bodies of K6/K7 from their ISA (runs here: hipcc cross-compiles, no GPU needed).

gfx950 issues the plain binary32 multiply / add / FMA (and v_mov, v_add_u32, v_and_b32) in ~2 cycles per wave64 instruction,
transcendental instructions in ~8 and everything else in ~4 (tools/valu_issue.hip -> profiles/r02/valu_issue.csv).  A count of
VALU instructions alone therefore says little about how busy the pipe is; this tool weights the static instruction mix of
the three 36-sample bodies of k67w_update_strong<8, true, false, false> (LDS window, global fast reciprocal, global IEEE division)
with the measured costs and writes profiles/<round>/valu_mix_k67w.json; bench.py uses the file of the SAME round directory as the
counter profile for `roofline.valu_busy_estimate` (a mix of another round's kernel is never borrowed).

usage: tools/valu_mix.py [out.json]
"""
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-promote-alloca-to-vector-limit=2048"]
FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32", "v_and_b32",
        "v_fmaak_f32", "v_fmamk_f32", "v_lshrrev_b32", "v_or_b32", "v_sub_u32", "v_subrev_u32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"}


def measured_costs():
    """cycles per wave-instruction per SIMD at 8 waves/SIMD from the microbenchmark, by mnemonic"""
    costs = {}
    path = os.path.join(ROOT, "profiles", "r02", "valu_issue.csv")
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            if r["waves_per_simd"] == "8":
                m = re.match(r"(v_[a-z0-9_]+)", r["kind"])
                if m and "dependent" not in r["kind"] and "alternating" not in r["kind"]:
                    costs[m.group(1)] = float(r["cycles_per_op_per_simd"])
    return costs


def base(op):
    return re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04", "valu_mix_k67w.json")
    asm = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", os.path.join(ROOT, "apd-mvs_amd", "csrc", "apd_kernels_k67w.hip"),
                          "-o", "-"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    m = re.search(r"^_ZN3apd18k67w_update_strongILi8ELb1ELb0ELb0EEEvNS_9FrameArgsEii:(.*?)s_endpgm", asm, re.S | re.M)
    assert m, "kernel not found in the ISA"
    blocks, cur = [], None
    for line in m.group(1).split("\n"):
        lab = re.match(r"^(\.LBB\d+_\d+):", line)
        if lab:
            cur = {"name": lab.group(1), "ops": {}}
            blocks.append(cur)
        elif cur is not None and re.match(r"^\s+[a-z]", line):
            op = line.split()[0]
            cur["ops"][op] = cur["ops"].get(op, 0) + 1
    costs = measured_costs()
    fast_c = costs.get("v_fma_f32", 2.2)
    slow_c = costs.get("v_fract_f32", 4.1)
    trans_c = costs.get("v_rcp_f32", 8.1)

    def account(b):
        valu = {k: v for k, v in b["ops"].items() if k.startswith("v_")}
        n = sum(valu.values())
        cyc = 0.0
        classes = {"fast": 0, "slow": 0, "transcendental": 0}
        for op, cnt in valu.items():
            o = base(op)
            if o in TRANS:
                cyc += cnt * trans_c
                classes["transcendental"] += cnt
            elif o in FAST:
                cyc += cnt * fast_c
                classes["fast"] += cnt
            else:
                cyc += cnt * slow_c
                classes["slow"] += cnt
        return {"block": b["name"], "valu_insts": n, "issue_cycles": round(cyc, 1), "mean_cycles_per_inst": round(cyc / n, 3),
                "per_sample_insts": round(n / 36.0, 2), "per_sample_cycles": round(cyc / 36.0, 2), "classes": classes,
                "lds_reads": sum(v for k, v in b["ops"].items() if k.startswith("ds_read")),
                "global_loads": sum(v for k, v in b["ops"].items() if k.startswith("global_load")),
                "mix": dict(sorted(valu.items(), key=lambda kv: -kv[1]))}
    res = {"kernel": "k67w_update_strong<8, true, false, false>", "class_cycles": {"fast": fast_c, "slow": slow_c, "transcendental": trans_c},
           "source": "static ISA of the 36-sample bodies (first of the two copies: propagation phase), costs from profiles/r02/valu_issue.csv"}
    win = [b for b in blocks if b["ops"].get("ds_read2st64_b32", 0) >= 30]
    glob = [b for b in blocks if sum(v for k, v in b["ops"].items() if k.startswith("global_load")) >= 30]
    res["window_body"] = account(win[0])
    glob.sort(key=lambda b: sum(b["ops"].values()))
    res["global_fast_body"] = account(glob[0])
    # since round 4 the IEEE-division body is two rolled loops (one sample per trip): the block that holds the division sequence
    ieee = [b for b in blocks if any(k.startswith("v_div_fmas_f32") for k in b["ops"]) and any(k.startswith("global_load") for k in b["ops"])]
    if ieee:
        one = account(ieee[0])
        one["note"] = "rolled: this block is ONE sample (36 trips per NCC); per_sample_* = the block itself"
        one["per_sample_insts"], one["per_sample_cycles"] = float(one["valu_insts"]), one["issue_cycles"]
        res["global_ieee_body"] = one
    else:
        res["global_ieee_body"] = account(glob[-1])
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    for k in ("window_body", "global_fast_body", "global_ieee_body"):
        b = res[k]
        print("%-18s %5d VALU  %7.0f issue cycles  %.2f cycles/inst  (%.1f inst, %.1f cycles per sample)  %s" % (
            k, b["valu_insts"], b["issue_cycles"], b["mean_cycles_per_inst"], b["per_sample_insts"], b["per_sample_cycles"], b["classes"]))


if __name__ == "__main__":
    main()
