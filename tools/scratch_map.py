#!/usr/bin/env python3
"""Where a kernel touches scratch memory: per basic block of its ISA, the number of scratch loads / stores next to the block's
VALU / LDS / global-memory instruction counts (runs here: hipcc cross-compiles, no GPU needed).  A spill inside a 36-sample body
costs on every NCC; one in the per-pixel prologue costs once.

usage: tools/scratch_map.py <file.hip> <mangled-name substring> [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"]


def main():
    src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    if os.path.basename(src) == "apd_kernels_k67w.hip":
        extra = ["-mllvm", "-amdgpu-promote-alloca-to-vector-limit=2048"] + extra
    asm = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-S", "--cuda-device-only", src, "-o", "-"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    for f in re.split(r"\n(?=_ZN3apd\S+:\s)", asm):
        name = f.split(":")[0]
        if pat not in name:
            continue
        print(name)
        blocks, cur = [], {"name": "entry", "n": 0, "ld": 0, "st": 0, "valu": 0, "ds": 0, "vmem": 0, "rcp": 0, "line": 0}
        blocks.append(cur)
        for ln, line in enumerate(f.split("\n")):
            lab = re.match(r"^(\.LBB\d+_\d+):", line)
            if lab:
                cur = {"name": lab.group(1), "n": 0, "ld": 0, "st": 0, "valu": 0, "ds": 0, "vmem": 0, "rcp": 0, "line": ln}
                blocks.append(cur)
            elif re.match(r"^\s+[a-z]", line):
                op = line.split()[0]
                cur["n"] += 1
                cur["ld"] += op.startswith("scratch_load")
                cur["st"] += op.startswith("scratch_store")
                cur["valu"] += op.startswith("v_")
                cur["ds"] += op.startswith("ds_")
                cur["vmem"] += op.startswith(("global_", "buffer_", "flat_"))
                cur["rcp"] += op.startswith("v_rcp_f32")
        tot_ld = sum(b["ld"] for b in blocks)
        tot_st = sum(b["st"] for b in blocks)
        print("  %d blocks, %d instructions, scratch loads %d, stores %d" % (len(blocks), sum(b["n"] for b in blocks), tot_ld, tot_st))
        print("  %-12s %6s %6s %5s %5s %5s %5s %5s" % ("block", "insts", "valu", "rcp", "ds", "vmem", "s.ld", "s.st"))
        for b in blocks:
            if b["ld"] or b["st"] or b["n"] >= 200:
                print("  %-12s %6d %6d %5d %5d %5d %5d %5d" % (b["name"], b["n"], b["valu"], b["rcp"], b["ds"], b["vmem"], b["ld"], b["st"]))
        for l in f.split("\n"):
            if re.search(r"; (ScratchSize|Occupancy|NumVgprs|codeLenInByte)", l):
                print("  " + l.strip())


if __name__ == "__main__":
    main()
