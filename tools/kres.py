#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: tools/kres.py apd-mvs_amd/csrc/apd_kernels.hip [extra hipcc flags...]
       tools/kres.py --all      every kernel file of the library with the flags apd-mvs_amd/build.py compiles it with
                                (the table committed as profiles/<round>/kernel_resources.txt)"""
import re
import subprocess
import sys

import os

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"]
if len(sys.argv) > 1 and sys.argv[1] == "--all":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    per_file = {"apd_kernels_k67w.hip": ["-mllvm", "-amdgpu-promote-alloca-to-vector-limit=2048"]}   # build.py: FILE_FLAGS
    for f in ("apd_kernels_k67w.hip", "apd_kernels_k1415w.hip", "apd_kernels_weak.hip", "apd_kernels.hip", "apd_fusion.hip", "apd_exchange.hip"):
        print("== csrc/%s %s" % (f, " ".join(per_file.get(f, []))))
        sys.stdout.flush()
        subprocess.call([sys.executable, os.path.abspath(__file__), os.path.join(root, "apd-mvs_amd", "csrc", f)] + per_file.get(f, []))
    sys.exit(0)
src, extra = sys.argv[1], sys.argv[2:]
r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
rows, cur = [], None
for line in r.stdout.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, _, v = m.group(1).partition(": ")
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    elif cur is not None:
        cur[k.strip()] = v
print("%-58s %5s %5s %6s %7s %4s %6s" % ("kernel", "VGPR", "AGPR", "spill", "scratch", "occ", "LDS"))
for c in rows:
    print("%-58s %5s %5s %6s %7s %4s %6s" % (c["name"][-58:], c.get("VGPRs"), c.get("AGPRs"), c.get("VGPRs Spill"),
                                            c.get("ScratchSize [bytes/lane]"), c.get("Occupancy [waves/SIMD]"), c.get("LDS Size [bytes/block]")))
