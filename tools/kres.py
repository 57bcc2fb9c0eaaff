#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: tools/kres.py apd-mvs_amd/csrc/apd_kernels.hip [extra hipcc flags...]"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"]
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
