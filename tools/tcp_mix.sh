#!/bin/bash
# tools/tcp_mix on the GPU box: timings, then the access counts of the same launches (one rocprofv3 --pmc pass per group).
# Usage: tools/tcp_mix.sh <out_dir>
OUT=${1:-gpurun_out/tcp_mix}; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
python -c "import sys; sys.path.insert(0, 'apd-mvs_amd'); import build; build.build_tools()" > /dev/null
tools/_build/tcp_mix | tee "$OUT/tcp_mix.txt"
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$OUT/g$i" -o pmc -- tools/_build/tcp_mix > /dev/null 2> "$OUT/g$i.err" || echo "group $i failed"
done
python - "$OUT" <<'PY' | tee -a "$OUT/tcp_mix.txt"
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)
for path in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "mix" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for name, v in per.items():
        v.sort()
        for j, (_, val) in enumerate(v):
            rows[j][name] = val
names = ["hits only", "misses only", "1 hit per miss", "3 hits per miss", "7 hits per miss", "32 lanes miss", "16 lanes miss", "8 lanes miss", "4 lanes miss"]
for j in sorted(rows):
    if j % 2 == 0:
        continue  # first launch of every case is the warm-up
    c = rows[j]
    n = c.get("SQ_INSTS_VMEM_RD", 0.0)
    if n:
        print("%-16s per wave-level gather: tag accesses %6.2f  L1->L2 requests %6.2f   L2 hit rate %.3f" % (
            names[j // 2], c["TCP_TOTAL_CACHE_ACCESSES_sum"] / n, c["TCP_TCC_READ_REQ_sum"] / n,
            c.get("TCC_HIT_sum", 0.0) / max(c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0), 1.0)))
PY
find "$OUT" -type f -size +1M -delete
