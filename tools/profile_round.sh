#!/bin/bash
# One round's measurement set, on the GPU box: counter profiles (tools/profile_bench.py) of every workload a bench line is
# committed for, then the bench lines themselves -- made AFTER their profiles, in the same checkout, so that every roofline
# field follows from the committed counters (tools/recompute_roofline.py).  Usage: tools/profile_round.sh <out_dir> [round, e.g. r04]
OUT=${1:-gpurun_out/profile_round}
ROUND=${2:-r06}
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
run_profile() {  # workload steps
  timeout 1500 python tools/profile_bench.py "$OUT" --workload "$1" --steps "$2" --warmup 1 > "$OUT/profile_$1.log" 2>&1 || echo "profile of $1 failed" >> "$OUT/errors.txt"
}
run_profile eth3d_office_fullres_8src 24
run_profile eth3d_office_halfres_2src 3      # configs[0]
run_profile eth3d_pipes_fullres_10src_apd 6
run_profile synthetic_4096x3072_16src 8
run_profile tt_family_1080p_10src 24
run_profile eth3d_office_fullres_8src_hard 6
run_profile eth3d_pipes_fullres_10src_apd_hard 3
# whole passes (K14 / K15): the sub-lines of bench.PASS_WORKLOADS
for key in configs2_pipes_apd_whole_pass configs2_pipes_apd_geometric_pass configs2_pipes_hard_whole_pass; do
  APD_PROFILE_PASS_KEY=$key timeout 1500 python tools/profile_bench.py "$OUT" > "$OUT/profile_$key.log" 2>&1 || echo "profile of $key failed" >> "$OUT/errors.txt"
done
mkdir -p profiles/$ROUND && cp "$OUT"/pmc_bench_*.json "$OUT"/pmc_pass_*.json profiles/$ROUND/   # where they will be committed; bench.py looks under profiles/*/
# the two lines of the round: the default command and the driver's; each carries the `workloads` block (every BASELINE config)
# stdout of each command = the compact line the driver parses (line_*.json, <= 2000 bytes); the full block is bench_workloads.json
python bench.py > "$OUT/line_default.json" 2> "$OUT/bench_default.err"; cp bench_workloads.json "$OUT/bench_default.json"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/line_driver_s20_w5.json" 2>/dev/null; cp bench_workloads.json "$OUT/bench_driver_s20_w5.json"
wc -c "$OUT"/line_*.json
# the rocprofv3 --kernel-trace --stats summary of the driver's command, workloads block included
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/apd_trace_driver -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_driver_traced.json" 2> "$OUT/trace_driver.err"
find /tmp/apd_trace_driver -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_bench_driver_s20_w5_with_workloads.csv" \;
rm -rf /tmp/apd_trace_driver

for f in "$OUT"/bench_default.json "$OUT"/bench_driver_s20_w5.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
def show(tag, v):
    r = v.get("roofline") or {}
    print("%-46s %8.2f Mpix*iter/s  %-16s frac %-7s %s ms/launch  src %s" % (tag, v["value"], r.get("bound"), r.get("frac"), r.get("avg_launch_ms"), r.get("pmc_source")))
    for kn, kr in sorted((v.get("pass_kernels") or {}).items()):
        print("%-46s %8s %-16s frac %-7s %s ms/launch  src %s" % ("      " + kn, "", kr.get("bound"), kr.get("frac"), kr.get("avg_launch_ms"), kr.get("pmc_source")))
show(sys.argv[1].split("/")[-1], d)
for k, v in (d.get("workloads") or {}).items():
    show("  " + k, v)
print("  wall_s", d.get("wall_s"))
PY
done
python tools/recompute_roofline.py "$OUT" | tail -20
