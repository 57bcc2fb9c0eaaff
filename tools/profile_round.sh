#!/bin/bash
# One round's measurement set, on the GPU box: counter profiles (tools/profile_bench.py) of every workload a bench line is
# committed for, then the bench lines themselves -- made AFTER their profiles, in the same checkout, so that every roofline
# field follows from the committed counters (tools/recompute_roofline.py).  Usage: tools/profile_round.sh <out_dir>
OUT=${1:-gpurun_out/profile_round}
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
run_profile() {  # workload steps
  timeout 1500 python tools/profile_bench.py "$OUT" --workload "$1" --steps "$2" --warmup 1 > "$OUT/profile_$1.log" 2>&1 || echo "profile of $1 failed" >> "$OUT/errors.txt"
}
run_profile eth3d_office_fullres_8src 24
run_profile eth3d_pipes_fullres_10src_apd 6
run_profile synthetic_4096x3072_16src 8
run_profile tt_family_1080p_10src 24
mkdir -p profiles/r03 && cp "$OUT"/pmc_bench_*.json profiles/r03/   # where they will be committed; bench.py looks under profiles/*/
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_driver_s20_w5.json" 2>/dev/null
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_ref_default_pass_s3_w1.json" 2>/dev/null
python bench.py --workload eth3d_pipes_fullres_10src_apd --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_apd_s3_w1.json" 2>/dev/null
python bench.py --workload synthetic_4096x3072_16src --steps 8 --warmup 1 --no-cpu-baseline > "$OUT/bench_16src_s8_w1.json" 2>/dev/null
python bench.py --workload tt_family_1080p_10src --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_tt1080p_s20_w5.json" 2>/dev/null

for f in "$OUT"/bench_*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
r = d["roofline"]
print("%-34s %8.2f %s  %s frac %s  %.3f ms/launch  src %s" % (sys.argv[1].split("/")[-1], d["value"], d["unit"], r["bound"], r["frac"], r["avg_launch_ms"], r.get("pmc_source")))
PY
done
