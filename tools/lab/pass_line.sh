#!/bin/bash
# Whole-pass sub-line: its counter profile first, then the two bench lines of the round (which read it from profiles/r04/).
O=gpurun_out/pass_line; mkdir -p $O; cd /root/repo; export TMPDIR=/tmp
for key in ${PASS_KEYS:-configs2_pipes_apd_whole_pass configs2_pipes_apd_geometric_pass}; do
  APD_PROFILE_PASS_KEY=$key timeout 1200 python tools/profile_bench.py $O > $O/profile_$key.log 2>&1; tail -3 $O/profile_$key.log
done
cp $O/pmc_pass_*.json profiles/r04/
(time python bench.py) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
(time python bench.py --steps 20 --warmup 5 --no-cpu-baseline) > $O/bench_driver_s20_w5.json 2> $O/bench_driver.err; tail -4 $O/bench_driver.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/pass_line/bench_driver_s20_w5.json").readline())
print("headline", d["value"], d["wall_s"])
for k, v in d["workloads"].items():
    print(k, v["value"], v.get("ms_per_step"), v.get("ms_per_pass"))
for key in ("configs2_pipes_apd_whole_pass", "configs2_pipes_apd_geometric_pass"):
    p = d["workloads"][key]
    print(key, json.dumps({k: p[k] for k in ("kernel_ms_per_pass", "share_of_kernel_time", "rank_ms_per_pass", "wall_with_prior_uploads_ms")}))
    print({k: (v["avg_launch_ms"], v["frac"], v["hbm"] and v["hbm"]["frac"], v["pmc_source"]) for k, v in p["pass_kernels"].items()})
PY
python tools/recompute_roofline.py $O | tail -12
