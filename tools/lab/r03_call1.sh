#!/bin/bash
# Round 3, GPU call 1: parity of the ordered WEAK lists, A/B of list order / XCD mapping for K9/K10, live-view sweep, duplicate statistics.
OUT=gpurun_out/r03a; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
export TUNE_WORKLOAD=eth3d_pipes_fullres_10src_apd TUNE_STEPS=3
echo "== K9/K10 list order A/B"
tools/tune.sh "" "-DAPD_LAB_K910_INTERLEAVED" "-DAPD_WEAK_SUPER_SHIFT=10" "-DAPD_WEAK_SUPER_SHIFT=10 -DAPD_LAB_K910_INTERLEAVED" "-DAPD_K910_WAVES=3" "-DAPD_WEAK_SUPER_SHIFT=3" "-DAPD_WEAK_SUPER_SHIFT=5" 2>&1 | tee $OUT/ab_k910_order.txt
python apd-mvs_amd/build.py --force > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
echo "== live source views sweep (default build)"
for n in 1 2 5 10; do
  timeout 600 python bench.py --workload custom_6200x4130_${n}src_apd --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); w=d['weak_path']
print('N', d['config']['num_src'], 'weak_fraction', w['weak_fraction'], 'k910 ms/launch', w['avg_launch_ms'], 'per view', round(w['avg_launch_ms']/d['config']['num_src'],3), 'k67 ms/launch', d['roofline']['avg_launch_ms'], 'value', d['value'])"
done 2>&1 | tee $OUT/nsweep.txt
echo "== counters, default build"
timeout 1200 python tools/profile_bench.py $OUT --workload eth3d_pipes_fullres_10src_apd --steps 3 --warmup 1 2>&1 | tail -5
echo "== duplicate hypotheses"
timeout 600 python tools/dup_stats.py 4096 3072 8 5 2>&1 | tee $OUT/dup_stats_4096x3072_8src.txt | tail -12
ls -la $OUT
