#!/bin/bash
# Round 3, GPU call 2: parity of the tap-cooperative K9/K10, A/B against per-lane gathers, counters.
OUT=gpurun_out/r03b; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -8
export TUNE_WORKLOAD=eth3d_pipes_fullres_10src_apd TUNE_STEPS=3
echo "== K9/K10 cooperative gathers A/B"
tools/tune.sh "" "-DAPD_LAB_K910_NO_COOP" "-DAPD_LAB_COOP_IMAJOR" "-DAPD_K910_WAVES=3" 2>&1 | tee $OUT/ab_k910_coop.txt
python apd-mvs_amd/build.py --force > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
echo "== counters, default build"
timeout 1200 python tools/profile_bench.py $OUT --workload eth3d_pipes_fullres_10src_apd --steps 3 --warmup 1 2>&1 | tail -5
echo "== three-pass timing 4096x3072"
timeout 600 python tools/pass_timing.py 2>&1 | tail -40 | tee $OUT/pass_timing.txt
ls $OUT
