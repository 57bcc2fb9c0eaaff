#!/bin/bash
OUT=gpurun_out/r03y; mkdir -p $OUT; export TMPDIR=/tmp
export TUNE_WORKLOAD=eth3d_pipes_fullres_10src_apd TUNE_STEPS=3
tools/tune.sh "" "-DAPD_K910_WINDOW=0" "-DAPD_K910_WIN_DIVERGENT=0" "-DAPD_K910_WIN_H=22" 2>&1 | tee $OUT/ab_k910_centre_window.txt
python apd-mvs_amd/build.py --force > /tmp/build.log 2>&1
bash tools/lab/bench_quick.sh 2>&1 | tee $OUT/bench_quick.txt
