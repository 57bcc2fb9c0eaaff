#!/bin/bash
OUT=gpurun_out/r03w; mkdir -p $OUT; export TMPDIR=/tmp
export TUNE_WORKLOAD=eth3d_pipes_fullres_10src_apd TUNE_STEPS=3
# 16.1 KB -> 8 workgroups per CU today; pads: 3584 -> 19.6 KB (8), 6656 -> 22.6 KB (7), 10240 -> 26.1 KB (6), 16384 -> 32.1 KB (4, 5?)
tools/tune.sh "" "-DAPD_LAB_K910_LDS_PAD=6656" "-DAPD_LAB_K910_LDS_PAD=10240" "-DAPD_LAB_K910_LDS_PAD=16384" 2>&1 | tee $OUT/ab_k910_occupancy.txt
