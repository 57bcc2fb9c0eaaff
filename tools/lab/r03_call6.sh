#!/bin/bash
OUT=gpurun_out/r03f; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest drop-in"; timeout 900 python -m pytest tests/test_gpu_dropin_binary.py -m gpu -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -4
export TUNE_WORKLOAD=eth3d_pipes_fullres_10src_apd TUNE_STEPS=3
tools/tune.sh "" "-mllvm -amdgpu-promote-alloca-to-vector-limit=2048" "-DAPD_K910_WAVES=1" 2>&1 | tee $OUT/ab_k910_alloca.txt
