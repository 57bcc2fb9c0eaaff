#!/bin/bash
# Same-box A/B of K14's launch bounds in the (sample, lane)-pair variant (ten and more sources): 4 waves per SIMD (128 VGPRs, 42 spilled) against
# 3 (164 VGPRs, none): the whole-pass sub-lines of bench.py on configs[2].
# tools/_build/libapd_k14w3.so = the library linked with apd_kernels_k1415w.hip compiled with -DAPD_K14W_WAVES=3 (git-ignored, travels with the snapshot).
O=gpurun_out/k14waves; mkdir -p $O; cd /root/repo
cp apd-mvs_amd/_build/libapd_mi355x.so /tmp/cur.so
for which in cur w3 cur w3; do
  if [ $which = w3 ]; then cp tools/_build/libapd_k14w3.so apd-mvs_amd/_build/libapd_mi355x.so; else cp /tmp/cur.so apd-mvs_amd/_build/libapd_mi355x.so; fi
  touch apd-mvs_amd/_build/libapd_mi355x.so
  timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --only-workloads configs2_pipes_apd_3iter --only-workloads configs2_pipes_apd_whole_pass --only-workloads configs2_pipes_apd_geometric_pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
for k in ('configs2_pipes_apd_whole_pass', 'configs2_pipes_apd_geometric_pass'):
    p = d['workloads'][k]
    print('$which', k, 'ms_per_pass', p['ms_per_pass'], 'K14', p['kernel_ms_per_pass']['DepthToWeak'], 'K15', p['kernel_ms_per_pass']['LocalRefine'])
" | tee -a $O/ab.txt
done
cp /tmp/cur.so apd-mvs_amd/_build/libapd_mi355x.so
