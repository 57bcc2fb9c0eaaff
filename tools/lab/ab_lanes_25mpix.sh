#!/bin/bash
# Same-box A/B: one or two views in flight at the 25.6 Mpix level (6 views of 6200 x 4130, ten sources; the coarser levels as the table says).
# apd-mvs_amd/_build/APD_l2 = the host sources compiled with -DAPD_LANES_ABOVE_12MPIX=2 next to the regular binary (git-ignored, travels with the snapshot).
O=gpurun_out/lanes25; mkdir -p $O; cd /root/repo
d=/tmp/eth_6; rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 6200 --height 4130 --views 6 --src 5 --textureless 0.2 --jpeg > /dev/null
for bin in APD APD_l2 APD APD_l2; do
  rm -rf $d/APD
  t1=$(date +%s%N); apd-mvs_amd/_build/$bin $d 0 --seed 7 > /tmp/l.log 2>&1; rc=$?; t2=$(date +%s%N)
  echo "$bin: rc $rc wall $(( (t2-t1)/1000000 )) ms | $(grep Stages /tmp/l.log) | $(md5sum $d/APD/APD.ply | cut -c1-8)" | tee -a $O/ab.txt
done
