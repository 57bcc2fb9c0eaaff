#!/bin/bash
# Same-box A/B: how many lanes may work on ONE geometric pass in the reference's order (APD = 2; APD_g<N> = host built with -DAPD_GS_LANES_PER_PASS=N,
# git-ignored binaries next to the regular one).  24 views of 1080p, 10 sources.
O=gpurun_out/gslanes; mkdir -p $O; cd /root/repo
d=/tmp/tt24; rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
for bin in APD APD_g1 APD_g3 APD_g9 APD APD_g1 APD_g3 APD_g9; do
  rm -rf $d/APD
  t1=$(date +%s%N); apd-mvs_amd/_build/$bin $d 0 --seed 7 > /tmp/l.log 2>&1; rc=$?; t2=$(date +%s%N)
  echo "$bin: rc $rc wall $(( (t2-t1)/1000000 )) ms | $(grep Stages /tmp/l.log | sed 's/images + cameras.*upload) [0-9]* ms, //') | $(md5sum $d/APD/APD.ply | cut -c1-8)" | tee -a $O/ab.txt
done
