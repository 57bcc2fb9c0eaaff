#!/bin/bash
# Views in flight with the wavefront scheduler (reference order), 24 views of 1080p: default table (6 / 4) against a fixed count at every level.
O=gpurun_out/lanesw; mkdir -p $O; cd /root/repo
d=/tmp/tt24; rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
for args in "" "--ranks 4" "--ranks 6" "--ranks 8" "--ranks 12" "" "--ranks 6" "--ranks 8"; do
  rm -rf $d/APD
  t1=$(date +%s%N); apd-mvs_amd/_build/APD $d 0 --seed 7 $args > /tmp/l.log 2>&1; rc=$?; t2=$(date +%s%N)
  echo "[$args]: rc $rc wall $(( (t2-t1)/1000000 )) ms | $(grep Stages /tmp/l.log | sed 's/images + cameras.*upload) [0-9]* ms, //') | $(md5sum $d/APD/APD.ply | cut -c1-8)" | tee -a $O/ab.txt
done
