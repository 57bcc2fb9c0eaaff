#!/bin/bash
# Same-box A/B of the multi-rank scheduler: APD_l2 = passes separated by a join + all-gather (the scheduler before), APD = tasks of a level
# handed out across passes, only depth-reading halves wait for the exchange.  24 views of 1080p, device lists on one device (peer copies) and --rccl.
# apd-mvs_amd/_build/APD_l2 = a build of host/multi_device.cpp of the commit before the change (git-ignored, travels with the snapshot).
O=gpurun_out/overlap; mkdir -p $O; cd /root/repo
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_dropin_binary.py > $O/pytest.log 2>&1; tail -2 $O/pytest.log
d=/tmp/tt24; rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
for args in "0,0" "0,0,0" "0 --jacobi --rccl"; do
  for bin in APD_l2 APD APD_l2 APD; do
    rm -rf $d/APD
    t1=$(date +%s%N); apd-mvs_amd/_build/$bin $d $args --seed 7 > /tmp/l.log 2>&1; rc=$?; t2=$(date +%s%N)
    echo "$bin $args: rc $rc wall $(( (t2-t1)/1000000 )) ms | $(grep Stages /tmp/l.log | sed 's/images + cameras.*upload) [0-9]* ms, //') | $(md5sum $d/APD/APD.ply | cut -c1-8)" | tee -a $O/ab.txt
  done
done
