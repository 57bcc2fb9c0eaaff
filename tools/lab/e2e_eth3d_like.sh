#!/bin/bash
# BASELINE.json configs[1] / configs[2] frame size end to end with a realistic neighbourhood: V views of 6200 x 4130, ten sources each, four
# pyramid levels x four passes, fusion -- `APD folder 0` (in memory, the reference's order) and once through the files (--files) for the md5.
V=${1:-12}
d=/tmp/eth_$V; rm -rf $d ${d}_f; mkdir -p $d
t0=$(date +%s%N)
python tools/make_synthetic_dense.py $d --width 6200 --height 4130 --views $V --src 10 --textureless 0.2 --jpeg > /dev/null
cp -r $d ${d}_f
t1=$(date +%s%N)
apd-mvs_amd/_build/APD $d 0 --seed 7 > /tmp/eth_a.log 2>&1; rc=$?
t2=$(date +%s%N)
echo "$V views 6200x4130, 10 sources: APD folder 0: rc $rc, wall $(( (t2-t1)/1000000 )) ms (folder written in $(( (t1-t0)/1000000 )) ms)"
grep -E "rank\(s\)|Stages|Fused|Fusion \+|Start-up|in flight" /tmp/eth_a.log | sort -u | tail -10
[ $rc -ne 0 ] && tail -5 /tmp/eth_a.log
md5sum $d/APD/APD.ply
if [ "${2:-files}" = files ]; then
  t3=$(date +%s%N)
  apd-mvs_amd/_build/APD ${d}_f 0 --seed 7 --files > /tmp/eth_f.log 2>&1; rc=$?
  t4=$(date +%s%N)
  echo "$V views 6200x4130, 10 sources: APD folder 0 --files: rc $rc, wall $(( (t4-t3)/1000000 )) ms"
  md5sum ${d}_f/APD/APD.ply
fi
