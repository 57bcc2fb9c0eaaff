#!/bin/bash
# BASELINE.json configs[3] at its real view count on ONE device: a Tanks&Temples-Family-shaped folder (152 views of 1920 x 1080, ten
# sources each, two pyramid levels, eight passes) through the drop-in binary as the reference is called (`APD folder 0`), with the
# stage line that exposes what does not shard (image decode, level images, final gather, fusion).  Usage: tools/e2e_c4.sh [views] [extra APD flags]
V=${1:-152}; shift
d=/tmp/c4_$V; rm -rf $d; mkdir -p $d
t0=$(date +%s%N)
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views $V --src 10 --textureless 0.2 --jpeg > /dev/null
t1=$(date +%s%N)
apd-mvs_amd/_build/APD $d 0 --seed 7 "$@" > /tmp/c4.log 2>&1; rc=$?
t2=$(date +%s%N)
echo "$V views 1920x1080, 10 sources, APD folder 0 $*: rc $rc, wall $(( (t2-t1)/1000000 )) ms (folder written in $(( (t1-t0)/1000000 )) ms)"
grep -E "rank\(s\)|Device buffers|Stages|Fused|Fusion|Exchanges|All passes" /tmp/c4.log | tail -8
[ $rc -ne 0 ] && tail -5 /tmp/c4.log
md5sum $d/APD/APD.ply
