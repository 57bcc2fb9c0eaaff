#!/bin/bash
d=/tmp/tt24; rm -rf $d ${d}_j; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
cp -r $d ${d}_j
t0=$(date +%s%N); apd-mvs_amd/_build/APD $d 0 --seed 7 > /tmp/tt_a.log 2>&1; t1=$(date +%s%N)
apd-mvs_amd/_build/APD ${d}_j 0 --jacobi --seed 7 > /tmp/tt_b.log 2>&1; t2=$(date +%s%N)
echo "24 views 1920x1080, 10 sources: APD folder 0: $(( (t1-t0)/1000000 )) ms; --jacobi: $(( (t2-t1)/1000000 )) ms"
grep -E "rank\(s\)|Stages|Fused" /tmp/tt_a.log | tail -3; grep -E "rank\(s\)|Stages|Fused" /tmp/tt_b.log | tail -3
