# End to end on the round's final build: tools/e2e_timing.sh, tools/e2e_c4.sh 152, and 24 views of 1080p on the easy and the hard scene
O=gpurun_out/lab; mkdir -p $O
bash tools/e2e_timing.sh both > $O/e2e_timing.txt 2>&1; grep "^==" $O/e2e_timing.txt
bash tools/e2e_c4.sh 152 > $O/e2e_c4_152.txt 2>&1; head -3 $O/e2e_c4_152.txt; grep Stages $O/e2e_c4_152.txt
{
for kind in easy hard; do
  d=/tmp/tt24_$kind; rm -rf $d
  python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg $([ $kind = hard ] && echo --hard) > /dev/null
  t0=$(date +%s%N); apd-mvs_amd/_build/APD $d 0 --seed 12345 > /tmp/tt24_$kind.log 2>&1; rc=$?; t1=$(date +%s%N)
  echo "== 24 views of 1920x1080, 10 sources, $kind scene, APD folder 0: rc $rc, wall $(( (t1-t0)/1000000 )) ms"
  grep -E "Stages|Fused|All passes" /tmp/tt24_$kind.log
done
} > $O/e2e_tt24_easy_vs_hard.txt 2>&1; cat $O/e2e_tt24_easy_vs_hard.txt
