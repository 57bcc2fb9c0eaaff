# A/B on one box: level images shared by the handles of a rank (default) against copied and packed per (view, pass) (--copy-images)
d=/tmp/tt24; rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
run() { # label flags...
  local label=$1; shift
  rm -rf $d/APD
  local t0=$(date +%s%N); timeout 300 apd-mvs_amd/_build/APD $d 0 --seed 7 "$@" > /tmp/ab.log 2>&1; local t1=$(date +%s%N)
  echo "$label: wall $(( (t1-t0)/1000000 )) ms | $(grep -E '^Stages' /tmp/ab.log) | $(md5sum $d/APD/APD.ply | cut -c1-8)"
}
run "warm-up run"
run "shared" ; run "copied" --copy-images
run "shared" ; run "copied" --copy-images
run "shared, one view in flight" --ranks 1; run "copied, one view in flight" --ranks 1 --copy-images
