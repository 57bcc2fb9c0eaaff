# A/B on one box: colour images of the fusion decoded + uploaded behind the passes (default) against after them (--late-fusion-inputs)
d=/tmp/tt24; rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
run() { # label flags...
  local label=$1; shift
  rm -rf $d/APD
  local t0=$(date +%s%N); apd-mvs_amd/_build/APD $d 0 --seed 7 "$@" > /tmp/ab.log 2>&1; local t1=$(date +%s%N)
  echo "$label: wall $(( (t1-t0)/1000000 )) ms | $(grep -E '^Stages' /tmp/ab.log) | $(md5sum $d/APD/APD.ply | cut -c1-8)"
}
run "warm-up run"
run "default" ; run "late fusion inputs" --late-fusion-inputs
run "default" ; run "late fusion inputs" --late-fusion-inputs
