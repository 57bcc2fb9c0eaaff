# A/B on one box: views in flight per rank (--ranks N) in the reference's order and with --jacobi; 24 views of 1080p, 10 sources
d=/tmp/tt24; rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
run() { # label flags...
  local label=$1; shift
  rm -rf $d/APD
  local t0=$(date +%s%N); timeout 300 apd-mvs_amd/_build/APD $d 0 --seed 7 "$@" > /tmp/ab.log 2>&1; local rc=$?; local t1=$(date +%s%N)
  echo "$label: rc $rc wall $(( (t1-t0)/1000000 )) ms | $(grep -E '^Stages' /tmp/ab.log) | $(md5sum $d/APD/APD.ply | cut -c1-8)"
}
run "warm-up"
run "default (table: 6 at the coarse level, 3 at 1080p)"
for k in 1 2 3 4 6 8; do run "--ranks $k" --ranks $k; done
for k in 1 3 6; do run "--jacobi --ranks $k" --jacobi --ranks $k; done
run "--files" --files
