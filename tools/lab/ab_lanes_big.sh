#!/bin/bash
# Views in flight at 6200 x 4130 with a hardware queue per lane: does a tag-bound K9/K10 of one view overlap a VALU-bound K14 of another?
# Usage: tools/lab/ab_lanes_big.sh [out_dir]
O=${1:-gpurun_out/r06_lanes_big}
mkdir -p $O
d=/tmp/big6
rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 6200 --height 4130 --views 6 --src 5 --textureless 0.2 --jpeg > /dev/null
run() {  # name flags
  local name=$1; shift
  rm -rf ${d}_$name; cp -r $d ${d}_$name
  apd-mvs_amd/_build/APD ${d}_$name 0 --seed 12345 "$@" > $O/$name.log 2>&1 || tail -5 $O/$name.log
  echo "== $name: APD folder 0 $*: $(grep -E '^Stages' $O/$name.log | sed 's/.*passes \([0-9]*\) ms.*/passes \1 ms/')  $(md5sum ${d}_$name/APD/APD.ply | cut -c1-8)"
  rm -rf ${d}_$name
}
{
run warm
run default
for n in 1 2 3 4; do run ranks$n --ranks $n; done
run default_again
} 2>&1 | tee $O/ab_lanes_big.txt
