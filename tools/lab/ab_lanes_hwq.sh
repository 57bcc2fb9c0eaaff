#!/bin/bash
# Views in flight per device again, now that every lane's stream has a hardware queue of its own (GPU_MAX_HW_QUEUES = 8 set by the binary).
# Usage: tools/lab/ab_lanes_hwq.sh [out_dir]
O=${1:-gpurun_out/r06_lanes}
mkdir -p $O
d=/tmp/tt24
rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
run() {  # name queues flags
  local name=$1 q=$2; shift 2
  rm -rf ${d}_$name; cp -r $d ${d}_$name
  GPU_MAX_HW_QUEUES=$q apd-mvs_amd/_build/APD ${d}_$name 0 --seed 12345 "$@" > $O/$name.log 2>&1 || tail -5 $O/$name.log
  echo "== $name: GPU_MAX_HW_QUEUES=$q APD folder 0 $*: $(grep -E '^Stages' $O/$name.log | sed 's/.*passes \([0-9]*\) ms.*/passes \1 ms/')  $(md5sum ${d}_$name/APD/APD.ply | cut -c1-8)"
}
{
run warm 8
run default_q8 8
for n in 2 3 4 6 8 10; do run ranks${n}_q8 8 --ranks $n; done
for n in 8 12; do run ranks${n}_q16 16 --ranks $n; done
run default_q8_again 8
} 2>&1 | tee $O/ab_lanes_hwq.txt
