#!/bin/bash
# Views in flight run on their own HIP streams; the runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).
# Does the number of hardware queues change the passes of 24 x 1080p?  Usage: tools/lab/ab_hw_queues.sh [out_dir]
O=${1:-gpurun_out/r06_hwq}
mkdir -p $O
d=/tmp/tt24
rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
run() {  # name queues devices flags
  local name=$1 q=$2 dev=$3; shift 3
  rm -rf ${d}_$name; cp -r $d ${d}_$name
  if [ "$q" = default ]; then
    apd-mvs_amd/_build/APD ${d}_$name $dev --seed 12345 "$@" > $O/$name.log 2>&1 || tail -5 $O/$name.log
  else
    GPU_MAX_HW_QUEUES=$q apd-mvs_amd/_build/APD ${d}_$name $dev --seed 12345 "$@" > $O/$name.log 2>&1 || tail -5 $O/$name.log
  fi
  echo "== $name: GPU_MAX_HW_QUEUES=$q APD folder $dev $*: $(grep -E '^Stages' $O/$name.log | sed 's/.*passes \([0-9]*\) ms.*/passes \1 ms/')  $(md5sum ${d}_$name/APD/APD.ply | cut -c1-8)"
}
{
run warm default 0
for q in default 2 4 6 8 12; do run one_q$q $q 0; done
for q in default 8; do run two_copy_q$q $q 0,0 --no-rccl; done
for q in default 8; do run jacobi_q$q $q 0 --jacobi; done
} 2>&1 | tee $O/ab_hw_queues.txt
