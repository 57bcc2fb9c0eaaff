#!/bin/bash
# two ranks on one device, 24 x 1080p: where do the 1.3 s between --no-rccl and --rccl go?  Usage: tools/lab/ab_exchange_two.sh [out_dir]
O=${1:-gpurun_out/r06_exchange2}
mkdir -p $O
d=/tmp/tt24
rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
run() {
  local name=$1 dev=$2; shift 2
  rm -rf ${d}_$name; cp -r $d ${d}_$name
  apd-mvs_amd/_build/APD ${d}_$name $dev --seed 12345 "$@" > $O/$name.log 2>&1 || tail -5 $O/$name.log
  echo "== $name: APD folder $dev $*"
  grep -E "^Stages|^Exchanges" $O/$name.log
}
{
run warm 0 --jacobi
run two_rccl 0,0 --rccl
run two_copy 0,0 --no-rccl
run two_copy_sync 0,0 --no-rccl --exchange-device-sync
run two_copy_latefusion 0,0 --no-rccl --late-fusion-inputs
run two_rccl_latefusion 0,0 --rccl --late-fusion-inputs
run two_copy_again 0,0 --no-rccl
} 2>&1 | tee $O/ab_exchange_two.txt
