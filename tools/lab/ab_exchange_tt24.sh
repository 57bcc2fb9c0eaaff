#!/bin/bash
# 24 views of 1920 x 1080, 10 sources: what the exchange of depth maps costs with two ranks on the one device -- direct reads (round 6:
# one gather kernel per rank; round 5: one hipMemcpyAsync per block, 1.3 s behind RCCL) against RCCL, and the single rank beside them.
# Usage: tools/lab/ab_exchange_tt24.sh [out_dir]
O=${1:-gpurun_out/r06_exchange}
mkdir -p $O
d=/tmp/tt24
rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
run() {  # name devices flags...
  local name=$1 dev=$2; shift 2
  rm -rf ${d}_$name; cp -r $d ${d}_$name
  local t0=$(date +%s%N)
  apd-mvs_amd/_build/APD ${d}_$name $dev --seed 12345 "$@" > $O/$name.log 2>&1 || tail -5 $O/$name.log
  local t1=$(date +%s%N)
  echo "== $name: APD folder $dev $*: wall $(( (t1 - t0) / 1000000 )) ms"
  grep -E "^Stages|^Exchanges|Exchange of depth" $O/$name.log
  md5sum ${d}_$name/APD/APD.ply | cut -c1-32
}
{
run warm 0 --jacobi          # page cache, code objects
run one 0 --jacobi
run two_copy 0,0 --no-rccl
run two_rccl 0,0 --rccl
run eight_copy 0,0,0,0,0,0,0,0 --no-rccl
run eight_rccl 0,0,0,0,0,0,0,0 --rccl
run one_again 0 --jacobi
} 2>&1 | tee $O/ab_exchange_tt24.txt
