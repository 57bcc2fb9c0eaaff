#!/bin/bash
# End-to-end wall times of DESIGN.md section 6 on the GPU box: the in-memory pipeline (tools/mvs_pipeline.py) and the file-based
# drop-in binary over the same synthetic dense folders; the two APD.ply files must be identical.
# Usage: tools/e2e_timing.sh [small|big|both]
set -u
WHAT=${1:-both}
run_case() {  # name width height views src
  local d=/tmp/e2e_$1
  rm -rf $d ${d}_b; mkdir -p $d
  python tools/make_synthetic_dense.py $d --width $2 --height $3 --views $4 --src $5 --textureless 0.2 --jpeg > /dev/null
  cp -r $d ${d}_b
  local t0=$(date +%s%N)
  python tools/mvs_pipeline.py $d --seed 12345 > /tmp/e2e_pipe.log 2>&1 || tail -5 /tmp/e2e_pipe.log
  local t1=$(date +%s%N)
  apd-mvs_amd/_build/APD ${d}_b 0 --files --seed 12345 > /tmp/e2e_bin.log 2>&1 || tail -5 /tmp/e2e_bin.log
  local t2=$(date +%s%N)
  # the C++ multi-device scheduler with one rank (host/multi_device.cpp: state resident, depth maps exchanged in memory;
  # Jacobi over views, so its cloud differs slightly from the two Gauss-Seidel runs above by construction)
  rm -rf ${d}_c; cp -r ${d}_b ${d}_c; rm -rf ${d}_c/APD
  local t3=$(date +%s%N)
  apd-mvs_amd/_build/APD ${d}_c 0 --jacobi --seed 12345 > /tmp/e2e_mem.log 2>&1 || tail -5 /tmp/e2e_mem.log
  local t4=$(date +%s%N)
  # ... and in the reference's own order (--in-memory): the bytes of the file-based run
  rm -rf ${d}_d; cp -r ${d}_b ${d}_d; rm -rf ${d}_d/APD
  local t5=$(date +%s%N)
  apd-mvs_amd/_build/APD ${d}_d 0 --seed 12345 > /tmp/e2e_gs.log 2>&1 || tail -5 /tmp/e2e_gs.log
  local t6=$(date +%s%N)
  echo "== $1: drop-in binary as the reference calls it (APD folder 0: in memory, reference order) $(( (t6 - t5) / 1000000 )) ms"
  grep -iE "stages" /tmp/e2e_gs.log | tail -1
  md5sum ${d}_d/APD/APD.ply
  echo "== $1: $4 views of $2x$3, $5 sources: in-memory pipeline $(( (t1 - t0) / 1000000 )) ms (python start-up included), drop-in binary through the files (APD folder 0 --files) $(( (t2 - t1) / 1000000 )) ms, drop-in binary with the in-memory scheduler (APD folder 0 --jacobi) $(( (t4 - t3) / 1000000 )) ms"
  grep -iE "fusion|points|total|pass" /tmp/e2e_pipe.log | tail -4
  grep -iE "fus|points|total|exchange|gather" /tmp/e2e_mem.log | tail -6
  md5sum $d/APD/APD.ply ${d}_b/APD/APD.ply
}
if [ "$WHAT" != big ]; then run_case small 1920 1080 12 8; fi
if [ "$WHAT" != small ]; then run_case big 6200 4130 4 3; fi
