O=gpurun_out/lab; mkdir -p $O
python tools/make_synthetic_dense.py /tmp/tt24 --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
APD=apd-mvs_amd/_build/APD
run() {  # tag, args...
  tag=$1; shift
  rm -rf /tmp/tt24/APD
  t0=$(date +%s.%N); $APD /tmp/tt24 "$@" --seed 12345 > $O/$tag.log 2>&1; t1=$(date +%s.%N)
  echo "== $tag: wall $(python3 -c "print(round($t1 - $t0, 2))") s | $(grep -h 'Exchanges' $O/$tag.log | cut -c1-260)"; grep -h "Stages\|Device buffers" $O/$tag.log; md5sum /tmp/tt24/APD/APD.ply
}
{
run rccl_sync_first_cold 0,0 --rccl
run no_rccl 0,0 --no-rccl
run rccl_sync 0,0 --rccl
run rccl_async 0,0 --rccl --async-rccl
run no_rccl_devsync 0,0 --no-rccl --exchange-device-sync
run rccl_devsync 0,0 --rccl --exchange-device-sync
run single_rank 0
} 2>&1 | tee $O/ab_rccl_async_tt24.txt
