# K14 after the round-5 restructure: waves per SIMD, chunk size, pair walk threshold, corner reuse (6200 x 4130, 10 sources, three pass kinds)
O=gpurun_out/lab; mkdir -p $O
TUNE_PASS="6200 4130 10" TUNE_GREP="K14|K15" bash tools/tune_pass.sh "" "-DAPD_K14W_WAVES=3" "-DAPD_K14_CHUNK=16" "-DAPD_K14_CHUNK=4" "-DAPD_K14_PAIRS_FROM_N=99" "-DAPD_WIN_CORNER_REUSE=0" "-DAPD_K14W_WAVES=3 -DAPD_K14_CHUNK=16" 2>&1 | tee $O/tune_k14.txt
