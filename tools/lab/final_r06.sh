# round 6, final build: parity sweeps with new seeds (the pair variant of K14 now runs from 2 sources on in geometric passes, K9/K10 with
# lane = (pixel, hypothesis)), the fusion sweep, and the end-to-end runs of DESIGN.md section 6
O=gpurun_out/r06_final; mkdir -p $O
(timeout 1200 python tools/parity_fuzz.py 400 90000) > $O/parity_fuzz_400_final_build.txt 2>&1; tail -n 1 $O/parity_fuzz_400_final_build.txt
(APD_FUZZ_HARD=1 timeout 900 python tools/parity_fuzz.py 200 91000) > $O/parity_fuzz_hard_200_final_build.txt 2>&1; tail -n 1 $O/parity_fuzz_hard_200_final_build.txt
(APD_FUZZ_HARD=1 APD_FUZZ_SCALE=3 timeout 1200 python tools/parity_fuzz.py 80 92000) > $O/parity_fuzz_hard_80_scale3_final_build.txt 2>&1; tail -n 1 $O/parity_fuzz_hard_80_scale3_final_build.txt
(timeout 600 python tools/fusion_fuzz.py 40 950) > $O/fusion_fuzz_40_final_build.txt 2>&1; tail -n 1 $O/fusion_fuzz_40_final_build.txt
bash tools/e2e_timing.sh both > $O/e2e_timing.txt 2>&1; grep "^==" $O/e2e_timing.txt
bash tools/e2e_c4.sh 152 > $O/e2e_c4_152.txt 2>&1; head -3 $O/e2e_c4_152.txt; tail -1 $O/e2e_c4_152.txt
