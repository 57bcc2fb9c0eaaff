# parity sweeps on the round's final build: 700 random configurations (new seeds), 200 hard ones at three times the frame size, 40 fusion jobs
O=gpurun_out/lab; mkdir -p $O
(timeout 1500 python tools/parity_fuzz.py 700 70000) > $O/parity_fuzz_700_final_build.txt 2>&1; tail -n 1 $O/parity_fuzz_700_final_build.txt
(APD_FUZZ_HARD=1 APD_FUZZ_SCALE=3 timeout 1800 python tools/parity_fuzz.py 120 71000) > $O/parity_fuzz_hard_120_scale3_final_build.txt 2>&1; tail -n 1 $O/parity_fuzz_hard_120_scale3_final_build.txt
(timeout 600 python tools/fusion_fuzz.py 40 900) > $O/fusion_fuzz_40_final_build.txt 2>&1; tail -n 1 $O/fusion_fuzz_40_final_build.txt
