O=gpurun_out/lab; mkdir -p $O
bash tools/profile_round.sh $O r05 > $O/profile_round.log 2>&1; tail -n 45 $O/profile_round.log
cat $O/errors.txt 2>/dev/null
(APD_FUZZ_HARD=1 timeout 900 python tools/parity_fuzz.py 300 60000) > $O/parity_fuzz_hard_300.txt 2>&1; tail -n 2 $O/parity_fuzz_hard_300.txt
