export APD_ALLOW_STALE_LIBRARY=1   # lab builds with ad-hoc flags
O=gpurun_out/lab; mkdir -p $O
export TMPDIR=/tmp
(time python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_parity.py -m gpu -x -q -k "hard") > $O/pytest_hard.log 2>&1; tail -5 $O/pytest_hard.log
for key in configs2_pipes_apd_whole_pass configs2_pipes_apd_geometric_pass configs2_pipes_hard_whole_pass; do
  APD_PROFILE_PASS_KEY=$key timeout 1200 python tools/profile_bench.py $O > $O/profile_$key.log 2>&1 || echo "profile of $key failed"; tail -2 $O/profile_$key.log
done
mkdir -p profiles/r05 && cp $O/pmc_pass_*.json profiles/r05/
(time python bench.py --steps 20 --warmup 5 --no-cpu-baseline) > $O/line_driver_s20_w5.json 2> $O/bench_driver.err; cp bench_workloads.json $O/bench_driver_s20_w5.json; wc -c $O/line_driver_s20_w5.json; cat $O/line_driver_s20_w5.json
(APD_FUZZ_HARD=1 timeout 900 python tools/parity_fuzz.py 300 60000) > $O/parity_fuzz_hard_300.txt 2>&1; tail -n 2 $O/parity_fuzz_hard_300.txt
APD_EXTRA_FLAGS=-DAPD_LAB_WIN_STATS python apd-mvs_amd/build.py --force > $O/lab_build.log 2>&1
python tools/win_stats.py 6200 4130 8 6 > $O/win_stats_easy_6200x4130_8src.txt 2>&1; python tools/win_stats.py --hard 6200 4130 8 6 > $O/win_stats_hard_6200x4130_8src.txt 2>&1
tail -n 30 $O/win_stats_easy_6200x4130_8src.txt $O/win_stats_hard_6200x4130_8src.txt
