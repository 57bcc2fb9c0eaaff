# the two bench lines of the round (after their counter profiles: tools/profile_round.sh): stdout = compact line, bench_workloads.json = full block
O=gpurun_out/lab; mkdir -p $O
python bench.py > $O/line_default.json 2> $O/bench_default.err; cp bench_workloads.json $O/bench_default.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/line_driver_s20_w5.json 2> /dev/null; cp bench_workloads.json $O/bench_driver_s20_w5.json
wc -c $O/line_*.json; cat $O/line_driver_s20_w5.json
cp $O/line_*.json $O/bench_default.json $O/bench_driver_s20_w5.json profiles/r05/ && python tools/recompute_roofline.py profiles/r05 | tail -4
