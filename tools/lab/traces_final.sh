#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the final build: the driver's bench command (workloads block included) and the 24-view end-to-end run.
O=gpurun_out/traces; mkdir -p $O; cd /root/repo; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/apd_trace_driver -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_traced.json 2> $O/trace_driver.err
find /tmp/apd_trace_driver -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_driver_s20_w5_with_workloads.csv \;
head -6 $O/kernel_stats_bench_driver_s20_w5_with_workloads.csv | cut -c1-150
d=/tmp/tt24; rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tt24 -o tt24 -- apd-mvs_amd/_build/APD $d 0 --seed 7 --clean-exit > $O/tt24_prof.log 2>&1
find /tmp/prof_tt24 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_e2e_tt24_final.csv \;
head -12 $O/kernel_stats_e2e_tt24_final.csv | cut -c1-150; grep -E "Stages" $O/tt24_prof.log
