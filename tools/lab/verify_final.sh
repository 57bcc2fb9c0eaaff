#!/bin/bash
# Last check of a round on the final build: full GPU suite, smoke, the driver's bench command, kernel trace of the 24-view end-to-end run.
mkdir -p gpurun_out/verify; cd /root/repo; O=gpurun_out/verify
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
(time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench_driver_s20_w5.json 2> $O/bench.err; head -c 400 $O/bench_driver_s20_w5.json; echo; tail -4 $O/bench.err
d=/tmp/tt24; rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tt24 -o tt24 -- /root/repo/apd-mvs_amd/_build/APD $d 0 --seed 7 --clean-exit > /tmp/tt24_prof.log 2>&1
f=$(find /tmp/prof_tt24 -name "*kernel_stats.csv" | head -1); cp "$f" /root/repo/$O/kernel_stats_e2e_tt24_final.csv; head -8 /root/repo/$O/kernel_stats_e2e_tt24_final.csv | cut -c1-160
grep -E "Stages" /tmp/tt24_prof.log
