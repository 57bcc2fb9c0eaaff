#!/bin/bash
# Round 3, final measurement set on the round's last kernel build: counter profiles + bench lines of every workload, whole passes, end to end.
OUT=gpurun_out/r03final; mkdir -p $OUT; export TMPDIR=/tmp
date +%s > $OUT/t0
bash tools/profile_round.sh $OUT 2>&1 | tail -8 | tee $OUT/profile_round_tail.txt
date +%s > $OUT/t1
timeout 600 python tools/pass_timing.py 4096 3072 8 0.2 2>/dev/null | grep -v "^HIP\|^ROCm" > $OUT/pass_timing.txt; tail -30 $OUT/pass_timing.txt
bash tools/e2e_timing.sh both 2>&1 | tee $OUT/e2e_timing.txt
rm -rf /tmp/e2e_small_c/APD; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e2e_prof -o e2e -- apd-mvs_amd/_build/APD /tmp/e2e_small_c 0 --jacobi --seed 12345 > /dev/null 2>&1
cp $(find /tmp/e2e_prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_e2e_small_jacobi.csv 2>/dev/null
date +%s > $OUT/t2
find $OUT -type f -size +1M -delete
