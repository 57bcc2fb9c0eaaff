#!/bin/bash
# Runs on the GPU box: default bench line + rocprofv3 kernel trace + PMC passes (each in its own run).
# Usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="${@:---steps 6 --warmup 1 --no-cpu-baseline}"
echo "== bench (default flags)"; python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json
echo "== kernel trace"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --no-workloads $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "== pmc FETCH_SIZE"; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python bench.py --no-workloads $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "== pmc WRITE_SIZE"; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python bench.py --no-workloads $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
echo "== pmc SQ"; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq -o pmc -- python bench.py --no-workloads $ARGS > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
for d in pmc_fetch pmc_write pmc_sq trace; do python tools/pmc_summary.py $OUT/$d $OUT/${d}_summary.csv; done
python tools/pmc_summary.py --traffic $OUT/pmc_fetch_summary.csv $OUT/pmc_write_summary.csv eth3d_office_fullres_8src $OUT/pmc_traffic.json $OUT/pmc_sq_summary.csv
find $OUT -type f -size +1M -print -delete
find $OUT -type f | xargs ls -la | head -60
du -sh $OUT
