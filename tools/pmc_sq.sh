#!/bin/bash
# SQ counter passes for the hot kernel (each group in its own rocprofv3 run). Usage: tools/pmc_sq.sh <tag> [bench args]
set -u
TAG=${1:-sq}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="${@:---workload synthetic_4096x3072_8src --steps 2 --warmup 2 --no-cpu-baseline}"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/g$i -o pmc -- python bench.py $ARGS > $OUT/bench_g$i.json 2> $OUT/g$i.err || echo "group $i failed: $(tail -2 $OUT/g$i.err)"
  python tools/pmc_summary.py $OUT/g$i $OUT/g${i}_summary.csv > /dev/null
done
find $OUT -type f -size +1M -delete
grep -h k67 $OUT/g*_summary.csv | cut -d, -f2-6 | sed 's/"//g'
