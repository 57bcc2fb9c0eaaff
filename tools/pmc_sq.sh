#!/bin/bash
# SQ / TCP / TA counter passes for the hot kernel (each group in its own rocprofv3 run, kernel-trace only).
# Usage: tools/pmc_sq.sh <tag> [bench args]
set -u
TAG=${1:-sq}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="${@:---workload synthetic_4096x3072_8src --steps 2 --warmup 1 --no-cpu-baseline}"
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TD_TD_BUSY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/g$i -o pmc -- python bench.py --no-workloads $ARGS > $OUT/bench_g$i.json 2> $OUT/g$i.err || echo "group $i failed: $(tail -2 $OUT/g$i.err)"
  python tools/pmc_summary.py $OUT/g$i $OUT/g${i}_summary.csv > /dev/null
done
find $OUT -type f -size +1M -delete
grep -h k67 $OUT/g*_summary.csv | python -c "
import csv,sys
for r in csv.reader(sys.stdin):
    print('%-34s n=%s mean=%.4g  [%s]' % (r[1], r[2], float(r[3]), r[-1]))"
