#!/bin/bash
# Counter passes for the window-less first iteration of K6/K7 (the <.., true, true, ..> instantiation) beside the converged ones, configs[1] (each group in its own rocprofv3 run, kernel-trace only).  Usage: tools/lab/pmc_iter0.sh <tag>
set -u
TAG=${1:-iter0}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--workload ${APD_PMC_WORKLOAD:-eth3d_office_fullres_8src} --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/g$i -o pmc -- python bench.py --no-workloads $ARGS > $OUT/bench_g$i.json 2> $OUT/g$i.err || echo "group $i failed: $(tail -2 $OUT/g$i.err)"
  python tools/pmc_summary.py $OUT/g$i $OUT/g${i}_summary.csv > /dev/null
done
find $OUT -type f -size +1M -delete
grep -h "k910\|k67" $OUT/g*_summary.csv | python -c "
import csv,sys
for r in csv.reader(sys.stdin):
    print('%-44s %-34s n=%s mean=%.5g' % (r[0].replace('void apd::','')[:44], r[1], r[2], float(r[3])))" | tee $OUT/summary.txt
