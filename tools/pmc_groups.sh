#!/bin/bash
# Ad-hoc counter groups for the hot kernels: tools/pmc_groups.sh <tag> "<group1>" "<group2>" ...   (bench args via PMC_ARGS)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="${PMC_ARGS:---workload synthetic_4096x3072_8src --steps 2 --warmup 1 --no-cpu-baseline}"
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/g$i -o pmc -- python bench.py --no-workloads $ARGS > $OUT/bench_g$i.json 2> $OUT/g$i.err || echo "group $i failed: $(tail -2 $OUT/g$i.err)"
  python tools/pmc_summary.py $OUT/g$i $OUT/g${i}_summary.csv > /dev/null
done
find $OUT -type f -size +1M -delete
grep -h "k67\|k910" $OUT/g*_summary.csv | python -c "
import csv,sys
for r in csv.reader(sys.stdin):
    print('%-20s %-34s n=%s mean=%.4g  [%s]' % (r[0][10:30], r[1], r[2], float(r[3]), r[-1]))"
