#!/bin/bash
# Counter groups over the three-pass pipeline of tools/pass_timing.py (every kernel of an APD pass, K14/K15 included).
# Usage: tools/pmc_pass.sh <tag> "<group1>" "<group2>" ...   (size via PMC_PASS="4096 3072 8")
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/g$i -o pmc -- python tools/pass_timing.py ${PMC_PASS:-4096 3072 8} > $OUT/pass_g$i.txt 2> $OUT/g$i.err || echo "group $i failed: $(tail -2 $OUT/g$i.err)"
  python tools/pmc_summary.py $OUT/g$i $OUT/g${i}_summary.csv > /dev/null
done
find $OUT -type f -size +1M -delete
grep -hE "k14w|k15w|k910|k67w|k3_" $OUT/g*_summary.csv | python -c "
import csv,sys
for r in csv.reader(sys.stdin):
    print('%-28s %-26s n=%s mean=%.4g  [%s]' % (r[0][:28], r[1], r[2], float(r[3]), r[-1][:80]))"
