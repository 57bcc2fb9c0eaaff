#!/bin/bash
# Round 3, GPU call 5: full suite after the clean-up, APD bench, wide-load access counts.
OUT=gpurun_out/r03e; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -8
echo "== APD bench"; timeout 600 python bench.py --workload eth3d_pipes_fullres_10src_apd --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_apd.json | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['kernel_ms_timed_region'], d['roofline']['bound'])"
echo "== default bench"; timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tee $OUT/bench_default.json | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['iterations'], d['kernel_ms_timed_region'])"
echo "== tcp patterns"
tools/_build/tcp_patterns | tail -8 | tee $OUT/tcp_patterns_wide.txt
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD TCP_TCC_READ_REQ_sum --output-format csv -d /tmp/tcpp -o p -- tools/_build/tcp_patterns > /dev/null 2>&1
python - <<'PY' | tee -a $OUT/tcp_patterns_wide.txt
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/tcpp/**/*counter_collection.csv', recursive=True)[0])))
d = collections.defaultdict(dict)
for r in rows:
    if 'gather_wide' in r['Kernel_Name']:
        d[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        d[int(r['Dispatch_Id'])]['name'] = r['Kernel_Name'][:40]
ids = sorted(d)
for n, i in enumerate(ids[1::2]):
    c = d[i]
    print('%s pattern %d  tag accesses per load instruction %6.2f   L1->L2 requests %6.2f' % (c['name'], n % 3, c['TCP_TOTAL_CACHE_ACCESSES_sum'] / c['SQ_INSTS_VMEM_RD'], c['TCP_TCC_READ_REQ_sum'] / c['SQ_INSTS_VMEM_RD']))
PY
