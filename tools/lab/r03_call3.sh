#!/bin/bash
# Round 3, GPU call 3: full GPU suite on the ordered-list K9/K10 + K3 over lists; L1 tag accesses by lane address pattern.
OUT=gpurun_out/r03c; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -12
echo "== APD bench"; timeout 600 python bench.py --workload eth3d_pipes_fullres_10src_apd --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_apd.json | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['kernel_ms_timed_region'])"
echo "== three-pass timing 4096x3072"
timeout 600 python tools/pass_timing.py 2>&1 | grep -E "== pass|K3 |K9 |K14|K15" | tee $OUT/pass_timing.txt
echo "== tcp patterns"
tools/_build/tcp_patterns | tee $OUT/tcp_patterns.txt
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD TCP_TCC_READ_REQ_sum --output-format csv -d /tmp/tcpp -o p -- tools/_build/tcp_patterns > /dev/null 2>&1
python - <<'PY' | tee -a $OUT/tcp_patterns.txt
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/tcpp/**/*counter_collection.csv', recursive=True)[0])))
d = collections.defaultdict(dict)
for r in rows:
    if 'gather' in r['Kernel_Name']:
        d[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
ids = sorted(d)
for n, i in enumerate(ids[1::2]):   # second launch of every pattern
    c = d[i]
    print('pattern %2d  tag accesses per gather %6.2f   L1->L2 requests per gather %6.2f' % (n, c['TCP_TOTAL_CACHE_ACCESSES_sum'] / c['SQ_INSTS_VMEM_RD'], c['TCP_TCC_READ_REQ_sum'] / c['SQ_INSTS_VMEM_RD']))
PY
ls $OUT
