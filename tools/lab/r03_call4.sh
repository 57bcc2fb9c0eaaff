#!/bin/bash
# Round 3, GPU call 4: multi-device host + subset folders; L1 tag accesses by lane pattern (opaque loads).
OUT=gpurun_out/r03d; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest drop-in binary"; timeout 1500 python -m pytest tests/test_gpu_dropin_binary.py -m gpu -q -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -30
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
