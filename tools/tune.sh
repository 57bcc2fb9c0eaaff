#!/bin/bash
# Rebuild the library with different -D flags on the GPU box and bench each.
# Usage: [TUNE_WORKLOAD=synthetic_4096x3072_8src_apd] tools/tune.sh "<flags1>" "<flags2>" ...
WL=${TUNE_WORKLOAD:-synthetic_4096x3072_8src}
ARGS="--workload $WL --steps 2 --warmup 2 --no-cpu-baseline"
for f in "$@"; do
  APD_EXTRA_FLAGS="$f" python apd-mvs_amd/build.py --force > /tmp/build.log 2>&1 || { echo "BUILD FAILED for $f"; tail -5 /tmp/build.log; continue; }
  echo "== flags: [$f]"
  timeout 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline())
w=d.get('weak_path') or {}
print('value', d['value'], 'k67 ms/launch', d['roofline']['avg_launch_ms'], 'k910 ms/launch', w.get('avg_launch_ms'), 'q', d['quality_within_1pct_depth'])"
done
