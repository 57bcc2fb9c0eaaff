#!/bin/bash
# Rebuild the library with different -D flags on the GPU box and bench each. Usage: tools/tune.sh "<flags1>" "<flags2>" ...
ARGS="--workload synthetic_4096x3072_8src --steps 2 --warmup 2 --no-cpu-baseline"
for f in "$@"; do
  APD_EXTRA_FLAGS="$f" python apd-mvs_amd/build.py --force > /tmp/build.log 2>&1 || { echo "BUILD FAILED for $f"; tail -5 /tmp/build.log; continue; }
  echo "== flags: [$f]"
  timeout 300 python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('value', d['value'], 'ms/launch', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'q', d['quality_within_1pct_depth'])"
done
