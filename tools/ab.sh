#!/bin/bash
# A/B of library options on the GPU box (no rebuild): tools/ab.sh "<--opt name=value ...>" "<...>" ; workload via TUNE_WORKLOAD
WL=${TUNE_WORKLOAD:-synthetic_4096x3072_8src}
ARGS="--workload $WL --steps ${TUNE_STEPS:-3} --warmup 1 --no-cpu-baseline"
for e in "$@"; do
  echo "== options: [$e] workload $WL"
  timeout 600 python bench.py --full-line --no-workloads $ARGS $e 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline())
w=d.get('weak_path') or {}
it=d.get('iterations') or {}
print('value', d['value'], 'first_ms', it.get('first_ms'), 'later_ms', it.get('later_ms_per_step'), 'k67 ms/launch', d['roofline']['avg_launch_ms'], 'k910 ms/launch', w.get('avg_launch_ms'), 'q', d['quality_within_1pct_depth'])"
done
