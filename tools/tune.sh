#!/bin/bash
# Rebuild the library with different -D flags on the GPU box and bench each.
# Usage: [TUNE_WORKLOAD=synthetic_4096x3072_8src_apd] [TUNE_STEPS=3] tools/tune.sh "<flags1>" "<flags2>" ...
WL=${TUNE_WORKLOAD:-synthetic_4096x3072_8src}
ARGS="--workload $WL --steps ${TUNE_STEPS:-3} --warmup 1 --no-cpu-baseline"
export APD_ALLOW_STALE_LIBRARY=1   # lab builds with ad-hoc flags: the build-id guard of apd_mvs_amd.lib() is for the product
for f in "$@"; do
  APD_EXTRA_FLAGS="$f" python apd-mvs_amd/build.py --force > /tmp/build.log 2>&1 || { echo "BUILD FAILED for $f"; tail -5 /tmp/build.log; continue; }
  echo "== flags: [$f] workload $WL"
  timeout 600 python bench.py --full-line --no-workloads $ARGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline())
w=d.get('weak_path') or {}
it=d.get('iterations') or {}
print('value', d['value'], 'first_ms', it.get('first_ms'), 'later_ms', it.get('later_ms_per_step'), 'k67 ms/launch', d['roofline']['avg_launch_ms'], 'k910 ms/launch', w.get('avg_launch_ms'), 'post_loop_ms', d.get('post_loop_ms'), 'q', d['quality_within_1pct_depth'])"
  if [ -n "${TUNE_PASS:-}" ]; then  # per-kernel times of the three-pass pipeline (K5, K14, K15, K9/K10 with geometry)
    timeout 600 python tools/pass_timing.py ${TUNE_PASS} 2>/dev/null | grep -E "== pass|K5 |K6 |K9 |K14|K15"
  fi
done
