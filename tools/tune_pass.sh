#!/bin/bash
# Rebuild with each flag set and print the per-kernel times of the three-pass pipeline (tools/pass_timing.py).
# Usage: [TUNE_PASS="4096 3072 8"] [TUNE_GREP="K14|K15"] tools/tune_pass.sh "<flags1>" "<flags2>" ...
export APD_ALLOW_STALE_LIBRARY=1   # lab builds with ad-hoc flags: the build-id guard of apd_mvs_amd.lib() is for the product
for f in "$@"; do
  APD_EXTRA_FLAGS="$f" python apd-mvs_amd/build.py --force > /tmp/build.log 2>&1 || { echo "BUILD FAILED for $f"; tail -5 /tmp/build.log; continue; }
  echo "== flags: [$f]"
  timeout 600 python tools/pass_timing.py ${TUNE_PASS:-4096 3072 8} 2>/dev/null | grep -E "== pass|${TUNE_GREP:-K5 |K6 |K9 |K14|K15}"
done
