#!/bin/bash
# Same-box A/B of K14's phase-0 test: tools/_build/libapd_old.so (every sample within the radius above 0.5) against the built library
# (no local minimum of at most 0.5 within the radius).  Per-kernel pass timing at 4096x3072 / 8 sources and 24 views of 1080p end to end.
# The 'old' library is a build of the commit before the change, copied to tools/_build/libapd_old.so (git-ignored, travels with the snapshot).
mkdir -p gpurun_out/k14peak; cd /root/repo; O=gpurun_out/k14peak
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_fullsize_parity.py > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cp apd-mvs_amd/_build/libapd_mi355x.so /tmp/new.so
for which in new old new old; do
  if [ $which = old ]; then cp tools/_build/libapd_old.so apd-mvs_amd/_build/libapd_mi355x.so; else cp /tmp/new.so apd-mvs_amd/_build/libapd_mi355x.so; fi
  touch apd-mvs_amd/_build/libapd_mi355x.so
  echo "== $which" >> $O/pass_timing.txt
  timeout 300 python tools/pass_timing.py 4096 3072 8 0.2 2>/dev/null | grep -E "== pass|K14|K15" >> $O/pass_timing.txt
  echo "== $which" >> $O/tt24.txt
  timeout 300 tools/lab/tt_like.sh 2>&1 | grep -E "views 1920|Stages" | head -2 >> $O/tt24.txt
done
cp /tmp/new.so apd-mvs_amd/_build/libapd_mi355x.so
cat $O/pass_timing.txt $O/tt24.txt
