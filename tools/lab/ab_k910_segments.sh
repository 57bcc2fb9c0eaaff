# K9/K10 with / without the 16-byte row-segment loads of the sub-patch taps: time + L1 counters (configs[2])
cd /root/repo
for v in 1 0; do
  APD_EXTRA_FLAGS="-DAPD_SUBPATCH_ROW_SEGMENTS=$v" python apd-mvs_amd/build.py --force > /tmp/b.log 2>&1 || { tail -3 /tmp/b.log; continue; }
  echo "== APD_SUBPATCH_ROW_SEGMENTS=$v"
  python bench.py --workload eth3d_pipes_fullres_10src_apd --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('value', d['value'], 'k910 ms/launch', d['weak_path']['avg_launch_ms'])"
  for grp in tcp sq; do
    APD_PROFILE_PASSES=$grp python tools/profile_bench.py /tmp/prof_$v --workload eth3d_pipes_fullres_10src_apd --steps 3 --warmup 1 2>&1 | grep -E "^k910"
  done
done
